#!/bin/bash
# round 5, GPU call 38: rocprofv3 trace + PMC passes of two kernels this round changed: the 3-bit 128-row blocks x 2 K slices combined in the launch
# (M = 1024 on 4096^2, bf16) and the fused-rotation decode of configs[4] (3584 x 4096 [N x K], Hadamard 512)
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 500 bash tools/prof_case.sh w3_m1024_4096_inlaunch --bits 3 --M 1024 --N 4096 --K 4096 --bf16 --tid 4 2>&1 | tail -6
timeout 400 bash tools/prof_case.sh lean_m1_4096x3584 --bits 4 --M 1 --N 4096 --K 3584 2>&1 | tail -4
