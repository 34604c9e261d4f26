#!/bin/bash
# round 5, GPU call 30: randomised parity sweeps on the final library (automatic plans over random template ids; the lean kernels' shapes)
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
FUZZ_OUT=gpurun_out/r05_gpu_fuzz.json timeout 500 python tools/gpu_fuzz.py 24000 505 2>&1 | tail -4
FUZZ_DECODE=3 FUZZ_OUT=gpurun_out/r05_gpu_fuzz_lean_shapes.json timeout 300 python tools/gpu_fuzz.py 16000 506 2>&1 | tail -4
