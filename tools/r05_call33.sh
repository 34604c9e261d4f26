#!/bin/bash
# round 5, GPU call 33: split-K block kernel with the per-half-step preparation (scale multiplies, next scale reads, word shuffle) moved in front of the barrier
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
rm -f gpurun_out/time_cases.jsonl
timeout 600 python -m pytest tests/test_qgemm_gpu.py -x -q -m gpu -k "splitk or seam_under_load" 2>&1 | tail -2
timeout 400 python tools/time_cases.py "4,256,4096,4096,f16;4,256,4096,4096,bf16,family=6,m_tiles=4,splitk=2;4,256,11008,4096,f16;4,1024,4096,4096,f16;4,64,8192,8192,f16;4,96,14336,4096,f16;4,256,8192,8192,f16;4,512,4096,4096,f16;2,96,8192,8192,f16;4,128,8192,8192,f16" --steps 400 --tag prep_before_barrier 2>&1 | cut -c1-250
