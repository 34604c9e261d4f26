#!/bin/bash
# round 5, GPU call 32: two / four rows on K = 4096 layers of more than one round of workgroups: the lean kernel forced (one_shot = 4) against the automatic plan
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
rm -f gpurun_out/time_cases.jsonl
C=""
for N in 8192 11008 14336; do for M in 2 3 4; do C="$C;4,$M,$N,4096,f16;4,$M,$N,4096,f16,family=0,one_shot=4"; done; done
for N in 6144 8192; do for M in 2 4; do C="$C;4,$M,$N,2048,f16;4,$M,$N,2048,f16,family=0,one_shot=4"; done; done
timeout 500 python tools/time_cases.py "${C:1}" --steps 300 --tag rows 2>&1 | cut -c1-250
