#!/bin/bash
# round 5, GPU call 28: the split-K block kernel's K-split launches in XCD-group order (the row tiles of a (column tile, slice) on one XCD)
# against the natural order: parity / seam tests, then timings; traffic of M = 256 on 4096^2 (PMC) with the new order
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
rm -f gpurun_out/time_cases.jsonl
timeout 600 python -m pytest tests/test_qgemm_gpu.py -x -q -m gpu -k "splitk or seam_under_load or fuzz" 2>&1 | tail -2
CASES="4,256,4096,4096,f16;4,256,4096,4096,bf16;4,128,4096,4096,f16;4,64,8192,8192,f16;4,96,14336,4096,f16;4,256,8192,8192,f16;4,512,4096,4096,f16;4,80,8192,8192,f16;2,96,8192,8192,f16;4,1024,4096,4096,f16;4,384,4096,4096,f16"
echo "== xcd-group order"
timeout 300 python tools/time_cases.py "$CASES" --tag xcdgroups 2>&1 | cut -c1-290
echo "== natural order"
FLUTE_AMD_SK_NATURAL=1 timeout 300 python tools/time_cases.py "$CASES" --tag natural 2>&1 | cut -c1-290
