#!/bin/bash
# round 5, GPU call 24: (a) the 3-bit blocks' K slices combined inside the launch (xwg_seam) against the reduce launch;
# (b) the 3-bit block kernel's whole-line PLANE pieces (qgemm_block3.h FLUTE_B3_LINE_PLANES = 1 / 2, written at the end of round 4,
# never run): parity, then the 3-bit prefill / mid-M cases against the default build
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
rm -f gpurun_out/time_cases.jsonl
MID="3,1024,4096,4096,bf16;3,256,8192,8192,bf16;3,512,4096,4096,bf16;3,512,8192,8192,bf16;3,128,8192,8192,bf16;3,64,8192,8192,bf16;3,256,4096,4096,bf16;3,1024,4096,4096,f16;3,96,28672,8192,bf16"
PRE="3,4096,4096,4096,bf16;3,1024,28672,8192,bf16;3,4096,4096,4096,f16;3,2048,4096,11008,bf16"
echo "== default build: tests"
timeout 600 python -m pytest tests/test_qgemm_gpu.py -x -q -m gpu -k "block_prefill or seam_under_load or splitk or fuzz" 2>&1 | tail -3
echo "== in-launch combine"
timeout 300 python tools/time_cases.py "$MID" --tag inlaunch 2>&1 | cut -c1-260
echo "== reduce launch"
FLUTE_AMD_B3_TWO_LAUNCH=1 timeout 300 python tools/time_cases.py "$MID" --tag twolaunch 2>&1 | cut -c1-260
echo "== prefill, default"
timeout 300 python tools/time_cases.py "$PRE" --tag lp0 2>&1 | cut -c1-200
cp flute_amd/csrc/libflute_amd.so /tmp/libflute_amd_default.so
for v in lp1 lp2; do
  cp flute_amd/csrc/libflute_amd_$v.so flute_amd/csrc/libflute_amd.so
  echo "== $v"
  timeout 300 python -m pytest tests/test_qgemm_gpu.py -x -q -m gpu -k "block_prefill" 2>&1 | tail -2
  timeout 300 python tools/time_cases.py "$PRE;3,1024,4096,4096,bf16;3,256,8192,8192,bf16" --tag $v 2>&1 | cut -c1-200
done
cp /tmp/libflute_amd_default.so flute_amd/csrc/libflute_amd.so
