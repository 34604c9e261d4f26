#!/bin/bash
# round 5, GPU call 35: A/B on one box - block prefill kernels (qgemm_block2.h / qgemm_block3.h) with the half step's register-only preparation in
# front of the barrier (new) against the committed kernels (old), alternating; parity of the new ones first
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
rm -f gpurun_out/time_cases.jsonl
timeout 600 python -m pytest tests/test_qgemm_gpu.py -x -q -m gpu -k "block_prefill or seam_under_load or full_size" 2>&1 | tail -2
C="4,4096,4096,4096,f16;4,2048,4096,4096,f16;4,4096,4096,4096,bf16;4,2048,4096,4096,bf16;4,4096,11008,4096,f16;3,4096,4096,4096,bf16;3,1024,4096,4096,bf16;2,4096,4096,4096,f16;3,1024,28672,8192,bf16"
cp flute_amd/csrc/libflute_amd.so /tmp/new.so
for rep in 1 2; do
  for v in old new; do
    if [ $v = old ]; then cp flute_amd/csrc/libflute_amd_oldblk.so flute_amd/csrc/libflute_amd.so; else cp /tmp/new.so flute_amd/csrc/libflute_amd.so; fi
    timeout 300 python tools/time_cases.py "$C" --steps 100 --tag $v$rep 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(j['tag'], j['bits'], j['M'], j['N'], j['K'], j['dtype'], j['us'], j['plan']['family'], j['plan']['m_block'], j['plan']['splitk'])"
  done
done
cp /tmp/new.so flute_amd/csrc/libflute_amd.so
