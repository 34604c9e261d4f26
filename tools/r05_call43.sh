#!/bin/bash
# round 5, GPU call 43: the per-wave MFMA kernel and the skinny kernel with the bf16 scale multiply of common.h's mul_scale4 (new) against the committed
# library (old), alternating on one box; parity first
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
rm -f gpurun_out/time_cases.jsonl
timeout 900 python -m pytest tests/test_qgemm_gpu.py -x -q -m gpu -k "mfma or skinny or golden or random_vs_oracle or group_sizes or ragged or fuzz" 2>&1 | tail -3
C="4,256,4096,4096,bf16;4,16,28672,8192,bf16;4,64,28672,8192,bf16;4,32,4096,4096,bf16;4,128,4096,4096,bf16;4,16,11008,4096,bf16;3,64,4096,4096,bf16;3,32,8192,8192,bf16;2,64,8192,8192,bf16;4,8,14336,4096,bf16"
cp flute_amd/csrc/libflute_amd.so /tmp/new.so
for rep in 1 2; do
  for v in old new; do
    if [ $v = old ]; then cp flute_amd/csrc/libflute_amd_oldtile.so flute_amd/csrc/libflute_amd.so; else cp /tmp/new.so flute_amd/csrc/libflute_amd.so; fi
    timeout 300 python tools/time_cases.py "$C" --steps 200 --tag $v$rep 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(j['tag'], j['bits'], j['M'], j['N'], j['K'], j['dtype'], j['us'], j['plan']['family'], j['plan']['m_block'], j['plan']['splitk'])"
  done
done
cp /tmp/new.so flute_amd/csrc/libflute_amd.so
