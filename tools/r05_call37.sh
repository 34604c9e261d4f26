#!/bin/bash
# round 5, GPU call 37: split-K block kernel, loader-wave variant: stage hand-off through LDS words instead of two workgroup barriers per step (new)
# against the committed kernel (old), alternating on one box; parity of the new one first (its own timeout: a hand-off bug would hang)
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
rm -f gpurun_out/time_cases.jsonl
timeout 240 python -m pytest tests/test_qgemm_gpu.py -x -q -m gpu -k "splitk or seam_under_load" 2>&1 | tail -3
C="4,256,4096,4096,f16;4,64,8192,8192,f16;4,128,8192,8192,f16;4,96,8192,8192,f16;4,1024,4096,4096,bf16;4,256,11008,4096,bf16,family=6,m_tiles=8,splitk=1;4,512,4096,4096,bf16,family=6,m_tiles=8,splitk=2"
cp flute_amd/csrc/libflute_amd.so /tmp/new.so
for rep in 1 2; do
  for v in old new; do
    if [ $v = old ]; then cp flute_amd/csrc/libflute_amd_oldsk.so flute_amd/csrc/libflute_amd.so; else cp /tmp/new.so flute_amd/csrc/libflute_amd.so; fi
    timeout 200 python tools/time_cases.py "$C" --steps 300 --tag $v$rep 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(j['tag'], j['bits'], j['M'], j['N'], j['K'], j['dtype'], j['us'], j['plan']['family'], j['plan']['m_tiles'], j['plan']['splitk'])"
  done
done
cp /tmp/new.so flute_amd/csrc/libflute_amd.so
