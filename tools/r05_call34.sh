#!/bin/bash
# round 5, GPU call 34: A/B on one box - split-K block kernel with the per-half-step preparation in front of the barrier (new) against the committed kernel (old), alternating
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
rm -f gpurun_out/time_cases.jsonl
C="4,256,4096,4096,f16;4,256,11008,4096,f16;4,1024,4096,4096,f16;4,64,8192,8192,f16;4,96,14336,4096,f16;4,256,8192,8192,f16;4,512,4096,4096,f16;4,128,8192,8192,f16"
cp flute_amd/csrc/libflute_amd.so /tmp/new.so
for rep in 1 2; do
  for v in old new; do
    if [ $v = old ]; then cp flute_amd/csrc/libflute_amd_oldsk.so flute_amd/csrc/libflute_amd.so; else cp /tmp/new.so flute_amd/csrc/libflute_amd.so; fi
    timeout 300 python tools/time_cases.py "$C" --steps 400 --tag $v$rep 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(j['tag'], j['M'], j['N'], j['K'], j['us'], j['plan']['m_tiles'], j['plan']['splitk'])"
  done
done
cp /tmp/new.so flute_amd/csrc/libflute_amd.so
