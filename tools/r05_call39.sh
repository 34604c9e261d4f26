#!/bin/bash
# round 5, GPU calls 39 and 41: bf16 scale multiply of the MFMA kernels as two v_dot2_f32_bf16 + one v_cvt_pk_bf16_f32 per word (new) against unpack / multiply / convert
# (old), alternating on one box; parity of the new one first (one-hot rows bit-exact against round_bf16(lut * s))
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
rm -f gpurun_out/time_cases.jsonl
timeout 900 python -m pytest tests/test_qgemm_gpu.py -x -q -m gpu -k "block_prefill or splitk or mfma or skinny or golden or random_vs_oracle or group_sizes or full_size or fuzz or seam" 2>&1 | tail -4
C="4,4096,4096,4096,bf16;4,2048,4096,4096,bf16;4,1024,4096,4096,bf16;4,256,11008,4096,bf16;3,4096,4096,4096,bf16;3,1024,4096,4096,bf16;3,256,8192,8192,bf16;3,1024,28672,8192,bf16;4,64,8192,8192,bf16;4,16,28672,8192,bf16"
cp flute_amd/csrc/libflute_amd.so /tmp/new.so
for rep in 1 2; do
  for v in old new; do
    if [ $v = old ]; then cp flute_amd/csrc/libflute_amd_oldbf.so flute_amd/csrc/libflute_amd.so; else cp /tmp/new.so flute_amd/csrc/libflute_amd.so; fi
    timeout 300 python tools/time_cases.py "$C" --steps 100 --tag $v$rep 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(j['tag'], j['bits'], j['M'], j['N'], j['K'], j['dtype'], j['us'], j['plan']['family'], j['plan']['m_block'], j['plan']['splitk'])"
  done
done
cp /tmp/new.so flute_amd/csrc/libflute_amd.so
