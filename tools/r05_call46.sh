#!/bin/bash
# round 5, GPU call 46: the lean kernel's K = 8192 shape for two rows (automatic on layers of >= 80 % of a round): parity, the keys concerned tuned again, timing
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
rm -f gpurun_out/time_cases.jsonl
timeout 600 python -m pytest tests/test_qgemm_gpu.py tests/test_parity_sweep_gpu.py -x -q -m gpu -k "lean or decode_plan or 3584 or fuzz or golden" 2>&1 | tail -2
timeout 200 python -m flute_amd.tune --retune --shapes '3584,8192;4096,8192' --ms 2 --bits 4 2>&1 | tail -1
cp flute_amd/data/gfx950_tuned.json gpurun_out/gfx950_tuned.json
timeout 200 python tools/time_cases.py "4,2,3584,8192,f16;4,2,3584,8192,bf16;4,2,4096,8192,f16;4,1,3584,8192,f16" --steps 400 --tag k8192_auto 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(j['M'], j['N'], j['K'], j['dtype'], j['tid'], j['us'], j['plan']['one_shot'], j['plan']['waves'], j['plan']['kw'], j['plan']['grid'])"
