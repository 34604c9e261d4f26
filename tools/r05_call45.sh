#!/bin/bash
# round 5, GPU call 45: K = 8192 layers of one round of workgroups (the TP-8 shard of configs[3] among them): the lean kernel's (8, 2, 8) shape forced against the automatic plan
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
rm -f gpurun_out/time_cases.jsonl
C=""
for N in 3584 4096 2048 1024; do for M in 1 2; do C="$C;4,$M,$N,8192,f16;4,$M,$N,8192,f16,family=0,one_shot=4"; done; done
timeout 300 python tools/time_cases.py "${C:1}" --steps 400 --tag k8192 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(j['M'], j['N'], j['K'], j['ovr'] or 'auto', j['tid'], j['us'], j['plan']['one_shot'], j['plan']['waves'], j['plan']['kw'], j['plan']['grid'])"
