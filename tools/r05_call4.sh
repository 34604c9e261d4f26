#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r05
timeout 300 python -m pytest tests/test_qgemm_gpu.py -x -q -m gpu -k "lean" > gpurun_out/r05/pytest_lean2.log 2>&1
tail -3 gpurun_out/r05/pytest_lean2.log
rm -f gpurun_out/r05/time_cases_lean2.jsonl
C=""
for s in "4096,4096" "5120,4096" "6144,4096" "8192,4096" "11008,4096" "14336,4096" "28672,4096" "2048,4096" "3072,4096"; do
  C="$C;4,1,$s,f16,one_shot=3;4,1,$s,f16,one_shot=4,waves=4;4,1,$s,f16,one_shot=4,waves=8;4,1,$s,f16,one_shot=1"
done
timeout 600 python tools/time_cases.py "${C:1}" --steps 300 --tag lean2 --out gpurun_out/r05/time_cases_lean2.jsonl > gpurun_out/r05/time_cases_lean2.log 2>&1
python - <<'PY'
import json
for l in open("gpurun_out/r05/time_cases_lean2.jsonl"):
    d = json.loads(l)
    print(d["N"], d["K"], d["ovr"], d["us"], d["plan"]["one_shot"], d["plan"]["waves"], d["plan"]["kw"], d["plan"]["grid"])
PY
