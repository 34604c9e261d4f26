#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r05
timeout 600 python -m pytest tests/test_qgemm_gpu.py -x -q -m gpu -k "lean_mfma" > gpurun_out/r05/pytest_fastm.log 2>&1
tail -15 gpurun_out/r05/pytest_fastm.log | cut -c1-250
rm -f gpurun_out/r05/time_cases_fastm.jsonl
C=""
for s in "4096,4096" "4096,2048" "8192,4096" "11008,4096" "2048,4096"; do
  for m in 3 4 5 8 16; do
    C="$C;4,$m,$s,f16;4,$m,$s,f16,family=7"
  done
done
C="$C;4,16,4096,4096,bf16;4,16,4096,4096,bf16,family=7"
timeout 900 python tools/time_cases.py "${C:1}" --steps 300 --tag fastm --out gpurun_out/r05/time_cases_fastm.jsonl > gpurun_out/r05/time_cases_fastm.log 2>&1
python - <<'PY'
import json
for l in open("gpurun_out/r05/time_cases_fastm.jsonl"):
    d = json.loads(l)
    print(d["M"], d["N"], d["K"], d["dtype"], d["ovr"], d["tid"], d["us"], "fam", d["plan"]["family"], "os", d["plan"]["one_shot"], d["plan"]["waves"], d["plan"]["grid"])
PY
grep -c error gpurun_out/r05/time_cases_fastm.log
