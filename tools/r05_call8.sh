#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r05
timeout 600 python -m pytest tests/test_qgemm_gpu.py -x -q -m gpu -k "lean or persistent or full_size or decode_plan or fuzz or golden or forced" > gpurun_out/r05/pytest_lean4.log 2>&1
tail -4 gpurun_out/r05/pytest_lean4.log
rm -f gpurun_out/r05/time_cases_rows.jsonl
C=""
for s in "4096,4096" "11008,4096" "8192,4096" "6144,4096" "4096,2048"; do
  for m in 2 3 4; do
    C="$C;4,$m,$s,f16;4,$m,$s,f16,family=0,one_shot=4,waves=4;4,$m,$s,f16,family=0,one_shot=4,waves=8"
  done
done
C="$C;4,1,28672,8192,f16;4,2,28672,8192,f16;4,1,8192,28672,f16;4,1,14336,4096,f16;4,1,28672,4096,f16;4,1,8192,8192,f16;4,1,4096,4096,f16;4,1,11008,4096,f16"
timeout 900 python tools/time_cases.py "${C:1}" --steps 300 --tag rows --out gpurun_out/r05/time_cases_rows.jsonl > gpurun_out/r05/time_cases_rows.log 2>&1
python - <<'PY'
import json
for l in open("gpurun_out/r05/time_cases_rows.jsonl"):
    d = json.loads(l)
    print(d["M"], d["N"], d["K"], d["ovr"], d["tid"], d["us"], "fam", d["plan"]["family"], "os", d["plan"]["one_shot"], d["plan"]["waves"], d["plan"]["kw"], d["plan"]["grid"])
PY
grep -c error gpurun_out/r05/time_cases_rows.log
