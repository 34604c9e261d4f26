#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r05
timeout 600 python -m pytest tests/test_qgemm_gpu.py -x -q -m gpu -k "lean_mfma" > gpurun_out/r05/pytest_fastm3.log 2>&1
tail -5 gpurun_out/r05/pytest_fastm3.log | cut -c1-250
rm -f gpurun_out/r05/time_cases_fastm3.jsonl
C=""
for s in "4096,4096" "4096,2048" "2048,4096"; do
  for m in 5 8 16; do
    C="$C;4,$m,$s,f16,family=7,waves=8;4,$m,$s,f16,family=7,waves=12"
  done
done
timeout 900 python tools/time_cases.py "${C:1}" --steps 300 --tag fastm3 --out gpurun_out/r05/time_cases_fastm3.jsonl > gpurun_out/r05/time_cases_fastm3.log 2>&1
python - <<'PY'
import json
for l in open("gpurun_out/r05/time_cases_fastm3.jsonl"):
    d = json.loads(l)
    print(d["M"], d["N"], d["K"], d["dtype"], d["ovr"], d["tid"], d["us"], "fam", d["plan"]["family"], d["plan"]["waves"], d["plan"]["grid"])
PY
FLUTE_AMD_LIB=$PWD/flute_amd/csrc/libflute_amd_stamps.so timeout 300 python tools/stamps_fast.py "16,4096,4096,family=7,waves=12;5,4096,4096,family=7,waves=12" > gpurun_out/r05/stamps_fastm_run3.jsonl 2> gpurun_out/r05/stamps_fastm_run3.err
cat gpurun_out/r05/stamps_fastm_run3.jsonl
