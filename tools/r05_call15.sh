#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r05
timeout 1100 python -m pytest tests -m gpu -x -q --durations=5 > gpurun_out/r05/pytest_gpu_run4.log 2>&1
tail -12 gpurun_out/r05/pytest_gpu_run4.log | cut -c1-300
rm -f gpurun_out/r05/time_cases_w3.jsonl
timeout 600 python tools/time_cases.py "3,1,8192,8192,bf16;3,1,28672,8192,bf16;3,1,8192,28672,bf16;3,2,28672,8192,bf16;3,1,4096,4096,bf16;3,1,14336,4096,f16;4,1,4096,8192,f16;4,1,8192,8192,f16;4,2,8192,8192,f16;4,4,8192,4096,f16,family=0;2,1,8192,8192,f16" --steps 300 --tag w3 --out gpurun_out/r05/time_cases_w3.jsonl > gpurun_out/r05/time_cases_w3.log 2>&1
python - <<'PY'
import json
for l in open("gpurun_out/r05/time_cases_w3.jsonl"):
    d = json.loads(l)
    print(d["bits"], d["M"], d["N"], d["K"], d["dtype"], d["ovr"], d["tid"], d["us"], "fam", d["plan"]["family"], "os", d["plan"]["one_shot"], d["plan"]["waves"], d["plan"]["grid"])
PY
