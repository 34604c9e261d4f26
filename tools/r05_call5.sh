#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r05
rm -f gpurun_out/r05/lab_fast_run3.jsonl
timeout 200 tools/ubench/oneshot_lab 4096 4096 64 a fast >> gpurun_out/r05/lab_fast_run3.jsonl 2>&1
timeout 200 tools/ubench/oneshot_lab 4096 4096 64 a2 fast >> gpurun_out/r05/lab_fast_run3.jsonl 2>&1
timeout 100 tools/ubench/oneshot_lab 11008 4096 64 b fast >> gpurun_out/r05/lab_fast_run3.jsonl 2>&1
timeout 100 tools/ubench/oneshot_lab 4096 8192 64 c fast >> gpurun_out/r05/lab_fast_run3.jsonl 2>&1
timeout 100 tools/ubench/oneshot_lab 8192 2048 64 d fast >> gpurun_out/r05/lab_fast_run3.jsonl 2>&1
timeout 200 tools/ubench/oneshot_lab_stamps 4096 4096 64 a fast > gpurun_out/r05/lab_fast_stamps_run3.jsonl 2>&1
python - <<'PY'
import json
for l in open("gpurun_out/r05/lab_fast_run3.jsonl"):
    try: d = json.loads(l)
    except Exception: print(l[:150]); continue
    if "variant" in d: print(d.get("tag"), d["variant"], d.get("us"), d.get("rel_err"), d.get("nbad"))
    elif "error" in d: print(d)
PY
