#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r05
timeout 300 python -m pytest tests/test_qgemm_gpu.py -x -q -m gpu -k "lean or golden or decode_plan" > gpurun_out/r05/pytest_lean3.log 2>&1
tail -3 gpurun_out/r05/pytest_lean3.log
rm -f gpurun_out/r05/lab_fast_run4.jsonl
timeout 200 tools/ubench/oneshot_lab 4096 4096 64 a fast >> gpurun_out/r05/lab_fast_run4.jsonl 2>&1
timeout 100 tools/ubench/oneshot_lab 11008 4096 64 b fast >> gpurun_out/r05/lab_fast_run4.jsonl 2>&1
timeout 200 tools/ubench/oneshot_lab_stamps 4096 4096 64 a fast > gpurun_out/r05/lab_fast_stamps_run4.jsonl 2>&1
python - <<'PY'
import json
for l in open("gpurun_out/r05/lab_fast_run4.jsonl"):
    try: d = json.loads(l)
    except Exception: continue
    if "variant" in d and ("fast" in d["variant"] or "pipe" in d["variant"]): print(d.get("tag"), d["variant"], d.get("us"), d.get("rel_err"), d.get("nbad"))
PY
rm -f gpurun_out/r05/time_cases_lean3.jsonl
C=""
for s in "16384,2048" "8192,2048" "4608,2048" "4096,2048"; do
  C="$C;4,1,$s,f16,one_shot=3;4,1,$s,f16,one_shot=4;4,1,$s,f16,one_shot=1"
done
C="$C;4,1,4096,4096,f16;4,1,11008,4096,f16;4,1,8192,4096,f16;4,1,14336,4096,f16"
timeout 300 python tools/time_cases.py "${C:1}" --steps 300 --tag lean3 --out gpurun_out/r05/time_cases_lean3.jsonl > gpurun_out/r05/time_cases_lean3.log 2>&1
python - <<'PY'
import json
for l in open("gpurun_out/r05/time_cases_lean3.jsonl"):
    d = json.loads(l)
    print(d["N"], d["K"], d["ovr"], d["tid"], d["us"], d["plan"]["one_shot"], d["plan"]["waves"], d["plan"]["kw"], d["plan"]["grid"])
PY
cp flute_amd/data/gfx950_tuned.json gpurun_out/r05/tuned_retune_m1.json
timeout 700 python -m flute_amd.tune --shapes "1024,4096;3584,4096;4096,4096;4608,4096;6144,4096;8192,4096;11008,4096;14336,4096;16384,4096;28672,4096;4608,2048;8192,2048;16384,2048" \
    --ms 1 --bits 4 --groups 64,128 --retune --rep 40 --budget-s 600 --out gpurun_out/r05/tuned_retune_m1.json > gpurun_out/r05/retune_m1.log 2>&1
tail -2 gpurun_out/r05/retune_m1.log
