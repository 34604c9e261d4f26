#!/bin/bash
# round 5, third GPU call: the lean kernel through the library - parity, forced against automatic plans, the bench at 20 / 2000 steps
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r05
timeout 300 python -m pytest tests/test_qgemm_gpu.py tests/test_abi.py -x -q -m gpu -k "lean or decode_plan or persistent or golden or abi" > gpurun_out/r05/pytest_lean.log 2>&1
tail -3 gpurun_out/r05/pytest_lean.log
rm -f gpurun_out/r05/time_cases_lean.jsonl
C=""
for s in "4096,4096" "11008,4096" "14336,4096" "6144,4096" "1024,4096" "28672,4096" "4096,8192" "8192,8192" "1024,8192" "3584,8192" "4096,2048" "8192,2048"; do
  C="$C;4,1,$s,f16;4,1,$s,f16,one_shot=4"
done
C="$C;4,1,4096,4096,bf16;4,1,4096,4096,bf16,one_shot=4;4,1,4096,8192,bf16;4,1,4096,8192,bf16,one_shot=4"
timeout 400 python tools/time_cases.py "${C:1}" --steps 300 --tag lean --out gpurun_out/r05/time_cases_lean.jsonl > gpurun_out/r05/time_cases_lean.log 2>&1
cat gpurun_out/r05/time_cases_lean.log | cut -c1-260
timeout 200 python bench.py --steps 20 --warmup 5 > gpurun_out/r05/bench_steps20_run1.json 2> gpurun_out/r05/bench_steps20_run1.err
timeout 200 python bench.py --steps 2000 --warmup 50 --no-extras --no-cpu > gpurun_out/r05/bench_steps2000_run1.json 2> gpurun_out/r05/bench_steps2000_run1.err
python - <<'PY'
import json
for f in ("gpurun_out/r05/bench_steps20_run1.json", "gpurun_out/r05/bench_steps2000_run1.json"):
    try:
        d = json.loads(open(f).readline())
        print(f, d["value"], d["ms_per_step"], d["roofline"]["kernel_us"], d["roofline"]["kernel_us_hip_events"], d["roofline"]["kernel_us_clock"], d.get("m256"), d["eager_us_per_step"])
    except Exception as e:
        print(f, "ERR", e)
PY
