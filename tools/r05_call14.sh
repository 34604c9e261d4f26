#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r05
cp flute_amd/data/gfx950_tuned.json gpurun_out/r05/tuned_retune_m16.json
timeout 300 python -m flute_amd.tune --shapes "3584,4096;4096,4096;2048,4096;4096,2048" --ms 16 --bits 4 --groups 64,128 --retune --rep 120 --budget-s 250 --out gpurun_out/r05/tuned_retune_m16.json > gpurun_out/r05/retune_m16.log 2>&1
tail -1 gpurun_out/r05/retune_m16.log
cp gpurun_out/r05/tuned_retune_m16.json flute_amd/data/gfx950_tuned.json
timeout 1100 python -m pytest tests -m gpu -x -q --durations=5 > gpurun_out/r05/pytest_gpu_run3.log 2>&1
tail -9 gpurun_out/r05/pytest_gpu_run3.log
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r05/bench_steps20_run3.json 2> gpurun_out/r05/bench_steps20_run3.err
timeout 300 python bench.py --steps 2000 --warmup 50 > gpurun_out/r05/bench_steps2000_run3.json 2> gpurun_out/r05/bench_steps2000_run3.err
python - <<'PY'
import json
for f in ("gpurun_out/r05/bench_steps20_run3.json", "gpurun_out/r05/bench_steps2000_run3.json"):
    try:
        d = json.loads(open(f).readline())
        print(f, d["value"], d["ms_per_step"], d["roofline"]["kernel_us_hip_events"], d["config"]["template_id"], d["config"]["plan"]["one_shot"], d["m256"]["us"] if d.get("m256") else None, d["eager_us_per_step"])
        for e in d["extras"]: print("   ", e["workload"][:75], e["us"], e.get("template_id"), e.get("speedup_vs_torch_mm"))
        print("   tp", d.get("tp_mlp_pair", {}).get("kernels_us"))
    except Exception as e:
        print(f, "ERR", e)
PY
