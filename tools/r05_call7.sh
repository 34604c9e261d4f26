#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r05
cp flute_amd/data/gfx950_tuned.json gpurun_out/r05/tuned_retune_m1.json
timeout 700 python -m flute_amd.tune --shapes "1024,4096;3584,4096;4096,4096;4608,4096;6144,4096;8192,4096;11008,4096;14336,4096;16384,4096;28672,4096;4608,2048;8192,2048;16384,2048" \
    --ms 1 --bits 4 --groups 64,128 --retune --rep 120 --budget-s 600 --out gpurun_out/r05/tuned_retune_m1.json > gpurun_out/r05/retune_m1.log 2>&1
tail -1 gpurun_out/r05/retune_m1.log
cp gpurun_out/r05/tuned_retune_m1.json flute_amd/data/gfx950_tuned.json      # (the box's copy of the tree: the bench below runs on the new table)
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r05/bench_steps20_run2.json 2> gpurun_out/r05/bench_steps20_run2.err
timeout 300 python bench.py --steps 2000 --warmup 50 > gpurun_out/r05/bench_steps2000_run2.json 2> gpurun_out/r05/bench_steps2000_run2.err
python - <<'PY'
import json
for f in ("gpurun_out/r05/bench_steps20_run2.json", "gpurun_out/r05/bench_steps2000_run2.json"):
    try:
        d = json.loads(open(f).readline())
        print(f, d["value"], d["ms_per_step"], d["roofline"]["kernel_us"], d["roofline"]["kernel_us_hip_events"], d["config"]["template_id"], d["config"]["plan"]["one_shot"], d.get("m256", {}).get("us"), d["eager_us_per_step"])
        for e in d["extras"]: print("   ", e["workload"][:70], e["us"], e.get("template_id"))
    except Exception as e:
        print(f, "ERR", e)
PY
timeout 600 bash tools/prof_bench.sh > gpurun_out/r05/prof_bench.log 2>&1
tail -30 gpurun_out/r05/prof_bench.log
