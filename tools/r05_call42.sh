#!/bin/bash
# round 5, GPU call 42 (final library): the whole GPU suite, smoke(), fuzz sweeps, the bench lines
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_final.log 2>&1; tail -3 gpurun_out/pytest_gpu_final.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
FUZZ_OUT=gpurun_out/r05_gpu_fuzz.json timeout 500 python tools/gpu_fuzz.py 24000 505 2>&1 | grep "^fuzz"
FUZZ_DECODE=3 FUZZ_OUT=gpurun_out/r05_gpu_fuzz_lean_shapes.json timeout 300 python tools/gpu_fuzz.py 16000 506 2>&1 | grep "^fuzz"
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_20.json 2> gpurun_out/bench_20.err
python bench.py --steps 2000 --warmup 50 > gpurun_out/bench_2000.json 2> gpurun_out/bench_2000.err
for f in bench_20 bench_2000; do python - $f <<'PY'
import json, sys
j = json.loads(open("gpurun_out/" + sys.argv[1] + ".json").read().strip().splitlines()[-1]); r = j["roofline"]
print(sys.argv[1], j["value"], j["unit"], "us", r.get("kernel_us"), "events", r.get("kernel_us_hip_events"), "m256", j.get("m256", {}).get("us"))
PY
done
