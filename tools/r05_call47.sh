#!/bin/bash
# round 5, GPU call 47 (final library): K = 2048 two-row keys tuned again (the lean kernel up to two rounds), then the whole GPU suite, smoke(), the bench lines
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
rm -f gpurun_out/time_cases.jsonl
timeout 200 python -m flute_amd.tune --retune --shapes '8192,2048;4608,2048;16384,2048' --ms 2 --bits 4 2>&1 | tail -1
cp flute_amd/data/gfx950_tuned.json gpurun_out/gfx950_tuned.json
timeout 200 python tools/time_cases.py "4,2,8192,2048,f16;4,2,4608,2048,f16;4,2,6144,2048,f16;4,2,16384,2048,f16" --steps 400 --tag k2048 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(j['M'], j['N'], j['K'], j['dtype'], j['tid'], j['us'], j['plan']['one_shot'], j['plan']['waves'], j['plan']['kw'], j['plan']['grid'])"
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_final.log 2>&1; tail -3 gpurun_out/pytest_gpu_final.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_20.json 2> gpurun_out/bench_20.err
python bench.py --steps 2000 --warmup 50 > gpurun_out/bench_2000.json 2> gpurun_out/bench_2000.err
for f in bench_20 bench_2000; do python - $f <<'PY'
import json, sys
j = json.loads(open("gpurun_out/" + sys.argv[1] + ".json").read().strip().splitlines()[-1]); r = j["roofline"]
print(sys.argv[1], j["value"], j["unit"], "us", r.get("kernel_us"), "events", r.get("kernel_us_hip_events"), "m256", j.get("m256", {}).get("us"))
PY
done
