#!/bin/bash
# round 5, GPU call 21: K = 3584 decode keys measured again with the lean kernel's 7-piece shape, the whole GPU suite, bench lines
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m flute_amd.tune --retune --shapes '4096,3584;8192,3584;2048,3584;14336,3584' --ms 1,2,4 --bits 4 2>&1 | tail -5
cp flute_amd/data/gfx950_tuned.json gpurun_out/gfx950_tuned.json
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_20.json 2> gpurun_out/bench_20.err; tail -c 3000 gpurun_out/bench_20.json
python bench.py --steps 2000 --warmup 50 > gpurun_out/bench_2000.json 2> gpurun_out/bench_2000.err; tail -c 1500 gpurun_out/bench_2000.json
