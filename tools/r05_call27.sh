#!/bin/bash
# round 5, GPU call 27: after the deep-layer rule (grid K split instead of more lane sharing) and the skinny-block fill rule: the buckets
# whose automatic ids changed plans tuned again, the regret sweep of the swept batch sizes and of the ones between
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 400 python -m flute_amd.tune --retune --shapes supported --ms 4,16,32 --bits 4,2 --rep 20 --budget-s 330 2>&1 | tail -1
cp flute_amd/data/gfx950_tuned.json gpurun_out/gfx950_tuned.json
timeout 200 python tools/regret.py --ms 1,2,4,16,64,256,1024 --budget-s 170 --steps 60 \
    --out gpurun_out/planner_regret_r05_final.json > gpurun_out/regret_r05_final.log 2>&1
timeout 260 python tools/regret.py --ms 32,48,96,128,384,512,2048 --budget-s 240 --steps 60 \
    --out gpurun_out/planner_regret_between_final.json > gpurun_out/regret_between_final.log 2>&1
tail -n 1 gpurun_out/regret_r05_final.log gpurun_out/regret_between_final.log
