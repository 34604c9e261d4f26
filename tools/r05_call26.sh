#!/bin/bash
# round 5, GPU call 26: after the planner fixes from call 25's regret sweeps (3-bit blocks x uneven K slices, cheapest K-split candidate,
# digit 3 = no lane sharing + two slabs per wave above M = 16): the 4-bit M = 32 / 64 / 128 buckets tuned again, both regret sweeps again
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 500 python -m flute_amd.tune --retune --shapes supported --ms 32,64,128 --bits 4 --rep 20 --budget-s 420 2>&1 | tail -2
cp flute_amd/data/gfx950_tuned.json gpurun_out/gfx950_tuned.json
timeout 330 python tools/regret.py --ms 32,48,96,128,384,512,2048 --budget-s 300 --steps 60 \
    --out gpurun_out/planner_regret_between_after.json > gpurun_out/regret_between_after.log 2>&1
timeout 300 python tools/regret.py --ms 1,2,4,16,64,256,1024 --budget-s 270 --steps 60 \
    --out gpurun_out/planner_regret_r05_after.json > gpurun_out/regret_r05_after.log 2>&1
tail -n 1 gpurun_out/regret_between_after.log gpurun_out/regret_r05_after.log
timeout 300 python -m pytest tests/test_qgemm_gpu.py -x -q -m gpu -k "block_prefill or seam_under_load or mfma_family or fuzz" 2>&1 | tail -2
