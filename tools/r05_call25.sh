#!/bin/bash
# round 5, GPU call 25: (a) the restricted in-launch seam build under test; (b) M = 64 on 8192 x 28672 (round 4's one regret case over
# 10 %) with the split-K block kernel forced; (c) the tuner's challenge pass on the M = 64 bucket (4 bits, ids 0 / 16) and the M = 128
# bucket (2 bits); (d) the regret sweep at the batch sizes between the swept ones and at the swept ones (the kernels changed)
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
rm -f gpurun_out/time_cases.jsonl
timeout 600 python -m pytest tests/test_qgemm_gpu.py -x -q -m gpu -k "block_prefill or seam_under_load or splitk" 2>&1 | tail -3
timeout 200 python tools/time_cases.py "4,64,28672,8192,f16;4,64,28672,8192,f16,family=6,m_tiles=4;4,64,28672,8192,f16,family=6,m_tiles=8;4,48,28672,8192,f16;4,48,28672,8192,f16,family=6,m_tiles=4;4,33,28672,8192,f16;4,33,28672,8192,f16,family=6,m_tiles=4" --tag m64 2>&1 | cut -c1-330
cp flute_amd/data/gfx950_tuned.json gpurun_out/tuned_challenged.json
timeout 260 python -m flute_amd.tune --out gpurun_out/tuned_challenged.json --ms 64 --bits 4 --groups 64,128 --challenge 0,16 \
    --rep 20 --budget-s 240 > gpurun_out/challenge_m64_b4.log 2>&1
timeout 160 python -m flute_amd.tune --out gpurun_out/tuned_challenged.json --ms 128 --bits 2 --groups 64 --challenge 0,7,3,15 \
    --rep 20 --budget-s 140 > gpurun_out/challenge_m128_b2.log 2>&1
cp gpurun_out/tuned_challenged.json flute_amd/data/gfx950_tuned.json
timeout 330 python tools/regret.py --ms 32,48,96,128,384,512,2048 --budget-s 300 --steps 60 \
    --out gpurun_out/planner_regret_between.json > gpurun_out/regret_between.log 2>&1
timeout 400 python tools/regret.py --ms 1,2,4,16,64,256,1024 --budget-s 370 --steps 60 \
    --out gpurun_out/planner_regret_r05.json > gpurun_out/regret_r05.log 2>&1
tail -3 gpurun_out/challenge_m64_b4.log gpurun_out/challenge_m128_b2.log gpurun_out/regret_between.log gpurun_out/regret_r05.log
