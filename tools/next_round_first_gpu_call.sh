#!/bin/bash
# What the measurements at the end of round 4 ask for first (run on the GPU box through gpurun, about 8 GPU-minutes; results
# come back under gpurun_out/):
#   1. the tuner's challenge pass on the M = 64 bucket (33 <= M <= 64) of the 4-bit keys: the table's ids of that bucket were
#      timed on the per-wave MFMA kernel before the split-K block kernel reached M < 128; the automatic ids (0 / 16: TileP 64 /
#      32, Stages 2, SMs_Multiple 1) now take 64-row tiles x K slices where the planner's model says so - measured 12 .. 19 %
#      under the incumbent on 8192^2 and 14336 x 4096 (profiles/r04/splitk_64_row_tiles_below_m128.json); five keys were moved
#      by hand from those timings, the rest needs this pass.  Then the same for the 2-bit M = 128 bucket (65 <= M <= 128).
#   2. the planner regret sweep at the batch sizes BETWEEN the ones round 4 swept (its candidate list now offers 3-bit blocks x
#      K slices): every case over 10 % is a threshold to fix.
# Copy gpurun_out/tuned_challenged.json over flute_amd/data/gfx950_tuned.json afterwards (tests/test_host.py checks the table).
#   3. (second call) the 3-bit block kernel's whole-line PLANE pieces (qgemm_block3.h, FLUTE_B3_LINE_PLANES = 1 / 2; lane mapping
#      modelled in tests/test_splitk_layout.py, never run): build HERE, before the call,
#          make -C flute_amd/csrc -j OBJDIR=build_lp1 LIB=libflute_amd_lp1.so TORCHLIB=libflute_amd_torch_lp1.so EXTRA=-DFLUTE_B3_LINE_PLANES=1
#      and on the box (its copy of the tree is scratch) put it in the library's place, then parity + timing:
#          cp flute_amd/csrc/libflute_amd_lp1.so flute_amd/csrc/libflute_amd.so
#          timeout 120 python -m pytest tests/test_qgemm_gpu.py -x -q -m gpu -k block_prefill
#          python tools/time_cases.py "3,4096,4096,4096,bf16;3,1024,28672,8192,bf16;3,1024,4096,4096,bf16;3,256,8192,8192,bf16"
#      (round 4, default build: 127.9 / 466 / 56.3 / 55.9 us).  Keep whichever of 0 / 1 / 2 wins; delete the others.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
cp flute_amd/data/gfx950_tuned.json gpurun_out/tuned_challenged.json
timeout 200 python -m flute_amd.tune --out gpurun_out/tuned_challenged.json --ms 64 --bits 4 --groups 64,128 --challenge 0,16 \
    --rep 20 --budget-s 180 > gpurun_out/challenge_m64_b4.log 2>&1
timeout 120 python -m flute_amd.tune --out gpurun_out/tuned_challenged.json --ms 128 --bits 2 --groups 64 --challenge 0,7,3,15 \
    --rep 20 --budget-s 100 > gpurun_out/challenge_m128_b2.log 2>&1
timeout 280 python tools/regret.py --ms 32,48,96,128,384,512,2048 --budget-s 260 --steps 60 \
    --out gpurun_out/planner_regret_between.json > gpurun_out/regret_between.log 2>&1
tail -2 gpurun_out/challenge_m64_b4.log gpurun_out/challenge_m128_b2.log gpurun_out/regret_between.log
