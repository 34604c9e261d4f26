#!/bin/bash
# round 5, GPU call 23: the decode buckets of the tuned table measured again (the kernels behind them changed this round)
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 1000 python -m flute_amd.tune --retune --shapes supported --ms 1,2,4 --budget-s 800 2>&1 | tail -3
cp flute_amd/data/gfx950_tuned.json gpurun_out/gfx950_tuned.json
