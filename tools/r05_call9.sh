#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r05
cp flute_amd/data/gfx950_tuned.json gpurun_out/r05/tuned_retune_m2_m4.json
timeout 700 python -m flute_amd.tune --shapes "1024,4096;3584,4096;4096,4096;4608,4096;6144,4096;8192,4096;11008,4096;14336,4096;16384,4096;28672,4096;4608,2048;8192,2048;16384,2048" \
    --ms 2,4 --bits 4 --groups 64,128 --retune --rep 120 --budget-s 600 --out gpurun_out/r05/tuned_retune_m2_m4.json > gpurun_out/r05/retune_m2_m4.log 2>&1
tail -1 gpurun_out/r05/retune_m2_m4.log
cp gpurun_out/r05/tuned_retune_m2_m4.json flute_amd/data/gfx950_tuned.json
timeout 1100 python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/r05/pytest_gpu_run2.log 2>&1
tail -14 gpurun_out/r05/pytest_gpu_run2.log
