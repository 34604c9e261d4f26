#!/bin/bash
# Round 6's evidence run (on the GPU box, from the repo root): GPU suite, bench lines at 2000 and 20 steps, the pure-read floor of the
# headline's launch shape, rocprofv3 trace + PMC passes of the bench command and of the M = 256 / M = 16 cases.
mkdir -p gpurun_out/r06
python -m pytest tests -q -m gpu 2>&1 | tail -5 > gpurun_out/r06/pytest_gpu.log
python bench.py > gpurun_out/r06/bench_2000steps.json 2> gpurun_out/r06/bench_2000steps.err
python bench.py --steps 20 --warmup 5 > gpurun_out/r06/bench_20steps.json 2> gpurun_out/r06/bench_20steps.err
timeout 200 tools/ubench/oneshot_lab 4096 4096 64 r06 floors > gpurun_out/r06/calibration_floors.jsonl 2>&1
bash tools/prof_bench.sh > gpurun_out/r06/prof_bench.log 2>&1
bash tools/prof_m256.sh > gpurun_out/r06/prof_m256.log 2>&1
bash tools/prof_case.sh fastm_m16_11008 --M 16 --N 11008 --K 4096 --steps 100 > gpurun_out/r06/prof_fastm_m16_11008.log 2>&1
bash tools/prof_case.sh fastm_m16_4096 --M 16 --N 4096 --K 4096 --steps 100 > gpurun_out/r06/prof_fastm_m16_4096.log 2>&1
bash tools/prof_case.sh persistm_m4_8192x28672 --M 4 --N 8192 --K 28672 --steps 60 > gpurun_out/r06/prof_persistm_m4_8192x28672.log 2>&1
bash tools/prof_case.sh persistm_m16_10240x8192 --M 16 --N 10240 --K 8192 --steps 60 > gpurun_out/r06/prof_persistm_m16_10240x8192.log 2>&1
for n in bench m256 fastm_m16_11008 fastm_m16_4096 persistm_m4_8192x28672 persistm_m16_10240x8192; do cp gpurun_out/prof/$n/summary.txt gpurun_out/r06/rocprof_${n}_summary.txt 2>/dev/null; done
cp gpurun_out/prof/bench/r06_bench_traffic.json gpurun_out/prof/m256/r06_m256_pmc.json gpurun_out/r06/ 2>/dev/null
cat gpurun_out/r06/pytest_gpu.log; head -c 600 gpurun_out/r06/bench_20steps.json; echo; tail -3 gpurun_out/r06/calibration_floors.jsonl
