#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r05
timeout 600 bash tools/prof_bench.sh > gpurun_out/r05/prof_bench2.log 2>&1
tail -12 gpurun_out/r05/prof_bench2.log | cut -c1-250
grep "^PMC\|^TRACE" gpurun_out/prof/bench/summary.txt | grep -i "qgem" | cut -c1-400
timeout 600 bash tools/prof_m256.sh > gpurun_out/r05/prof_m256.log 2>&1
tail -3 gpurun_out/r05/prof_m256.log | cut -c1-1200
grep "^PMC\|^TRACE" gpurun_out/prof/m256/summary.txt | grep -i "qgem\|splitk" | cut -c1-400
timeout 400 bash tools/prof_case.sh fastm_m16 --M 16 --N 4096 --K 4096 --tid 16 --steps 60 > gpurun_out/r05/prof_fastm.log 2>&1
grep "^PMC\|^TRACE" gpurun_out/prof/fastm_m16/summary.txt | grep -i "qgem" | cut -c1-400
