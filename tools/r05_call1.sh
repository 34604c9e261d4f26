#!/bin/bash
# round 5, first GPU call: the lean decode kernel in the lab (timing + stamps), the graph-replay probe, the widened GPU suite
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r05
timeout 150 tools/ubench/oneshot_lab 4096 4096 64 a fast > gpurun_out/r05/lab_fast_run1.jsonl 2>&1
timeout 150 tools/ubench/oneshot_lab_stamps 4096 4096 64 a fast > gpurun_out/r05/lab_fast_stamps_run1.jsonl 2>&1
timeout 100 tools/ubench/oneshot_lab 11008 4096 64 b fast >> gpurun_out/r05/lab_fast_run1.jsonl 2>&1
timeout 100 tools/ubench/oneshot_lab 4096 4096 128 c fast >> gpurun_out/r05/lab_fast_run1.jsonl 2>&1
timeout 60 tools/ubench/oneshot_lab 4096 4096 64 d floors >> gpurun_out/r05/lab_fast_run1.jsonl 2>&1
timeout 200 python tools/graph_probe.py > gpurun_out/r05/graph_probe.json 2> gpurun_out/r05/graph_probe.err
timeout 1100 python -m pytest tests -m gpu -x -q --durations=12 > gpurun_out/r05/pytest_gpu_run1.log 2>&1
tail -5 gpurun_out/r05/pytest_gpu_run1.log
grep -h '"variant"' gpurun_out/r05/lab_fast_run1.jsonl | cut -c1-220
