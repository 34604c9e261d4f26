#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r05
timeout 200 tools/ubench/oneshot_lab 4096 4096 64 a fast > gpurun_out/r05/lab_fast_run2.jsonl 2>&1
timeout 200 tools/ubench/oneshot_lab_stamps 4096 4096 64 a fast > gpurun_out/r05/lab_fast_stamps_run2.jsonl 2>&1
timeout 100 tools/ubench/oneshot_lab 11008 4096 64 b fast >> gpurun_out/r05/lab_fast_run2.jsonl 2>&1
timeout 100 tools/ubench/oneshot_lab 4096 8192 128 c fast >> gpurun_out/r05/lab_fast_run2.jsonl 2>&1
grep -h '"variant"' gpurun_out/r05/lab_fast_run2.jsonl | cut -c1-200
