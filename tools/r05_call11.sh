#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r05
FLUTE_AMD_LIB=$PWD/flute_amd/csrc/libflute_amd_stamps.so timeout 300 python tools/stamps_fast.py "16,4096,4096,family=7;5,4096,4096,family=7;16,4096,2048,family=7;1,4096,4096;4,4096,4096;16,11008,4096,family=7" > gpurun_out/r05/stamps_fastm_run1.jsonl 2> gpurun_out/r05/stamps_fastm_run1.err
cat gpurun_out/r05/stamps_fastm_run1.jsonl
tail -3 gpurun_out/r05/stamps_fastm_run1.err
