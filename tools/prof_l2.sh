#!/bin/bash
set -u
tag=$1; shift
out=gpurun_out/prof/$tag
mkdir -p $out
export TMPDIR=/tmp
rocprofv3 -f csv --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum -d $out/pmc1 -o p -- python tools/prof_case.py "$@" > $out/pmc1.log 2>&1
rocprofv3 -f csv --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum -d $out/pmc2 -o p -- python tools/prof_case.py "$@" > $out/pmc2.log 2>&1
python tools/prof_summary.py $out
