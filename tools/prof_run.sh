#!/bin/bash
# usage: tools/prof_run.sh <tag> <prof_case args...>   (run on the GPU box)
# kernel-trace pass + four PMC passes (TCC FETCH_SIZE / WRITE_SIZE each in their own pass), summaries into gpurun_out/prof/<tag>/
set -u
tag=$1; shift
out=gpurun_out/prof/$tag
mkdir -p $out
export TMPDIR=/tmp
rocprofv3 -f csv --kernel-trace --stats -d $out/trace -o t -- python tools/prof_case.py "$@" > $out/trace.log 2>&1
rocprofv3 -f csv --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $out/pmc1 -o p -- python tools/prof_case.py "$@" > $out/pmc1.log 2>&1
rocprofv3 -f csv --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $out/pmc2 -o p -- python tools/prof_case.py "$@" > $out/pmc2.log 2>&1
rocprofv3 -f csv --kernel-trace --pmc FETCH_SIZE -d $out/pmc3 -o p -- python tools/prof_case.py "$@" > $out/pmc3.log 2>&1
rocprofv3 -f csv --kernel-trace --pmc WRITE_SIZE -d $out/pmc4 -o p -- python tools/prof_case.py "$@" > $out/pmc4.log 2>&1
python tools/prof_summary.py $out
