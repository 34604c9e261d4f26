#!/bin/bash
# rocprofv3 passes over ONE case of tools/prof_case.py (run on the GPU box): kernel trace + stats, then SQ / TCC counter
# passes, each in its own run (gpurun refuses --pmc combined with the api trace domains).  Usage:
#   tools/prof_case.sh <name> --M 256 --N 4096 --K 4096 --ovr 6,-1,-1,-1,4   -> gpurun_out/prof/<name>/summary.txt
set -u
cd "$(dirname "$0")/.."
name=$1; shift
out=gpurun_out/prof/$name
mkdir -p $out
export TMPDIR=/tmp
CMD="python tools/prof_case.py $*"
echo "profiling: $CMD" > $out/cmd.txt
rocprofv3 -f csv --kernel-trace --stats -d $out/trace -o t -- $CMD > $out/trace.log 2>&1
rocprofv3 -f csv --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM -d $out/pmc1 -o p -- $CMD > $out/pmc1.log 2>&1
rocprofv3 -f csv --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $out/pmc2 -o p -- $CMD > $out/pmc2.log 2>&1
rocprofv3 -f csv --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d $out/pmc3 -o p -- $CMD > $out/pmc3.log 2>&1
rocprofv3 -f csv --kernel-trace --pmc WRITE_SIZE SQ_INST_CYCLES_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_MFMA SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_VMEM_TA_ADDR_FIFO_FULL -d $out/pmc4 -o p -- $CMD > $out/pmc4.log 2>&1
python tools/prof_summary.py $out > /dev/null
cat $out/summary.txt | grep -v "^STATS" | grep -i "qgem\|splitk" 
