#!/bin/bash
# rocprofv3 passes over the bench command (run on the GPU box): kernel trace + stats, then FETCH_SIZE and
# WRITE_SIZE in their own PMC passes (MI355X_MICROARCH.md: TCC slots; FETCH_SIZE x2 correction on gfx950).
set -u
cd "$(dirname "$0")/.."          # the bench command and gpurun_out/ are relative to the repo root
out=gpurun_out/prof/bench
mkdir -p $out
export TMPDIR=/tmp
# the tuner's pick first (untraced), then the traced runs launch that plan only
TID=$(python bench.py --steps 300 --warmup 20 --no-extras --no-cpu | python -c "import json,sys; print(json.loads(sys.stdin.readline())['config']['template_id'])")
CMD="python bench.py --steps 300 --warmup 20 --no-extras --no-cpu --template-id $TID"
echo "profiling: $CMD"
rocprofv3 -f csv --kernel-trace --stats -d $out/trace -o t -- $CMD > $out/trace.log 2>&1
rocprofv3 -f csv --kernel-trace --pmc FETCH_SIZE -d $out/pmc1 -o p -- $CMD > $out/pmc1.log 2>&1
rocprofv3 -f csv --kernel-trace --pmc WRITE_SIZE -d $out/pmc2 -o p -- $CMD > $out/pmc2.log 2>&1
python tools/prof_summary.py $out
grep -h '^{' $out/trace.log | tail -1 > $out/bench_under_trace.json
