#!/bin/bash
# rocprofv3 passes over the bench command (run on the GPU box): kernel trace + stats, then FETCH_SIZE and
# WRITE_SIZE in their own PMC passes (MI355X_MICROARCH.md: TCC slots; FETCH_SIZE x2 correction on gfx950).
# Writes gpurun_out/prof/bench/summary.txt and gpurun_out/prof/bench/r06_bench_traffic.json - the file bench.py
# reads `roofline.traffic` from (copy it to profiles/; it carries the plan it was measured on and bench.py
# drops it the moment the live plan differs).
set -u
cd "$(dirname "$0")/.."          # the bench command and gpurun_out/ are relative to the repo root
out=gpurun_out/prof/bench
mkdir -p $out
export TMPDIR=/tmp
# the tuner's pick first (untraced), then the traced runs launch that plan only
python bench.py --steps 300 --warmup 20 --no-extras --no-cpu > $out/untraced.json
TID=$(python -c "import json,sys; print(json.loads(open('$out/untraced.json').readline())['config']['template_id'])")
CMD="python bench.py --steps 300 --warmup 20 --no-extras --no-cpu --template-id $TID"
echo "profiling: $CMD"
rocprofv3 -f csv --kernel-trace --stats -d $out/trace -o t -- $CMD > $out/trace.log 2>&1
rocprofv3 -f csv --kernel-trace --pmc FETCH_SIZE -d $out/pmc1 -o p -- $CMD > $out/pmc1.log 2>&1
rocprofv3 -f csv --kernel-trace --pmc WRITE_SIZE -d $out/pmc2 -o p -- $CMD > $out/pmc2.log 2>&1
rocprofv3 -f csv --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU -d $out/pmc3 -o p -- $CMD > $out/pmc3.log 2>&1
python tools/prof_summary.py $out
grep -h '^{' $out/trace.log | tail -1 > $out/bench_under_trace.json
python - "$out" "$CMD" <<'PY'
import json, re, sys
out, cmd = sys.argv[1], sys.argv[2]
line = json.loads(open(out + "/untraced.json").readline())
summ = open(out + "/summary.txt").read()
fetch = write = None
kern = None
for l in summ.splitlines():
    if l.startswith("PMC ") and "qgem" in l:
        kern = l[4:].rsplit(": ", 1)[0]
        m = re.search(r"FETCH_SIZE=(\d+)", l)
        if m: fetch = int(m.group(1))
        m = re.search(r"WRITE_SIZE=(\d+)", l)
        if m: write = int(m.group(1))
trace = [l for l in summ.splitlines() if l.startswith("TRACE ") and "qgem" in l]
avg_ns = med_ns = ncalls = None
for l in trace:
    m = re.search(r"n=(\d+) min=\d+ med=(\d+) avg=([\d.]+)", l)
    if m and (ncalls is None or int(m.group(1)) > ncalls):
        ncalls, med_ns, avg_ns = int(m.group(1)), int(m.group(2)), float(m.group(3))
rec = {"source": "tools/prof_bench.sh: rocprofv3 --kernel-trace --stats, then --pmc FETCH_SIZE and --pmc WRITE_SIZE in "
                 "separate passes over `" + cmd + "`",
       "kernel": kern, "workload": line["config"]["workload"], "plan": line["config"]["plan"],
       "FETCH_SIZE_KB_per_launch": fetch, "WRITE_SIZE_KB_per_launch": write,
       "correction": "gfx950: FETCH_SIZE reports half the bytes of a wide coalesced read (MI355X_MICROARCH.md, HBM): doubled",
       "hbm_bytes_per_launch": None if fetch is None else (2 * fetch + (write or 0)) * 1024,
       "kernel_trace": trace, "kernel_trace_calls": ncalls, "kernel_us_rocprof_avg": None if avg_ns is None else round(avg_ns / 1e3, 4),
       "kernel_us_rocprof_median": None if med_ns is None else round(med_ns / 1e3, 4),
       "bench_untraced_ms_per_step": line["ms_per_step"]}
json.dump(rec, open(out + "/r06_bench_traffic.json", "w"), indent=1)
print(json.dumps(rec, indent=1))
PY
