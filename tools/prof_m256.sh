#!/bin/bash
# BASELINE.json's second half (TFLOP/s at M = 256, W4G64 4096 x 4096): rocprofv3 kernel trace + PMC passes of the automatic plan under the
# table's id, condensed into gpurun_out/prof/m256/r06_m256_pmc.json - the file bench.py's `m256` block reads traffic / MFMA-busy from
# (copy it to profiles/; it carries the plan it was measured on and bench.py drops it the moment the live plan differs).
set -u
cd "$(dirname "$0")/.."
TID=$(python - <<'PY'
import torch
from flute_amd import tune, utils
d = torch.device("cuda:0")
print(tune._tune(256, 4096, 4096, 4, 64, utils.get_device_num_sms(d), torch.float16, d, num_seeds=1, rep=60))
PY
)
bash tools/prof_case.sh m256 --M 256 --N 4096 --K 4096 --tid $TID --steps 60 > gpurun_out/prof_m256.log 2>&1
python - "$TID" <<'PY'
import json, re, sys, torch
from flute_amd import utils
tid = int(sys.argv[1])
summ = open("gpurun_out/prof/m256/summary.txt").read()
vals = {}
trace = None
for l in summ.splitlines():
    if l.startswith("PMC ") and ("splitk" in l or "qgemm" in l):
        for k, v in re.findall(r"(\w+)=(\d+)", l):
            vals[k] = int(v)
    if l.startswith("TRACE ") and ("splitk" in l or "qgemm" in l) and "reduce" not in l:
        trace = l
m = re.search(r"n=(\d+) min=(\d+) med=(\d+) avg=([\d.]+)", trace or "")
med_ns = int(m.group(3)) if m else None
plan = utils.get_plan(256, 4096, 4096, 4, 64, tid, 256, torch.float16)
busy = vals.get("SQ_VALU_MFMA_BUSY_CYCLES")
rec = {"source": "tools/prof_m256.sh: rocprofv3 --kernel-trace --stats, then four --pmc passes over `python tools/prof_case.py --M 256 --N 4096 --K 4096 --tid %d --steps 60`" % tid,
       "workload": "W4G64 fp16 M=256 K=4096 N=4096", "template_id": tid, "plan": plan, "kernel_trace": trace,
       "kernel_us_rocprof_median": None if med_ns is None else med_ns / 1e3,
       "FETCH_SIZE_KB_per_launch": vals.get("FETCH_SIZE"), "WRITE_SIZE_KB_per_launch": vals.get("WRITE_SIZE"),
       "hbm_bytes_per_launch": None if "FETCH_SIZE" not in vals else (2 * vals["FETCH_SIZE"] + vals.get("WRITE_SIZE", 0)) * 1024,
       "correction": "gfx950: FETCH_SIZE reports half the bytes of a wide coalesced read (MI355X_MICROARCH.md, HBM): doubled",
       "SQ_VALU_MFMA_BUSY_CYCLES": busy, "SQ_INSTS_MFMA": vals.get("SQ_INSTS_MFMA"), "SQ_INSTS_VALU": vals.get("SQ_INSTS_VALU"),
       "mfma_busy_frac_chip": None if not (busy and med_ns) else round(busy / (med_ns * 1e-9 * 2.4e9 * 1024), 4),
       "mfma_busy_formula": "SQ_VALU_MFMA_BUSY_CYCLES / (median kernel duration x 2.4 GHz x 1024 SIMDs)"}
json.dump(rec, open("gpurun_out/prof/m256/r06_m256_pmc.json", "w"), indent=1)
print(json.dumps(rec)[:1500])
PY
