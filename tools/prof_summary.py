"""Condense rocprofv3 csv output of tools/prof_run.sh into one small text file."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
lines = []
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        lines.append("STATS " + " | ".join(f"{k}={r[k]}" for k in r))
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True):
    d = defaultdict(list)
    for r in csv.DictReader(open(f)):
        d[r["Kernel_Name"][:60]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"]),
                                         r.get("VGPR_Count"), r.get("LDS_Block_Size"), r.get("Grid_Size"), r.get("Workgroup_Size")))
    for k, v in d.items():
        ds = sorted(x[0] for x in v)
        lines.append(f"TRACE {k} n={len(ds)} min={ds[0]} med={ds[len(ds)//2]} avg={sum(ds) / len(ds):.1f} max={ds[-1]} ns vgpr={v[0][1]} lds={v[0][2]} grid={v[0][3]} wg={v[0][4]}")
for p in ("pmc1", "pmc2", "pmc3", "pmc4"):
    for f in glob.glob(os.path.join(out, p, "**", "*counter_collection.csv"), recursive=True):
        acc = defaultdict(lambda: defaultdict(list))
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in acc.items():
            lines.append(f"PMC {k}: " + " ".join(f"{c}={sum(v)/len(v):.0f}" for c, v in sorted(cs.items())))
open(os.path.join(out, "summary.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
