"""Time the AUTOMATIC plan of a list of cases (what flute.qgemm launches): HBM-cold hipGraph replays, best of three.

    python tools/time_cases.py "bits,M,N,K,dtype[,override=value...];..." [--steps 200] [--out gpurun_out/time_cases.jsonl]

One JSON line per case: time, plan (family / block shape / split), template id.  Used for before / after runs of a kernel
change; with overrides (`family=3,m_tiles=4,splitk=2`: flute_overrides fields) for a forced plan next to the automatic one."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from flute_amd import dev, utils  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("cases")
ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--tag", default="")
ap.add_argument("--out", default="gpurun_out/time_cases.jsonl")
a = ap.parse_args()
d = torch.device("cuda:0")
num_sms = utils.get_device_num_sms(d)
os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
with open(a.out, "a") as f:
    for c in a.cases.split(";"):
        bits, M, N, K, dt, *ov = c.split(",")
        ovr = dev.Overrides(**{kv.split("=")[0]: int(kv.split("=")[1]) for kv in ov}) if ov else None
        bits, M, N, K = int(bits), int(M), int(N), int(K)
        dtype = torch.float16 if dt in ("f16", "float16") else torch.bfloat16
        lay = bench.Layer(M, N, K, bits, 64, dtype, d, bench.copies_for(N, K, bits))
        tid = lay.tune()
        try:
            plan = dev.get_plan(M, N, K, bits, 64, tid, num_sms, dtype, ovr)
        except RuntimeError as ex:
            print(json.dumps({"case": c, "error": str(ex)[:120]}), flush=True)
            continue
        lay.ovr = ovr
        steps = a.steps if M * N * K < (1 << 36) else max(20, a.steps // 4)
        us = min(bench.time_graph(lay, steps, 3, torch.cuda.synchronize)[0] for _ in range(3)) / steps * 1e3
        rec = {"tag": a.tag, "bits": bits, "M": M, "N": N, "K": K, "dtype": dt, "ovr": ",".join(ov), "tid": tid, "us": round(us, 2),
               "TFLOPs": round(2.0 * M * N * K / us / 1e6, 1),
               "plan": {k: plan[k] for k in ("family", "one_shot", "waves", "kw", "ring_depth", "m_block", "m_tiles", "splitk", "grid")}}
        print(json.dumps(rec), flush=True)
        f.write(json.dumps(rec) + "\n")
        del lay
        torch.cuda.empty_cache()
