"""Timing sweep over launch-plan overrides (development aid for the tuner).
Writes gpurun_out/sweep.json.  HBM-cold: rotating weight copies > 256 MiB."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from flute_amd import _lib, utils  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.get()
rows = []


def run(M, N, K, bits, g, dtype, tid, ovr, steps=300, hot=False):
    lay = bench.Layer(M, N, K, bits, g, dtype, dev, 1 if hot else bench.copies_for(N, K, bits))
    lay.template_id = tid
    lib.flute_set_overrides(*ovr)
    try:
        plan = utils.get_plan(M, N, K, bits, g, tid, lay.num_sms, dtype)
        ms, _ = bench.time_graph(lay, steps, 10, torch.cuda.synchronize)
        us = ms / steps * 1e3
        r = {"M": M, "N": N, "K": K, "bits": bits, "g": g, "dtype": str(dtype), "tid": tid,
             "ovr": ovr, "hot": hot, "us": round(us, 3), "GBps": round(lay.bytes() / us / 1e3, 1),
             "TFLOPs": round(lay.flops() / us / 1e6, 2), "plan": plan}
    except Exception as ex:  # noqa: BLE001
        r = {"M": M, "N": N, "K": K, "bits": bits, "ovr": ovr, "error": str(ex)[:200]}
    lib.flute_set_overrides(-1, -1, -1, -1, -1, -1)
    rows.append(r)
    print(json.dumps(r), flush=True)
    del lay
    torch.cuda.empty_cache()


f16 = torch.float16
# decode kernel: waves x kw x lut copies x splitk at the headline shape
for waves in (4, 8):
    for kw in (1, 2, 4, 8):
        if kw > waves:
            continue
        for copies in (32, 16, 1):
            run(1, 4096, 4096, 4, 64, f16, 16, (0, -1, waves, kw, 1, copies))
run(1, 4096, 4096, 4, 64, f16, 16, (0, -1, 8, 2, 2, 32))
run(1, 4096, 4096, 4, 64, f16, 16, (0, -1, 8, 2, 1, 32), hot=True)
run(1, 4096, 4096, 4, 64, f16, 16, (1, 1, -1, -1, -1, 32))          # MFMA kernel at M=1
for (n, k) in ((11008, 4096), (4096, 14336), (28672, 8192)):
    for waves, kw in ((8, 1), (8, 2), (4, 1), (4, 2)):
        run(1, n, k, 4, 64, f16, 16, (0, -1, waves, kw, 1, 32))
# M sweep: decode vs MFMA
for M in (2, 4, 8):
    run(M, 4096, 4096, 4, 64, f16, 16, (0, -1, 8, 2, 1, 32))
    run(M, 4096, 4096, 4, 64, f16, 16, (1, 1, -1, -1, -1, 32))
for M in (16, 32, 64, 256):
    for mt in (1, 2, 4):
        if mt * 16 > max(M, 16) * 1:
            continue
        for splitk in (1, 2, 4):
            run(M, 4096, 4096, 4, 64, f16, 16, (1, mt, -1, -1, splitk, 32))
for M in (16, 256):
    for mt in (1, 4):
        run(M, 11008, 4096, 4, 64, f16, 16, (1, mt, -1, -1, -1, 32))
# other bit widths (bf16 W3 70B shape, W2)
bf16 = torch.bfloat16
for waves, kw in ((8, 1), (8, 2), (4, 1)):
    run(1, 8192, 8192, 3, 64, bf16, 4, (0, -1, waves, kw, 1, 32))
run(1, 28672, 8192, 3, 64, bf16, 4, (0, -1, 8, 1, 1, 32))
run(16, 8192, 8192, 3, 64, bf16, 4, (1, 1, -1, -1, -1, 32))
run(1, 4096, 4096, 2, 64, f16, 4, (0, -1, 8, 2, 1, 32))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(rows, open("gpurun_out/sweep.json", "w"), indent=1)
