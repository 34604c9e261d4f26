"""Phase breakdown of the split-K block kernel (qgemm_splitk.h) from the FLUTE_STAMPS development build.

    make -C flute_amd/csrc OBJDIR=build_stamps LIB=libflute_amd_stamps.so EXTRA=-DFLUTE_STAMPS -j
    FLUTE_AMD_LIB=flute_amd/csrc/libflute_amd_stamps.so python tools/stamps_splitk.py

Every wave writes 100 MHz wall-clock stamps behind the slabs: 0 start, 1 prologue done (table, scales, first batch), 2 main
loop done, 3 K halves exchanged, 4 partial stores issued, 5 arrived (drain + barrier + atomic), 6 all slices in (owners) /
own share next (last arriver), 7 other slices' partials read, 8 stores retired; 9 = arrivals before this workgroup."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from flute_amd import dev, utils  # noqa: E402

d = torch.device("cuda:0")
f16 = torch.float16
out = []
CASES = ((256, 4096, 4096, 4, -1), (256, 4096, 4096, 2, -1), (256, 4096, 4096, 1, -1), (256, 4096, 4096, 8, -1), (256, 11008, 4096, 1, -1), (256, 11008, 4096, 2, -1))
if os.environ.get("STAMPS_KP4") == "1":      # round 6: four K parts per workgroup (64 x 64 tiles, no seam) against the two-slice plan
    CASES = ((256, 4096, 4096, 1, 4), (256, 4096, 4096, 2, 2))
for (M, N, K, sk, kw) in CASES:
    lay = bench.Layer(M, N, K, 4, 64, f16, d, 1 if os.environ.get("STAMPS_HOT") == "1" else bench.copies_for(N, K, 4))   # STAMPS_HOT=1: the weights stay cache-resident
    lay.template_id = 16
    lay.ovr = dev.Overrides(family=6, splitk=sk, kw=kw)
    plan = dev.get_plan(M, N, K, 4, 64, 16, lay.num_sms, f16, lay.ovr)
    nw = plan["grid"] * 8
    base = (65536 + sk * M * N * 4) // 8          # stamps sit behind splitk * M * N floats from the slab base (M = 256: whole tiles)
    ws64 = lay.ws.view(torch.int64)
    for i in range(len(lay.Q)):
        lay.step(i)
    torch.cuda.synchronize()
    lay.step(0)
    torch.cuda.synchronize()
    st = ws64[base: base + nw * 12].reshape(nw, 12).cpu().double()
    t0 = st[:, 0].min()
    us = (st - t0) / 100.0
    q = lambda x: [round(float(v), 2) for v in (x.min(), x.median(), x.max())]  # noqa: E731
    r = {"hot": os.environ.get("STAMPS_HOT") == "1", "M": M, "N": N, "K": K, "splitk": sk, "kw": plan["kw"], "m_tiles": plan["m_tiles"], "grid": plan["grid"], "start[min,med,max]": q(us[:, 0]), "prologue": q(us[:, 1] - us[:, 0]),
         "mainloop": q(us[:, 2] - us[:, 1]), "exchange": q(us[:, 3] - us[:, 2])}
    if sk > 1:
        last = st[:, 9] == sk - 1
        r.update({"publish_issue": q(us[:, 4] - us[:, 3]), "drain_arrive": q(us[:, 5] - us[:, 4]),
                  "wait_all(owners)": q((us[:, 6] - us[:, 5])[~last]) if (sk in (2, 4) and (~last).any()) else None,
                  "combine_loads": q((us[:, 7] - us[:, 6])[us[:, 7] > 0]) if sk in (2, 4) else None,
                  "arrive_time": q(us[:, 5]), "end": q(us[:, 8])})
    else:
        r["end"] = q(us[:, 8])
    out.append(r)
    print(json.dumps(r), flush=True)
    ws64[base: base + nw * 12].zero_()
    del lay
    torch.cuda.empty_cache()
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/stamps_splitk.json", "w"), indent=1)
