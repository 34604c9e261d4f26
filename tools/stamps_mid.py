"""Phase breakdown of the mid-M kernel (qgemm_mid.h) from the FLUTE_STAMPS development build:
    make -C flute_amd/csrc OBJDIR=build_stamps LIB=libflute_amd_stamps.so EXTRA=-DFLUTE_STAMPS -j
    FLUTE_AMD_LIB=flute_amd/csrc/libflute_amd_stamps.so python tools/stamps_mid.py
Every wave sums the shader cycles of each phase of a 64-k step over its steps; prints cycles per step."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from flute_amd import dev, utils  # noqa: E402

d = torch.device("cuda:0")
names = ["frag_reads_issued", "batch_issued", "lookups_waited", "mul+vmcnt_waited", "prefetch_issued", "frags_waited", "mfma_issued", "barrier"]
out = []
for (M, N, K, rt) in ((256, 4096, 4096, 4), (512, 4096, 4096, 8), (256, 11008, 4096, 4)):
    lay = bench.Layer(M, N, K, 4, 64, torch.float16, d, 2)
    lay.template_id = 16
    lay.ovr = dev.Overrides(family=5, m_tiles=rt)
    plan = dev.get_plan(M, N, K, 4, 64, 16, lay.num_sms, torch.float16, lay.ovr)
    nw = plan["grid"] * 8
    ws64 = lay.ws.view(torch.int64)
    for i in range(3):
        lay.step(i)
    torch.cuda.synchronize()
    ws64[: nw * 16].zero_()
    torch.cuda.synchronize()
    lay.step(0)
    torch.cuda.synchronize()
    st = ws64[: nw * 16].view(nw, 16).cpu().double()
    steps = st[:, 8].clamp(min=1)
    per = (st[:, :8] / steps[:, None])
    rec = {"M": M, "N": N, "K": K, "rt": rt, "grid": plan["grid"], "cycles_per_step_median": {n: round(per[:, i].median().item(), 1) for i, n in enumerate(names)},
           "total_per_step": round(per.sum(1).median().item(), 1)}
    print(json.dumps(rec))
    out.append(rec)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/stamps_mid.json", "w"), indent=1)
