"""Phase stamps of the skinny MFMA kernel (FLUTE_STAMPS development build):

    make -C flute_amd/csrc OBJDIR=build_stamps LIB=libflute_amd_stamps.so EXTRA=-DFLUTE_STAMPS -j
    FLUTE_AMD_LIB=$PWD/flute_amd/csrc/libflute_amd_stamps.so python tools/stamps_skinny.py

Shader-clock cycles since the wave's start, median over the waves of the launch (and per wave index)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import flute_amd  # noqa: E402
from flute_amd import dev, utils  # noqa: E402

d = torch.device("cuda:0")
num_sms = utils.get_device_num_sms(d)
ws = utils.get_workspace_streamk(d)
names = {2: "requests issued", 3: "table + scale image", 4: "barrier passed", 5: "k-step 0 released", 6: "k-step 1 released",
         7: "k-step D/2 released", 8: "last k-step released", 9: "k-steps done", 10: "reduced + stored", 11: "stores retired"}
out = []
for (M, N, K) in ((16, 14336, 4096), (4, 14336, 4096), (16, 4096, 4096)):
    bits, g, dtype = 4, 64, torch.float16
    tid = min(t for (b, t), c in flute_amd.TEMPLATE_CONFIGS.items() if b == bits and c["TileP"] == 32)
    torch.manual_seed(0)
    W = torch.randint(0, 16, (K, N), dtype=torch.uint8, device=d)
    S = torch.randn(N, K // g, device=d).to(dtype)
    table = torch.randn(16, device=d).to(dtype)
    table2 = utils.make_qmap2_from_qmap(table)
    Q = utils.pack(W, bits, [tid], num_sms)
    X = (torch.randn(M, K, device=d) / 100).to(dtype)
    ovr = dev.Overrides(family=5, waves=8)
    plan = dev.get_plan(M, N, K, bits, g, tid, num_sms, dtype, ovr)
    flush = torch.empty(512 << 20, dtype=torch.uint8, device=d)
    for rep in range(3):
        flush.sum()                                     # weights out of the caches
        ws.zero_()
        dev.qgemm_planned(X, Q, S, table, table2, ws, bits, g, tid, num_sms, ovr)
        torch.cuda.synchronize()
    nw = plan["grid"] * plan["waves"]
    st = ws[:nw * 128].view(torch.int64).view(nw, 16).cpu()
    rel = (st - st[:, 1:2]).float()
    wall = (st[:, 12] - st[:, 0].min()).float() * 10.0   # ns
    start = (st[:, 0] - st[:, 0].min()).float() * 10.0
    rec = {"M": M, "N": N, "K": K, "plan": {k: plan[k] for k in ("grid", "waves", "ring_depth")},
           "wave_start_ns_p50_max": [start.median().item(), start.max().item()], "wave_end_ns_p50_max": [wall.median().item(), wall.max().item()],
           "cycles_since_wave_start_median": {names[i]: rel[:, i].median().item() for i in sorted(names)},
           "by_wave_index_median": {names[i]: [rel[w::plan["waves"], i].median().item() for w in range(plan["waves"])] for i in (2, 5, 8, 9)}}
    print(json.dumps(rec), flush=True)
    out.append(rec)
    del W, Q, S, flush
    torch.cuda.empty_cache()
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/stamps_skinny.json", "w"), indent=1)
