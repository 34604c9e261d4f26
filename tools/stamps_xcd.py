"""Where do the waves of a big decode launch finish, by XCD (block id % 8) and by visit count?  FLUTE_STAMPS build.

    FLUTE_AMD_LIB=flute_amd/csrc/libflute_amd_stamps.so python tools/stamps_xcd.py
"""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from flute_amd import dev as dev_mod, utils  # noqa: E402
d = torch.device("cuda:0")
out = []
for (N, K, bits, dt, tid, shp) in ((28672, 8192, 4, torch.float16, 20, dict()), (28672, 8192, 4, torch.float16, 20, dict(waves=14, kw=1)),
                                   (8192, 28672, 4, torch.float16, 48, dict())):
    lay = bench.Layer(1, N, K, bits, 64, dt, d, bench.copies_for(N, K, bits))
    lay.template_id = tid
    lay.ovr = dev_mod.Overrides(**shp) if shp else dev_mod.Overrides()
    plan = dev_mod.get_plan(1, N, K, bits, 64, tid, lay.num_sms, dt, lay.ovr)
    W, G = plan["waves"], plan["grid"]
    nwaves = G * W
    ws64 = lay.ws.view(torch.int64)[8192:]       # stamps live behind the 64 KB of xwg state words (api.hip)
    for i in range(len(lay.Q)):
        lay.step(i)
    torch.cuda.synchronize()
    for rep in range(4):
        ws64[: nwaves * 8].zero_()
        torch.cuda.synchronize()
        lay.step(rep)
        torch.cuda.synchronize()
        st = ws64[: nwaves * 8].reshape(G, W, 8).cpu().double()
        t0 = st[:, :, 0].min()
        end = (st[:, :, 3] - t0) / 100.0                      # [wg, wave] us
        start = (st[:, :, 0] - t0) / 100.0
        wg_end = end.max(dim=1).values
        xcd = torch.arange(G) % 8
        rec = {"N": N, "K": K, "shape": shp, "rep": rep, "waves": W, "kw": plan["kw"], "visits": plan["visits"],
               "end_all[min,med,max]": [round(float(v), 2) for v in (end.min(), end.median(), end.max())],
               "wg_end_by_xcd[med,max]": [[round(float(wg_end[xcd == x].median()), 2), round(float(wg_end[xcd == x].max()), 2)] for x in range(8)],
               "wg_start_by_xcd[med]": [round(float(start.min(dim=1).values[xcd == x].median()), 2) for x in range(8)],
               "wg_end_quantiles[10,25,50,75,90,100]": [round(float(torch.quantile(wg_end, q)), 2) for q in (0.1, 0.25, 0.5, 0.75, 0.9, 1.0)],
               "slowest_wgs": torch.argsort(wg_end, descending=True)[:12].tolist()}
        out.append(rec)
        print(json.dumps(rec), flush=True)
    del lay
    torch.cuda.empty_cache()
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/stamps_xcd.json", "w"), indent=1)
