"""Phase stamps of the lean decode kernels (qgemm_fast.h / qgemm_fastm.h; FLUTE_STAMPS development build):

    make -C flute_amd/csrc OBJDIR=build_stamps LIB=libflute_amd_stamps.so EXTRA=-DFLUTE_STAMPS -j
    FLUTE_AMD_LIB=$PWD/flute_amd/csrc/libflute_amd_stamps.so python tools/stamps_fast.py "M,N,K[,override=value...];..."

One stamped HBM-cold launch per case: shader-clock cycles since each wave's start (p10 / median / p90 over the waves of the
launch), start skew and end of the launch in us (100 MHz chip clock)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from flute_amd import dev, utils  # noqa: E402

NAMES = {2: "requests_issued", 3: "table_word", 4: "image_written", 5: "first_activation_word", 6: "activations_written", 7: "barrier",
         8: "scale_image", 9: "loop_done", 10: "reduced", 11: "stored", 12: "store_acked"}
d = torch.device("cuda:0")
num_sms = utils.get_device_num_sms(d)
for c in (sys.argv[1] if len(sys.argv) > 1 else "16,4096,4096,family=7;1,4096,4096").split(";"):
    M, N, K, *ov = c.split(",")
    M, N, K = int(M), int(N), int(K)
    ovr = dev.Overrides(**{kv.split("=")[0]: int(kv.split("=")[1]) for kv in ov}) if ov else None
    lay = bench.Layer(M, N, K, 4, 64, torch.float16, d, bench.copies_for(N, K, 4))
    lay.template_id = 16
    lay.ovr = ovr if ovr is not None else dev.Overrides()
    plan = dev.get_plan(M, N, K, 4, 64, 16, num_sms, torch.float16, lay.ovr)
    nw = plan["grid"] * plan["waves"]
    ws64 = lay.ws.view(torch.int64)[8192:]                     # stamps live behind the 64 KB of xwg state words (api.hip)
    for i in range(len(lay.Q)):
        lay.step(i)
    torch.cuda.synchronize()
    ws64[: nw * 16].zero_()
    torch.cuda.synchronize()
    lay.step(0)                                                 # copy 0: pushed out of the caches by the other copies
    torch.cuda.synchronize()
    st = ws64[: nw * 16].reshape(nw, 16).cpu().double()
    ws64[: nw * 16].zero_()
    t0 = st[:, 0].min()
    q = lambda x: [round(float(torch.quantile(x, p))) for p in (0.1, 0.5, 0.9)]  # noqa: E731
    rec = {"case": c, "plan": {k: plan[k] for k in ("family", "one_shot", "waves", "kw", "ring_depth", "m_block", "grid", "lds_bytes")},
           "start_us[p50,p90,max]": [round(float(v), 2) for v in (((st[:, 0] - t0) / 100).median(), torch.quantile((st[:, 0] - t0) / 100, 0.9), ((st[:, 0] - t0) / 100).max())],
           "end_us[p50,p90,max]": [round(float(v), 2) for v in (((st[:, 13] - t0) / 100).median(), torch.quantile((st[:, 13] - t0) / 100, 0.9), ((st[:, 13] - t0) / 100).max())]}
    for i, name in NAMES.items():
        col = st[:, i]
        if (col > 0).any():
            rec[name] = q((col - st[:, 1])[col > 0])
    print(json.dumps(rec), flush=True)
    del lay
    torch.cuda.empty_cache()
