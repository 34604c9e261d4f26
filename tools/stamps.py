"""Phase breakdown of the column-per-lane MFMA kernel from the FLUTE_STAMPS development build.

    make -C flute_amd/csrc OBJDIR=build_stamps LIB=libflute_amd_stamps.so EXTRA=-DFLUTE_STAMPS -j
    FLUTE_AMD_LIB=flute_amd/csrc/libflute_amd_stamps.so python tools/stamps.py

Every wave writes four 100 MHz wall-clock stamps (start, prologue done, main loop done, stores
retired) into the workspace; this prints their distribution relative to the earliest start."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from flute_amd import _lib, utils  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.get()
f16 = torch.float16
FAM = int(os.environ.get("STAMPS_FAMILY", "2"))
dec_cases = [      # streaming decode kernel: (family 0, rows, waves, kw, splitk, -, -, ring depth)
    (1, 4096, 4096, (0, -1, -1, -1, -1, -1, -1, -1)),        # the planner's own choice: the one-shot variant
    (1, 4096, 4096, (0, -1, 8, 2, 1, -1, -1, 4)),
    (1, 4096, 4096, (0, -1, 16, 4, 1, -1, -1, 4)),
    (1, 4096, 4096, (0, -1, 16, 4, 1, -1, -1, 2)),
    (1, 4096, 4096, (0, -1, 8, 4, 1, -1, -1, 4)),
    (1, 11008, 4096, (0, -1, 11, 1, 1, -1, -1, 4)),
    (1, 28672, 8192, (0, -1, 14, 1, 1, -1, -1, 4)),
    (4, 4096, 4096, (0, -1, -1, -1, -1, -1, -1, -1)),
]
cases = [
    (256, 4096, 4096, (FAM, 1, 8, 8, 1, 4, -1)),
    (256, 4096, 512, (FAM, 1, 8, 8, 1, 4, -1)),
    (256, 4096, 4096, (FAM, 1, 8, 4, 1, 4, -1)),
    (256, 11008, 4096, (FAM, 1, 8, 2, 1, 4, -1)),
    (16, 4096, 4096, (FAM, 4, 8, 8, 1, 1, -1)),
    (16, 4096, 1024, (FAM, 4, 8, 8, 1, 1, -1)),
    (64, 4096, 4096, (FAM, 1, 8, 8, 1, 1, -1)),
    (4096, 4096, 4096, (FAM, 1, 8, 1, 1, 4, -1)),
]
out = []
ONLY_M = int(os.environ.get("STAMPS_ONLY_M", "0"))     # restrict to one batch size (ablation passes)
for (M, N, K, ovr) in (dec_cases if FAM == 0 else cases):
    if ONLY_M and M != ONLY_M:
        continue
    COLD = os.environ.get("STAMPS_COLD") == "1"      # rotate over > 256 MiB of weight copies: the stamped launch reads HBM
    lay = bench.Layer(M, N, K, 4, 64, f16, dev, bench.copies_for(N, K, 4) if COLD else 2)
    lay.template_id = 16
    from flute_amd import dev as dev_mod
    lay.ovr = dev_mod.overrides_from_tuple(ovr)
    plan = dev_mod.get_plan(M, N, K, 4, 64, 16, lay.num_sms, f16, lay.ovr)
    nwaves = plan["grid"] * plan["waves"]
    ws64 = lay.ws.view(torch.int64)[8192:]       # stamps live behind the 64 KB of xwg state words (api.hip)
    for i in range(len(lay.Q) if COLD else 3):
        lay.step(i)
    torch.cuda.synchronize()
    ws64[: nwaves * 8].zero_()
    torch.cuda.synchronize()
    lay.step(0)
    torch.cuda.synchronize()
    st = ws64[: nwaves * 8].reshape(nwaves, 8).cpu().double()
    t0 = st[:, 0].min()
    us = (st - t0) / 100.0
    q = lambda x: [round(float(v), 2) for v in (x.min(), x.median(), x.max())]  # noqa: E731
    r = {"M": M, "N": N, "K": K, "ovr": list(ovr), "grid": plan["grid"], "waves": plan["waves"],
         "start_us[min,med,max]": q(us[:, 0]),
         "prologue_us": q(us[:, 1] - us[:, 0]),
         "loop_cycles": q(st[:, 4]), "dma_wait_cycles": q(st[:, 7]), "pro_issue_lut": q(us[:, 5] - us[:, 0]),
         # streaming decode kernel: start -> every prologue load issued -> table/activation data back ->
         # table image written -> first unit ready (scales staged, barrier) -> first unit streamed -> end
         "dec_issue": q(us[:, 4] - us[:, 0]), "dec_first_data": q(us[:, 5] - us[:, 4]), "dec_table_image": q(us[:, 6] - us[:, 5]),
         "dec_x_scales_barrier": q(us[:, 1] - us[:, 6]), "dec_first_unit": q(us[:, 2] - us[:, 1]), "dec_rest": q(us[:, 3] - us[:, 2]),
         "pro_scales": q(us[:, 6] - us[:, 5]), "pro_barrier": q(us[:, 1] - us[:, 6]),
         "mainloop_us": q(us[:, 2] - us[:, 1]),
         "epilogue_us": q(us[:, 3] - us[:, 2]),
         "end_us": q(us[:, 3])}
    # keep only the fields of the kernel that ran
    drop = ('dec_',) if FAM != 0 else ('pro_', 'loop_cycles', 'dma_wait_cycles')
    r = {k: v for k, v in r.items() if not k.startswith(drop)}
    out.append(r)
    print(json.dumps(r), flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/stamps.json", "w"), indent=1)
