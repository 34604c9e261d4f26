"""Block-tiled prefill kernel lab: correctness (fp32 reference on the GPU, one-hot rows exact) and HBM-cold
timing against the per-wave MFMA kernel (family 2) and torch.mm fp16.  Writes gpurun_out/block_lab.json."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import flute_amd  # noqa: E402
from flute_amd import dev, utils  # noqa: E402

d = torch.device("cuda:0")
num_sms = utils.get_device_num_sms(d)
ws = utils.get_workspace_streamk(d)
f16, bf16 = torch.float16, torch.bfloat16
rows = []
what = sys.argv[1:] or ["check", "time"]


def tid_of(bits, tile_p):
    return min(t for (b, t), c in flute_amd.TEMPLATE_CONFIGS.items() if b == bits and c["TileP"] == tile_p)


def emit(r):
    rows.append(r)
    print(json.dumps(r), flush=True)


def check():
    nfail = 0
    for (bits, tile_p, g, dtype, K, N) in [(4, 32, 64, f16, 4096, 4096), (4, 64, 64, bf16, 2048, 1024), (4, 32, 128, f16, 3072, 512),
                                           (4, 32, 32, f16, 1024, 256), (4, 64, 256, bf16, 4096, 256), (4, 32, 64, f16, 4096, 11008),
                                           (2, 32, 64, f16, 4096, 2048), (2, 64, 128, bf16, 2048, 1024), (2, 32, 32, bf16, 1024, 256),
                                           (2, 64, 256, f16, 3072, 512),
                                           (3, 32, 64, f16, 4096, 2048), (3, 32, 128, bf16, 2048, 1024), (3, 32, 32, bf16, 1024, 512),
                                           (3, 32, 256, f16, 3072, 512), (3, 32, 64, bf16, 8192, 512)]:
        torch.manual_seed(K + N)
        W = torch.randint(0, 2 ** bits, (K, N), dtype=torch.uint8, device=d)
        S = torch.randn(N, K // g, device=d).to(dtype)
        table = torch.randn(2 ** bits, device=d).to(dtype)
        table2 = utils.make_qmap2_from_qmap(table)
        tid = tid_of(bits, tile_p)
        Q = utils.pack(W, bits, [tid], num_sms)
        What = table[W.long()] * torch.repeat_interleave(S, g, dim=1).T
        tol = 1e-3 if dtype == f16 else 8e-3
        for M in (1, 100, 256, 300, 1024):
            X = (torch.randn(M, K, device=d) / 100).to(dtype)
            ref = X.float() @ What.float()
            ks = torch.randint(0, K, (M,), device=d)
            E = torch.zeros(M, K, device=d, dtype=dtype)
            E[torch.arange(M, device=d), ks] = 1
            for shp in (dict(family=3, m_tiles=8), dict(family=3, m_tiles=4), dict(family=3, m_tiles=8, splitk=2),
                        dict(family=3, m_tiles=8, slabs_per_wave=3),
                        dict(family=3, m_tiles=8, splitk=2, slabs_per_wave=3), dict(family=3, m_tiles=4, slabs_per_wave=3),
                        dict(family=3, m_tiles=4, splitk=2, slabs_per_wave=3)):
                rec = {"kind": "check", "bits": bits, "tile_p": tile_p, "g": g, "dtype": str(dtype)[6:], "K": K, "N": N, "M": M, "shape": shp}
                try:
                    ovr = dev.Overrides(**shp)
                    pl = dev.get_plan(M, N, K, bits, g, tid, num_sms, dtype, ovr)
                    if pl["family"] != 3:
                        continue
                    out = dev.qgemm_planned(X, Q, S, table, table2, ws, bits, g, tid, num_sms, ovr)
                    out1 = dev.qgemm_planned(E, Q, S, table, table2, ws, bits, g, tid, num_sms, ovr)
                    torch.cuda.synchronize()
                    err = ((out.float() - ref).norm() / ref.norm()).item()
                    exact = bool(torch.equal(out1, What[ks]))
                    rec.update(err=err, onehot_exact=exact, ok=bool(err < tol and exact))
                    if not rec["ok"]:
                        bad = ((out.float() - ref).abs() > 0.05 * ref.abs().max()) | out.float().isnan()
                        rec["nbad"] = int(bad.sum().item())
                        rec["bad_rows"] = bad.any(1).nonzero().flatten()[:10].tolist()
                        rec["bad_cols"] = bad.any(0).nonzero().flatten()[:16].tolist()
                        rec["onehot_mismatch"] = int((out1 != What[ks]).sum().item())
                except Exception as ex:  # noqa: BLE001
                    rec.update(ok=False, error=str(ex)[:300])
                if not rec["ok"]:
                    nfail += 1
                emit(rec)
        del W, S, Q, What
        torch.cuda.empty_cache()
    emit({"kind": "check_summary", "failed": nfail})
    return nfail


def timing():
    for (M, N, K) in ((4096, 4096, 4096), (1024, 4096, 4096), (1024, 11008, 4096), (2048, 4096, 4096), (4096, 11008, 4096),
                      (512, 11008, 4096), (256, 11008, 4096), (512, 4096, 4096), (2048, 11008, 4096), (1024, 8192, 8192), (512, 28672, 8192),
                      (256, 4096, 4096), (256, 28672, 8192), (256, 8192, 8192), (128, 11008, 4096), (128, 28672, 8192)):
        for dtype in (f16, bf16):
            for shp in (dict(family=2), dict(family=3, m_tiles=8), dict(family=3, m_tiles=4), dict(family=3, m_tiles=8, slabs_per_wave=3), dict(family=3, m_tiles=4, slabs_per_wave=3), dict()):
                lay = bench.Layer(M, N, K, 4, 64, dtype, d, bench.copies_for(N, K, 4))
                lay.template_id = 16
                if shp.get("family") == 2:
                    lay.tune()                      # the per-wave kernel's own best template
                    lay.ovr = dev.Overrides(family=2)
                else:
                    lay.ovr = dev.Overrides(**shp)
                rec = {"kind": "time", "M": M, "N": N, "K": K, "dtype": str(dtype)[6:], "shape": shp}
                try:
                    pl = dev.get_plan(M, N, K, 4, 64, lay.template_id, num_sms, dtype, lay.ovr)
                    rec["plan"] = {k: pl[k] for k in ("family", "m_block", "m_tiles", "waves", "kw", "splitk", "grid")}
                    steps = 100 if M >= 2048 else 200
                    ms = min(bench.time_graph(lay, steps, 5, torch.cuda.synchronize)[0] for _ in range(2))
                    us = ms / steps * 1e3
                    rec.update(us=round(us, 2), TFLOPs=round(lay.flops() / us / 1e6, 1), frac=round(lay.flops() / us / 1e6 / 2500, 3))
                except Exception as ex:  # noqa: BLE001
                    rec["error"] = str(ex)[:200]
                emit(rec)
                del lay
                torch.cuda.empty_cache()
        if dtype is not None:
            Wd = [torch.randn(K, N, device=d, dtype=f16) for _ in range(max(2, (300 << 20) // (K * N * 2) + 1))]
            Xd = torch.randn(M, K, device=d, dtype=f16)

            class Dense:
                def step(self, i):
                    return torch.mm(Xd, Wd[i % len(Wd)])
            ms, _ = bench.time_graph(Dense(), 100, 5, torch.cuda.synchronize)
            us = ms / 100 * 1e3
            emit({"kind": "time", "M": M, "N": N, "K": K, "shape": "torch.mm fp16", "us": round(us, 2),
                  "TFLOPs": round(2 * M * N * K / us / 1e6, 1)})
            del Wd, Xd
            torch.cuda.empty_cache()


def timing_b2(bits=2):
    for (M, N, K) in ((4096, 4096, 4096), (2048, 4096, 4096), (4096, 11008 if bits == 2 else 11264, 4096), (1024, 11008 if bits == 2 else 11264, 4096),
                      (1024, 4096, 4096), (512, 8192, 8192), (256, 28672, 8192)):
        for dtype in (f16, bf16):
            for shp in (dict(family=2), dict(family=3, m_tiles=8), dict(family=3, m_tiles=4), dict()):
                lay = bench.Layer(M, N, K, bits, 64, dtype, d, bench.copies_for(N, K, bits))
                lay.template_id = tid_of(bits, 32)
                if shp.get("family") == 2:
                    lay.tune()
                lay.ovr = dev.Overrides(**shp)
                rec = {"kind": "time", "bits": bits, "M": M, "N": N, "K": K, "dtype": str(dtype)[6:], "shape": shp}
                try:
                    pl = dev.get_plan(M, N, K, bits, 64, lay.template_id, num_sms, dtype, lay.ovr)
                    rec["plan"] = {k: pl[k] for k in ("family", "m_block", "m_tiles", "waves", "kw", "splitk", "grid")}
                    ms = min(bench.time_graph(lay, 100, 5, torch.cuda.synchronize)[0] for _ in range(2))
                    us = ms * 10
                    rec.update(us=round(us, 2), TFLOPs=round(lay.flops() / us / 1e6, 1))
                except Exception as ex:  # noqa: BLE001
                    rec["error"] = str(ex)[:200]
                emit(rec)
                del lay
                torch.cuda.empty_cache()


rc = 0
if "time_b2" in what:
    timing_b2()
if "time_b3" in what:
    timing_b2(3)
if "check" in what:
    rc = check()
if "time" in what:
    timing()
os.makedirs("gpurun_out", exist_ok=True)
json.dump(rows, open("gpurun_out/block_lab.json", "w"), indent=1)
sys.exit(1 if rc else 0)
