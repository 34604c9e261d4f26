"""Skinny MFMA kernel lab (qgemm_skinny.h, override family 5): correctness matrix against an fp32 evaluation of the
reference formula (never stops at the first failure) and
HBM-cold timing against the automatic plan.

    python tools/skinny_lab.py [check] [time]

Writes gpurun_out/skinny_lab.json."""
import json
import os
import sys
import time

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
    cases = [(4, 32, 64, f16, 1024, 512), (4, 64, 128, bf16, 2048, 1024), (4, 32, 256, f16, 4096, 256), (4, 32, 32, bf16, 1024, 256),
             (2, 32, 64, f16, 1536, 512), (2, 64, 128, bf16, 2048, 1024), (4, 32, 64, f16, 4096, 4096), (4, 64, 64, f16, 4096, 11008),
             (4, 32, 64, bf16, 2048, 4096), (4, 32, 128, f16, 512, 8192), (2, 32, 64, bf16, 4096, 4096), (4, 32, 64, f16, 512, 128),
             (2, 32, 128, f16, 2048, 2048), (4, 64, 256, bf16, 4096, 1024)]
    shapes = [dict(), dict(waves=4), dict(waves=8)]
    for (bits, tile_p, g, dtype, K, N) in cases:
        torch.manual_seed(K + N + bits)
        W = torch.randint(0, 2 ** bits, (K, N), dtype=torch.uint8, device=d)
        S = torch.randn(N, K // g, device=d).to(dtype)
        table = torch.randn(2 ** bits, device=d).to(dtype)
        table2 = utils.make_qmap2_from_qmap(table)
        tid = tid_of(bits, tile_p)
        Q = utils.pack(W, bits, [tid], num_sms)
        What = table[W.long()] * torch.repeat_interleave(S, g, dim=1).T
        tol = 1e-3 if dtype == f16 else 8e-3
        for M in (1, 3, 8, 16):
            X = (torch.randn(M, K, device=d) / 100).to(dtype)
            ref = X.float() @ What.float()
            ks = torch.randint(0, K, (M,), device=d)
            E = torch.zeros(M, K, device=d, dtype=dtype)
            E[torch.arange(M), ks] = 1
            for shp in shapes:
                rec = {"kind": "check", "bits": bits, "tile_p": tile_p, "g": g, "dtype": str(dtype)[6:], "K": K, "N": N, "M": M, "shape": shp}
                try:
                    ovr = dev.Overrides(family=5, **shp)
                    plan = dev.get_plan(M, N, K, bits, g, tid, num_sms, dtype, ovr)
                    if plan["family"] != 5:
                        rec.update(ok=True, skipped=True)        # this launch shape does not exist for this layer
                        rows.append(rec)
                        continue
                    out = dev.qgemm_planned(X, Q, S, table, table2, ws, bits, g, tid, num_sms, ovr)
                    out1 = dev.qgemm_planned(E, Q, S, table, table2, ws, bits, g, tid, num_sms, ovr)
                    torch.cuda.synchronize()
                    err = ((out.float() - ref).norm() / ref.norm()).item()
                    exact = bool(torch.equal(out1, What[ks]))
                    rec.update(err=err, onehot_exact=exact, ok=bool(err < tol and exact),
                               plan={k: plan[k] for k in ("waves", "splitk", "grid", "ring_depth", "m_tiles")})
                    if not rec["ok"]:
                        bad = ((out.float() - ref).abs() > 0.05 * ref.abs().max()).nonzero()
                        rec["nbad"] = int(bad.shape[0])
                        rec["first_bad"] = bad[:8].tolist()
                        rec["onehot_mismatch"] = int((out1 != What[ks]).sum().item())
                except Exception as ex:  # noqa: BLE001
                    rec.update(ok=False, error=str(ex)[:300])
                if not rec["ok"]:
                    nfail += 1
                    emit(rec)
                else:
                    rows.append(rec)
        del W, S, Q, What
        torch.cuda.empty_cache()
    emit({"kind": "check_summary", "total": len([r for r in rows if r.get("kind") == "check"]),
          "ran": len([r for r in rows if r.get("kind") == "check" and not r.get("skipped")]), "failed": nfail})
    return nfail


def time_case(M, N, K, bits, g, dtype, shp, steps=300, tile_p=32, tag=""):
    tid = tid_of(bits, tile_p)
    lay = bench.Layer(M, N, K, bits, g, dtype, d, bench.copies_for(N, K, bits))
    lay.template_id = tid
    lay.ovr = dev.Overrides(**shp)
    rec = {"kind": "time", "tag": tag, "M": M, "N": N, "K": K, "bits": bits, "g": g, "dtype": str(dtype)[6:], "shape": shp}
    try:
        rec["plan"] = {k: v for k, v in dev.get_plan(M, N, K, bits, g, tid, num_sms, dtype, lay.ovr).items()
                       if k in ("family", "waves", "kw", "splitk", "grid", "lds_bytes", "ring_depth", "m_tiles", "m_block")}
        best = 1e9
        for _ in range(2):
            ms, _w = bench.time_graph(lay, steps, 10, torch.cuda.synchronize)
            best = min(best, ms / steps * 1e3)
        rec["us"] = round(best, 3)
    except Exception as ex:  # noqa: BLE001
        rec["error"] = str(ex)[:300]
    emit(rec)
    del lay
    torch.cuda.empty_cache()


def timing():
    for (tag, N, K, bits, dt) in (("4096^2", 4096, 4096, 4, f16), ("4096x11008", 11008, 4096, 4, f16), ("4096x14336", 14336, 4096, 4, f16),
                                  ("4096x28672", 28672, 4096, 4, f16), ("2048x8192", 8192, 2048, 4, f16), ("4096x14336 bf16", 14336, 4096, 4, bf16),
                                  ("4096x14336 W2", 14336, 4096, 2, f16), ("4096x6144", 6144, 4096, 4, f16), ("4096x8192", 8192, 4096, 4, f16)):
        for M in (16, 4, 8, 32):
            if M != 16 and tag not in ("4096x14336", "4096x11008"):
                continue
            for shp in [dict(family=2), dict(family=5, waves=8), dict(family=5, waves=4)]:
                time_case(M, N, K, bits, 64, dt, shp, steps=300, tag=f"{tag} M={M}")


if __name__ == "__main__":
    t0 = time.time()
    rc = 0
    if "check" in what:
        rc = check()
    if "time" in what:
        timing()
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump([r for r in rows if r.get("kind") != "check" or not r.get("ok")], open("gpurun_out/skinny_lab.json", "w"), indent=1)
    print(f"skinny_lab done in {time.time() - t0:.1f}s, failures: {rc}")
    sys.exit(1 if rc else 0)
