"""Streaming decode kernel lab: correctness matrix (never stops at the first failure) and HBM-cold timing of
launch shapes against the round-1 decode kernel (override family 4), one GPU call.

    python tools/decode_lab.py [check] [time] [time_small]

Writes gpurun_out/decode_lab.json (one record per case)."""
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
    cases = [
        (4, 32, 64, f16, 1024, 512), (4, 64, 64, bf16, 2048, 1024), (4, 32, 128, f16, 4096, 256),
        (4, 32, 256, bf16, 2048, 256), (4, 32, 32, f16, 1024, 256), (2, 32, 64, f16, 1536, 512),
        (2, 64, 128, bf16, 2048, 1024), (3, 32, 64, bf16, 2048, 1024), (3, 32, 64, f16, 1024, 512),
        (4, 32, 64, f16, 4416, 256), (4, 32, 64, f16, 192, 128), (4, 32, 64, f16, 4096, 4096),
        (4, 32, 64, f16, 8192, 3584), (3, 32, 64, bf16, 8192, 8192), (4, 64, 64, f16, 4096, 11008),
    ]
    shapes = [dict(), dict(waves=16, kw=4), dict(waves=8, kw=2), dict(waves=14, kw=1), dict(waves=5, kw=1),
              dict(waves=16, kw=16), dict(waves=12, kw=4, ring_depth=2), dict(waves=8, kw=8, ring_depth=2),
              dict(waves=1, kw=1), dict(waves=4, kw=2, splitk=2),
              # one-shot kernel (qgemm_oneshot.h); combinations a shape cannot take (K too long for kw x depth) are skipped
              dict(one_shot=0), dict(one_shot=1), dict(one_shot=1, ring_depth=4), dict(one_shot=1, ring_depth=8), dict(one_shot=1, ring_depth=2),
              dict(one_shot=1, waves=8), dict(one_shot=1, waves=16), dict(one_shot=1, waves=8, kw=8), dict(one_shot=1, waves=16, kw=4),
              dict(one_shot=1, waves=12, kw=4), dict(one_shot=1, waves=6, kw=2, ring_depth=4), dict(one_shot=1, waves=5, kw=1),
              # persistent one-shot kernel (qgemm_persist.h; M = 1 and whole chunks only - the ring kernel otherwise)
              dict(one_shot=2), dict(one_shot=2, waves=8), dict(one_shot=2, waves=5), dict(one_shot=2, waves=7, ring_depth=4),
              dict(one_shot=2, waves=4, ring_depth=2), dict(one_shot=2, ring_depth=2, slabs_per_wave=3),
              dict(one_shot=2, ring_depth=2, slabs_per_wave=4), dict(one_shot=2, waves=6, ring_depth=4, slabs_per_wave=3)]
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
        for M in ((1, 2) if bits == 3 else (1, 2, 3, 4)):
            X = (torch.randn(M, K, device=d) / 100).to(dtype)
            ref = X.float() @ What.float()
            ks = torch.randint(0, K, (M,), device=d)
            E = torch.zeros(M, K, device=d, dtype=dtype)
            E[torch.arange(M), ks] = 1
            for shp in shapes:
                for sms in ((num_sms, 8) if N <= 1024 else (num_sms,)):
                    rec = {"kind": "check", "bits": bits, "tile_p": tile_p, "g": g, "dtype": str(dtype)[6:], "K": K,
                           "N": N, "M": M, "shape": shp, "num_sms": sms}
                    try:
                        ovr = dev.Overrides(**({"family": 0, **shp}))
                        out = dev.qgemm_planned(X, Q, S, table, table2, ws, bits, g, tid, sms, ovr)
                        out1 = dev.qgemm_planned(E, Q, S, table, table2, ws, bits, g, tid, sms, ovr)
                        torch.cuda.synchronize()
                        err = ((out.float() - ref).norm() / ref.norm()).item()
                        exact = bool(torch.equal(out1, What[ks]))
                        rec.update(err=err, onehot_exact=exact, ok=bool(err < tol and exact))
                        if not rec["ok"]:
                            bad = ((out.float() - ref).abs() > 0.05 * ref.abs().max()).nonzero()
                            rec["nbad"] = int(bad.shape[0])
                            rec["first_bad"] = bad[:8].tolist()
                            rec["onehot_mismatch"] = int((out1 != What[ks]).sum().item())
                            rec["plan"] = dev.get_plan(M, N, K, bits, g, tid, sms, dtype, ovr)
                    except Exception as ex:  # noqa: BLE001
                        rec.update(ok=False, error=str(ex)[:300])
                        if shp.get("one_shot") == 1 and "Unsupported shape" in str(ex):
                            rec.update(ok=True, skipped=True)        # this launch shape does not exist for this K
                    if not rec["ok"]:
                        nfail += 1
                        emit(rec)
                    else:
                        rows.append(rec)
        del W, S, Q, What
        torch.cuda.empty_cache()
    import collections
    by = collections.Counter((r["bits"], r["K"], r["N"], r["M"], json.dumps(r["shape"]), r["num_sms"], r.get("err"), r.get("onehot_exact"))
                             for r in rows if r.get("kind") == "check" and not r.get("ok"))
    for k, v in sorted(by.items(), key=str):
        print("FAILED", k, v, flush=True)
    emit({"kind": "check_summary", "total": len([r for r in rows if r.get("kind") == "check"]), "failed": nfail})
    return nfail


def time_case(M, N, K, bits, g, dtype, shp, steps=300, tile_p=32, hadamard=0, tag=""):
    tid = tid_of(bits, tile_p)
    lay = bench.Layer(M, N, K, bits, g, dtype, d, bench.copies_for(N, K, bits), hadamard_size=hadamard)
    lay.template_id = tid
    lay.ovr = dev.Overrides(**shp)
    rec = {"kind": "time", "tag": tag, "M": M, "N": N, "K": K, "bits": bits, "g": g, "dtype": str(dtype)[6:], "shape": shp}
    try:
        rec["plan"] = {k: v for k, v in dev.get_plan(M, N, K, bits, g, tid, num_sms, dtype, lay.ovr).items()
                       if k in ("family", "waves", "kw", "splitk", "grid", "lds_bytes", "ring_depth", "visits", "k_chunks", "one_shot")}
        best = 1e9
        for _ in range(2):
            ms, _w = bench.time_graph(lay, steps, 10, torch.cuda.synchronize)
            best = min(best, ms / steps * 1e3)
        rec["us"] = round(best, 3)
        rec["GBps"] = round(lay.bytes() / best / 1e3, 1)
    except Exception as ex:  # noqa: BLE001
        rec["error"] = str(ex)[:300]
    emit(rec)
    del lay
    torch.cuda.empty_cache()


def timing(small_only=False):
    # headline: 4096 x 4096 W4G64 fp16 M=1
    one = [dict(one_shot=1, waves=w, ring_depth=d) for d in (8, 4) for w in (4, 8, 16)]
    for shp in [dict(), dict(one_shot=0), dict(one_shot=0, waves=16, kw=4)] + one:
        time_case(1, 4096, 4096, 4, 64, f16, shp, steps=1000, tag="headline")
    if small_only:
        for (tag, M, N, K, bits, dt, had) in (("11008", 1, 11008, 4096, 4, f16, 0), ("14336", 1, 14336, 4096, 4, f16, 0),
                                              ("TP8 shard", 1, 3584, 8192, 4, f16, 0), ("higgs had512", 1, 4096, 3584, 4, f16, 512),
                                              ("8192 W3", 1, 8192, 8192, 3, bf16, 0), ("big W4", 1, 28672, 8192, 4, f16, 0),
                                              ("big W3", 1, 28672, 8192, 3, bf16, 0), ("down proj 70B", 1, 8192, 28672, 4, f16, 0),
                                              ("M=2 4096", 2, 4096, 4096, 4, f16, 0), ("W2", 1, 4096, 4096, 2, f16, 0),
                                              ("bf16 4096", 1, 4096, 4096, 4, bf16, 0), ("8192 g128", 1, 8192, 8192, 4, f16, 0)):
            g = 128 if "g128" in tag else 64
            ds = (4, 2) if bits == 3 else (8, 4)
            for shp in [dict(one_shot=0)] + [dict(one_shot=1, waves=w, ring_depth=d) for d in ds for w in (4, 8, 16)]:
                time_case(M, N, K, bits, g, dt, shp, steps=300, hadamard=had, tag=tag)
        return
    for shp in (dict(), dict(waves=16, kw=1), dict(waves=11, kw=1, ring_depth=2), dict(waves=16, kw=4),
                dict(waves=12, kw=1), dict(waves=8, kw=1)):
        time_case(1, 11008, 4096, 4, 64, f16, shp, steps=500, tag="11008")
    for shp in (dict(), dict(waves=14, kw=1, ring_depth=2), dict(waves=16, kw=1), dict(waves=16, kw=4),
                dict(waves=7, kw=1), dict(waves=15, kw=1), dict(waves=10, kw=1)):
        time_case(1, 28672, 8192, 4, 64, f16, shp, steps=200, tag="big W4")
    for shp in (dict(), dict(waves=16, kw=2), dict(waves=8, kw=1), dict(waves=14, kw=2), dict(waves=16, kw=4)):
        time_case(1, 28672, 8192, 3, 64, bf16, shp, steps=200, tag="big W3")
    for shp in (dict(), dict(waves=16, kw=8), dict(waves=8, kw=4), dict(waves=16, kw=4)):
        time_case(1, 8192, 8192, 3, 64, bf16, shp, steps=300, tag="8192 W3")
    for shp in (dict(), dict(waves=16, kw=4), dict(waves=14, kw=2)):
        time_case(1, 3584, 8192, 4, 64, f16, shp, steps=500, tag="TP8 shard")
    for shp in (dict()):
        time_case(1, 4096, 3584, 4, 64, f16, shp, steps=500, hadamard=512, tag="higgs had512")
        time_case(1, 8192, 28672, 4, 64, f16, shp, steps=200, tag="down proj 70B")
        time_case(4, 4096, 4096, 4, 64, f16, shp, steps=500, tag="M=4")
        time_case(2, 28672, 8192, 4, 64, f16, shp, steps=200, tag="M=2 big")
        time_case(1, 4096, 4096, 2, 64, f16, shp, steps=500, tag="W2")
        time_case(1, 8192, 8192, 4, 128, f16, shp, steps=300, tag="8192 g128")
        time_case(1, 4096, 4096, 4, 64, bf16, shp, steps=500, tag="bf16 4096")


def timing_ring():
    """The persistent ring kernel on the layers that keep it (one_shot = 0): A/B of library builds."""
    for (tag, M, N, K, bits, dt) in (("big W4", 1, 28672, 8192, 4, f16), ("down proj 70B", 1, 8192, 28672, 4, f16), ("big W3", 1, 28672, 8192, 3, bf16),
                                     ("8192 W3", 1, 8192, 8192, 3, bf16), ("8192 g64", 1, 8192, 8192, 4, f16), ("M=2 big", 2, 28672, 8192, 4, f16),
                                     ("14336x8192", 1, 14336, 8192, 4, f16), ("W2 big", 1, 28672, 8192, 2, f16)):
        shapes = [dict(one_shot=0), dict(one_shot=0, ring_depth=2), dict(one_shot=0, ring_depth=4)]
        if bits != 3:
            shapes += [dict(one_shot=0, waves=16, kw=2), dict(one_shot=0, waves=12, kw=1)]
        for shp in shapes:
            time_case(M, N, K, bits, 64, dt, shp, steps=200, tag=tag)


def timing_persist():
    """The persistent one-shot kernel (one_shot = 2; ring_depth = pieces per segment, slabs_per_wave = register sets)
    against the ring and the one-shot kernel."""
    for (tag, M, N, K, bits, dt) in (("big W4", 1, 28672, 8192, 4, f16), ("down proj 70B", 1, 8192, 28672, 4, f16), ("big W3", 1, 28672, 8192, 3, bf16),
                                     ("8192 W3", 1, 8192, 8192, 3, bf16), ("8192 g64", 1, 8192, 8192, 4, f16),
                                     ("14336x8192", 1, 14336, 8192, 4, f16), ("W2 big", 1, 28672, 8192, 2, f16), ("4096 headline", 1, 4096, 4096, 4, f16),
                                     ("11008x4096", 1, 11008, 4096, 4, f16), ("4096x11008", 1, 4096, 11008, 4, f16), ("tp shard", 1, 3584, 8192, 4, f16),
                                     ("W3 4096", 1, 4096, 4096, 3, f16), ("W2 8192", 1, 8192, 8192, 2, f16)):
        pairs = ((2, 2), (2, 3), (2, 4), (4, 2)) if bits == 3 else ((4, 2), (4, 3), (2, 4), (2, 3), (8, 2))
        shapes = [dict(one_shot=0), dict(one_shot=1)] + [dict(one_shot=2, ring_depth=dd, slabs_per_wave=ns) for dd, ns in pairs]
        shapes += [dict(one_shot=2, waves=8, ring_depth=pairs[0][0], slabs_per_wave=2), dict(one_shot=2, waves=4, ring_depth=pairs[0][0], slabs_per_wave=2)]
        for shp in shapes:
            time_case(M, N, K, bits, 64, dt, shp, steps=200, tag=tag)


t0 = time.time()
rc = 0
if "check" in what:
    rc = check()
if "time" in what:
    timing()
if "time_small" in what:
    timing(small_only=True)
if "time_ring" in what:
    timing_ring()
if "time_persist_shape" in what:
    for (tag, M, N, K, bits, dt) in (("big W4", 1, 28672, 8192, 4, f16), ("down proj 70B", 1, 8192, 28672, 4, f16), ("big W3", 1, 28672, 8192, 3, bf16),
                                     ("8192 g64", 1, 8192, 8192, 4, f16), ("14336x8192", 1, 14336, 8192, 4, f16), ("W2 big", 1, 28672, 8192, 2, f16),
                                     ("8192x14336", 1, 8192, 14336, 4, f16)):
        for w in (4, 5, 6, 7, 8):
            for c in (1, 2, 3, 4):
                if w * c <= 16:
                    time_case(M, N, K, bits, 64, dt, dict(one_shot=2, waves=w, m_tiles=c), steps=200, tag=tag)
if "time_persist_rows" in what:
    for (tag, N, K, bits, dt) in (("28672x8192", 28672, 8192, 4, f16), ("8192x28672", 8192, 28672, 4, f16), ("14336x4096", 14336, 4096, 4, f16),
                                  ("8192^2", 8192, 8192, 4, f16), ("28672x8192 W3", 28672, 8192, 3, bf16), ("28672x8192 W2", 28672, 8192, 2, f16)):
        for M in (1, 2, 3, 4):
            if bits == 3 and M > 2:
                continue
            for shp in (dict(), dict(family=0, one_shot=0), dict(family=0, one_shot=2), dict(family=2), dict(family=5)):
                time_case(M, N, K, bits, 64, dt, shp, steps=200, tag=f"{tag} M={M}")
if "time_persist" in what:
    timing_persist()
os.makedirs("gpurun_out", exist_ok=True)
json.dump([r for r in rows if r.get("kind") != "check" or not r.get("ok")], open("gpurun_out/decode_lab.json", "w"), indent=1)
print(f"decode_lab done in {time.time() - t0:.1f}s, failures: {rc}")
sys.exit(1 if rc else 0)
