"""Round-6 scratch lab: timing cases for the split-K block kernel's variants (reads R06_CASE)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import splitk_lab as L  # noqa: E402

f16, bf16 = torch.float16, torch.bfloat16
case = os.environ.get("R06_CASE", "xcd")
tag = os.path.basename(os.environ.get("FLUTE_AMD_LIB", "shipped"))
if case == "xcd":
    for rep in range(2):
        for mb in (1, 2, 4):
            L.time_one(256, 4096, 4096, 4, f16, dict(family=6, splitk=1, kw=4, m_block=mb), tag=f"xcd_group_{mb}")
    for mb in (1, 2, 4, 8):
        L.time_one(512, 2048, 4096, 4, f16, dict(family=6, splitk=1, kw=4, m_block=mb), tag=f"xcd_group_{mb}")
    for mb in (1, 2):
        L.time_one(128, 8192, 4096, 4, f16, dict(family=6, splitk=1, kw=4, m_block=mb), tag=f"xcd_group_{mb}")
elif case == "abl":
    L.time_one(256, 4096, 4096, 4, f16, dict(family=6, splitk=1, kw=4), tag=tag)
    L.time_one(256, 4096, 4096, 4, f16, dict(family=6, splitk=2, kw=2, m_tiles=4), tag=tag)
elif case == "prefill":   # throughput-bound launches of the 128 x 128 tiles: ablation builds against the automatic plan
    for M in (1024, 2048, 4096):
        L.time_one(M, 4096, 4096, 4, f16, dict(family=6, splitk=1, kw=2, m_tiles=8), steps=60, tag=tag)
        if tag == "shipped":
            L.time_one(M, 4096, 4096, 4, f16, None, steps=60, tag="tuned table")
elif case == "block2":    # the 2- / 4-bit block kernel's ablation builds (-DFLUTE_B2_ABLATE=N): 256- and 128-row blocks
    L.time_one(4096, 4096, 4096, 4, f16, dict(family=3, m_tiles=16), steps=40, tag=tag)
    L.time_one(4096, 4096, 4096, 4, bf16, dict(family=3, m_tiles=16), steps=40, tag=tag)
    L.time_one(2048, 4096, 4096, 4, f16, dict(family=3, m_tiles=8), steps=40, tag=tag)
    L.time_one(8192, 4096, 4096, 4, f16, dict(family=3, m_tiles=16), steps=20, tag=tag)
elif case == "m256":
    L.time_one(256, 4096, 4096, 4, f16, dict(family=6, splitk=1, kw=4), tag=tag)
    L.time_one(256, 4096, 4096, 4, bf16, dict(family=6, splitk=1, kw=4), tag=tag)
    L.time_one(128, 8192, 4096, 4, f16, dict(family=6, splitk=1, kw=4), tag=tag)
    L.time_one(256, 11008, 4096, 4, f16, dict(family=6, splitk=1, kw=2, m_tiles=8), tag=tag)
elif case == "auto":      # the automatic plan against the forced alternatives on shapes the round-6 model extrapolates to
    for (M, N, K) in ((48, 3584, 14336), (64, 8192, 8192), (33, 8192, 8192), (96, 14336, 4096), (80, 8192, 8192), (128, 4096, 4096), (192, 4096, 4096),
                      (256, 4096, 4096), (512, 2048, 4096), (256, 2048, 8192), (128, 11008, 4096), (64, 4096, 4096), (384, 4096, 4096), (512, 4096, 4096)):
        L.time_one(M, N, K, 4, f16, None, tag="tuned table")
        L.time_one(M, N, K, 4, f16, dict(), tag="automatic (id 16)")
        L.time_one(M, N, K, 4, f16, dict(family=2), tag="per-wave")
        L.time_one(M, N, K, 4, f16, dict(family=6, kw=4), tag="kp4 best")
        L.time_one(M, N, K, 4, f16, dict(family=6, kw=2), tag="kp2 best")
elif case == "fastm":     # lean MFMA decode kernel with NG column groups per workgroup: check against fp32 + one-hot rows, then time
    import json
    from flute_amd import dev, utils
    import flute_amd
    d = L.d
    nfail = 0
    for (tile_p, g, dtype, K, N) in ((32, 64, f16, 4096, 4096), (32, 64, f16, 4096, 11008), (64, 128, bf16, 4096, 8192), (32, 256, f16, 4096, 5120),
                                     (32, 64, bf16, 2048, 6144), (64, 64, f16, 2048, 1296 * 4), (32, 64, f16, 4096, 208)):
        torch.manual_seed(K + N)
        if N % (4 * tile_p):
            continue
        W = torch.randint(0, 16, (K, N), dtype=torch.uint8, device=d)
        S = torch.randn(N, K // g, device=d).to(dtype)
        table = torch.randn(16, device=d).to(dtype)
        table2 = utils.make_qmap2_from_qmap(table)
        tid = L.tid_of(4, tile_p)
        Q = utils.pack(W, 4, [tid], L.num_sms)
        What = table[W.long()] * torch.repeat_interleave(S, g, dim=1).T
        for M in (5, 11, 16):
            X = (torch.randn(M, K, device=d) / 100).to(dtype)
            ref = X.float() @ What.float()
            ks = torch.randint(0, K, (M,), device=d)
            E = torch.zeros(M, K, device=d, dtype=dtype)
            E[torch.arange(M, device=d), ks] = 1
            for ng in (1, 2, 3):
                ovr = dev.Overrides(family=7, slabs_per_wave=ng)
                pl = dev.get_plan(M, N, K, 4, g, tid, L.num_sms, dtype, ovr)
                if pl["family"] != 7:
                    continue
                o = dev.qgemm_planned(X, Q, S, table, table2, L.ws, 4, g, tid, L.num_sms, ovr)
                o1 = dev.qgemm_planned(E, Q, S, table, table2, L.ws, 4, g, tid, L.num_sms, ovr)
                err = ((o.float() - ref).norm() / ref.norm()).item()
                ok = err < (1e-3 if dtype == f16 else 4e-3) and bool(torch.equal(o1, What[ks]))
                if not ok:
                    nfail += 1
                    print(json.dumps({"kind": "check_fastm", "N": N, "K": K, "g": g, "dtype": str(dtype), "M": M, "ng": ng, "grid": pl["grid"], "err": err,
                                      "onehot": bool(torch.equal(o1, What[ks]))}), flush=True)
        del W, S, Q, What
        torch.cuda.empty_cache()
    print(json.dumps({"kind": "check_fastm_summary", "failed": nfail}), flush=True)
    for (M, N, K) in ((16, 4096, 4096), (16, 11008, 4096), (16, 8192, 4096), (16, 6144, 4096), (16, 14336, 4096), (8, 11008, 4096), (16, 5120, 4096), (16, 8192, 2048),
                      (16, 28672, 4096)):
        L.time_one(M, N, K, 4, f16, None, steps=300, tag="tuned table")
        for ng in (1, 2, 3):
            L.time_one(M, N, K, 4, f16, dict(family=7, slabs_per_wave=ng), steps=300, tag=f"fastm ng={ng}")
elif case == "persistm":  # persistent MFMA decode kernel (family 8): check against fp32 + one-hot rows, then time against the table's plan
    import json
    from flute_amd import dev, utils
    d = L.d
    nfail = 0
    if os.environ.get("R06_SKIP_CHECK") != "1":
        for (tile_p, g, dtype, K, N) in ((32, 64, f16, 8192, 8192), (32, 64, f16, 4096, 14336), (64, 128, bf16, 8192, 3584), (32, 128, f16, 3584, 4096),
                                         (32, 64, bf16, 11008, 4096), (64, 64, f16, 1024, 1296 * 4), (32, 64, f16, 14336, 4096), (32, 128, bf16, 1280, 5248), (32, 64, f16, 1152, 2048), (32, 64, f16, 4096, 4096), (64, 128, bf16, 2048, 2048)):
            torch.manual_seed(K + N)
            if N % (4 * tile_p):
                continue
            W = torch.randint(0, 16, (K, N), dtype=torch.uint8, device=d)
            S = torch.randn(N, K // g, device=d).to(dtype)
            table = torch.randn(16, device=d).to(dtype)
            table2 = utils.make_qmap2_from_qmap(table)
            tid = L.tid_of(4, tile_p)
            Q = utils.pack(W, 4, [tid], L.num_sms)
            What = table[W.long()] * torch.repeat_interleave(S, g, dim=1).T
            for M in (3, 5, 11, 16):
                X = (torch.randn(M, K, device=d) / 100).to(dtype)
                ref = X.float() @ What.float()
                ks = torch.randint(0, K, (M,), device=d)
                E = torch.zeros(M, K, device=d, dtype=dtype)
                E[torch.arange(M, device=d), ks] = 1
                for ng in (1, 2, 3):
                    for vis in (-1, 1, 3, -16):
                        ovr = dev.Overrides(family=8, slabs_per_wave=ng, m_tiles=vis) if vis != -16 else dev.Overrides(family=8, slabs_per_wave=ng, one_shot=0)
                        try:
                            pl = dev.get_plan(M, N, K, 4, g, tid, L.num_sms, dtype, ovr)
                        except Exception as e:  # noqa: BLE001
                            print(json.dumps({"kind": "check_persistm", "N": N, "K": K, "ng": ng, "vis": vis, "error": str(e)[:100]}), flush=True)
                            nfail += 1
                            continue
                        o = dev.qgemm_planned(X, Q, S, table, table2, L.ws, 4, g, tid, L.num_sms, ovr)
                        o1 = dev.qgemm_planned(E, Q, S, table, table2, L.ws, 4, g, tid, L.num_sms, ovr)
                        err = ((o.float() - ref).norm() / ref.norm()).item()
                        ok = err < (1e-3 if dtype == f16 else 4e-3) and bool(torch.equal(o1, What[ks]))
                        if not ok:
                            nfail += 1
                            print(json.dumps({"kind": "check_persistm", "N": N, "K": K, "g": g, "dtype": str(dtype), "M": M, "ng": ng, "vis": vis, "grid": pl["grid"],
                                              "err": err, "onehot": bool(torch.equal(o1, What[ks]))}), flush=True)
            del W, S, Q, What
            torch.cuda.empty_cache()
        print(json.dumps({"kind": "check_persistm_summary", "failed": nfail}), flush=True)
    shapes = ((28672, 8192), (8192, 28672), (8192, 8192), (14336, 4096), (4096, 14336), (10240, 8192), (4096, 11008), (3584, 14336), (11008, 4096), (4096, 4096))
    for (N, K) in shapes:
        for M in (4, 8, 16):
            L.time_one(M, N, K, 4, f16, None, steps=200, tag="tuned table")
            for ng in (1, 2, 3):
                L.time_one(M, N, K, 4, f16, dict(family=8, slabs_per_wave=ng), steps=200, tag=f"persistm ng={ng}")
            for ng in (1, 2):
                L.time_one(M, N, K, 4, f16, dict(family=8, slabs_per_wave=ng, one_shot=0), steps=200, tag=f"persistm ng={ng}rings")
elif case == "persistm2":  # the 2-bit member of the persistent MFMA decode kernel: check against fp32 + one-hot rows, then time against the table's plan
    import json
    from flute_amd import dev, utils
    d = L.d
    nfail = 0
    for (tile_p, g, dtype, K, N) in ((32, 64, f16, 8192, 8192), (32, 64, f16, 4096, 14336), (64, 128, bf16, 8192, 3584), (32, 128, f16, 3584, 4096),
                                     (32, 64, bf16, 11008, 4096), (64, 64, f16, 1024, 1024 * 5), (32, 64, f16, 14336, 4096), (32, 128, bf16, 1280, 5376),
                                     (32, 64, f16, 1152, 2048), (32, 64, f16, 4096, 4096)):
        torch.manual_seed(K + N)
        if N % (8 * tile_p):
            continue
        W = torch.randint(0, 4, (K, N), dtype=torch.uint8, device=d)
        S = torch.randn(N, K // g, device=d).to(dtype)
        table = torch.randn(4, device=d).to(dtype)
        table2 = utils.make_qmap2_from_qmap(table)
        tid = L.tid_of(2, tile_p)
        Q = utils.pack(W, 2, [tid], L.num_sms)
        What = table[W.long()] * torch.repeat_interleave(S, g, dim=1).T
        for M in (3, 5, 11, 16):
            X = (torch.randn(M, K, device=d) / 100).to(dtype)
            ref = X.float() @ What.float()
            ks = torch.randint(0, K, (M,), device=d)
            E = torch.zeros(M, K, device=d, dtype=dtype)
            E[torch.arange(M, device=d), ks] = 1
            for ng in (1, 2, 3):
                for (vis, res) in ((-1, -1), (3, -1), (-1, 0)):
                    ovr = dev.Overrides(family=8, slabs_per_wave=ng, m_tiles=vis, one_shot=res)
                    try:
                        pl = dev.get_plan(M, N, K, 2, g, tid, L.num_sms, dtype, ovr)
                    except Exception as e:  # noqa: BLE001
                        print(json.dumps({"kind": "check_persistm2", "N": N, "K": K, "ng": ng, "vis": vis, "error": str(e)[:100]}), flush=True)
                        nfail += 1
                        continue
                    o = dev.qgemm_planned(X, Q, S, table, table2, L.ws, 2, g, tid, L.num_sms, ovr)
                    o1 = dev.qgemm_planned(E, Q, S, table, table2, L.ws, 2, g, tid, L.num_sms, ovr)
                    err = ((o.float() - ref).norm() / ref.norm()).item()
                    ok = err < (1e-3 if dtype == f16 else 4e-3) and bool(torch.equal(o1, What[ks]))
                    if not ok:
                        nfail += 1
                        print(json.dumps({"kind": "check_persistm2", "N": N, "K": K, "g": g, "dtype": str(dtype), "M": M, "ng": ng, "vis": vis, "res": res, "grid": pl["grid"],
                                          "err": err, "onehot": bool(torch.equal(o1, What[ks]))}), flush=True)
        del W, S, Q, What
        torch.cuda.empty_cache()
    print(json.dumps({"kind": "check_persistm2_summary", "failed": nfail}), flush=True)
    shapes = ((28672, 8192), (8192, 28672), (8192, 8192), (14336, 4096), (4096, 14336), (10240, 8192), (4096, 11008), (3584, 14336), (11008, 4096), (4096, 4096))
    for (N, K) in shapes:
        for M in (4, 8, 16):
            L.time_one(M, N, K, 2, f16, None, steps=200, tag="tuned table")
            for ng in (1, 2, 3):
                L.time_one(M, N, K, 2, f16, dict(family=8, slabs_per_wave=ng), steps=200, tag=f"persistm ng={ng}")
elif case == "persistm_abl":   # ablation builds of the persistent MFMA decode kernel (-DFLUTE_PM_ABLATE=N)
    for (N, K, ng) in ((28672, 8192, 1), (8192, 28672, 2), (4096, 14336, 1)):
        for M in (4, 16):
            L.time_one(M, N, K, 4, f16, dict(family=8, slabs_per_wave=ng), steps=200, tag=tag)
elif case == "m256x":     # M = 256 on 4096^2 and its neighbours: XCD group size and request distance
    for mb in (1, 2, 4):
        L.time_one(256, 4096, 4096, 4, f16, dict(family=6, splitk=1, kw=4, m_block=mb), tag=f"{tag} xcd_group_{mb}")
    L.time_one(256, 4096, 4096, 4, bf16, dict(family=6, splitk=1, kw=4), tag=tag)
    L.time_one(128, 4096, 4096, 4, f16, dict(family=6, splitk=2, kw=4), tag=tag)
    L.time_one(256, 11008, 4096, 4, f16, dict(family=6, splitk=1, kw=2, m_tiles=8), tag=tag)
    L.time_one(1024, 4096, 4096, 4, f16, dict(family=6, splitk=1, kw=2, m_tiles=8), tag=tag)
    L.time_one(256, 8192, 8192, 4, f16, dict(family=6, splitk=2, kw=2, m_tiles=8), tag=tag)
