"""Split-K block kernel lab (qgemm_splitk.h, family 6): correctness (fp32 reference on the GPU, one-hot rows exact, the
tile state words zero after every call, repeated launches bit-identical) and HBM-cold timing against the automatic plan
and torch.mm.  Writes gpurun_out/splitk_lab.jsonl.   python tools/splitk_lab.py [check] [stress] [time] [time_more]"""
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
what = sys.argv[1:] or ["check", "stress", "time"]
os.makedirs("gpurun_out", exist_ok=True)
out = open("gpurun_out/splitk_lab.jsonl", "a")


def tid_of(bits, tile_p):
    return min(t for (b, t), c in flute_amd.TEMPLATE_CONFIGS.items() if b == bits and c["TileP"] == tile_p)


def emit(r):
    s = json.dumps(r)
    print(s, flush=True)
    out.write(s + "\n")
    out.flush()


def state_clean():
    return int(ws[:65536].view(torch.int32).abs().sum().item()) == 0


def check():
    nfail = 0
    cases = [(4, 32, 64, f16, 4096, 4096), (4, 64, 64, bf16, 2048, 1024), (4, 32, 128, f16, 3072, 512), (4, 32, 32, f16, 1024, 256),
             (4, 64, 256, bf16, 4096, 256), (4, 32, 64, f16, 4096, 11008), (2, 32, 64, f16, 4096, 2048), (2, 64, 128, bf16, 2048, 1024),
             (2, 32, 32, bf16, 1024, 256), (4, 32, 64, bf16, 8192, 1024)]
    for (bits, tile_p, g, dtype, K, N) in cases:
        torch.manual_seed(K + N)
        W = torch.randint(0, 2 ** bits, (K, N), dtype=torch.uint8, device=d)
        S = torch.randn(N, K // g, device=d).to(dtype)
        table = torch.randn(2 ** bits, device=d).to(dtype)
        table2 = utils.make_qmap2_from_qmap(table)
        tid = tid_of(bits, tile_p)
        Q = utils.pack(W, bits, [tid], num_sms)
        What = table[W.long()] * torch.repeat_interleave(S, g, dim=1).T
        tol = 1e-3 if dtype == f16 else 8e-3
        for M in (1, 100, 128, 200, 256, 300, 1024):
            X = (torch.randn(M, K, device=d) / 100).to(dtype)
            ref = X.float() @ What.float()
            ks = torch.randint(0, K, (M,), device=d)
            E = torch.zeros(M, K, device=d, dtype=dtype)
            E[torch.arange(M, device=d), ks] = 1
            for sk in (1, 2, 3, 4, 6, 8, 16):
                shp = dict(family=6, splitk=sk)
                rec = {"kind": "check", "bits": bits, "tile_p": tile_p, "g": g, "dtype": str(dtype)[6:], "K": K, "N": N, "M": M, "splitk": sk}
                try:
                    ovr = dev.Overrides(**shp)
                    try:
                        pl = dev.get_plan(M, N, K, bits, g, tid, num_sms, dtype, ovr)
                    except RuntimeError:
                        continue                                   # not a legal split of this K
                    o = dev.qgemm_planned(X, Q, S, table, table2, ws, bits, g, tid, num_sms, ovr)
                    o1 = dev.qgemm_planned(E, Q, S, table, table2, ws, bits, g, tid, num_sms, ovr)
                    o2 = dev.qgemm_planned(X, Q, S, table, table2, ws, bits, g, tid, num_sms, ovr)
                    torch.cuda.synchronize()
                    err = ((o.float() - ref).norm() / ref.norm()).item()
                    exact = bool(torch.equal(o1, What[ks]))
                    same = bool(torch.equal(o, o2))
                    clean = state_clean()
                    rec.update(err=err, onehot_exact=exact, repeat_identical=same, state_clean=clean, grid=pl["grid"],
                               ok=bool(err < tol and exact and same and clean))
                    if not rec["ok"]:
                        bad = ((o.float() - ref).abs() > 0.05 * ref.abs().max()) | o.float().isnan()
                        rec["nbad"] = int(bad.sum().item())
                        rec["bad_rows"] = bad.any(1).nonzero().flatten()[:12].tolist()
                        rec["bad_cols"] = bad.any(0).nonzero().flatten()[:16].tolist()
                        rec["onehot_mismatch"] = int((o1 != What[ks]).sum().item())
                        if not clean:
                            ws[:65536].zero_()
                except Exception as ex:  # noqa: BLE001
                    rec.update(ok=False, error=str(ex)[:300])
                if not rec["ok"]:
                    nfail += 1
                emit(rec)
        del W, S, Q, What
        torch.cuda.empty_cache()
    emit({"kind": "check_summary", "failed": nfail})
    return nfail


def stress():
    """Uneven load: the split-K launches of several shapes back to back in one graph (workgroups of neighbouring launches
    overlap at the seams), every result compared word for word with the first one."""
    nfail = 0
    for (M, N, K, sk, dtype) in ((256, 4096, 4096, 4, f16), (256, 4096, 4096, 8, f16), (200, 11008, 4096, 4, f16), (256, 4096, 4096, 2, bf16),
                                 (512, 4096, 4096, 4, f16), (128, 4096, 4096, 8, bf16), (256, 2048, 8192, 16, f16)):
        lay = bench.Layer(M, N, K, 4, 64, dtype, d, 3)
        lay.template_id = tid_of(4, 32)
        lay.ovr = dev.Overrides(family=6, splitk=sk)
        first = [lay.step(c).clone() for c in range(3)]
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        outs = []
        with torch.cuda.graph(graph):
            for i in range(60):
                outs.append(lay.step(i))
        nbad = 0
        for rep in range(5):
            graph.replay()
            torch.cuda.synchronize()
            nbad += sum(0 if torch.equal(o, first[i % 3]) else 1 for i, o in enumerate(outs))
        Wref = None
        rec = {"kind": "stress", "M": M, "N": N, "K": K, "splitk": sk, "dtype": str(dtype)[6:], "launches": 300, "mismatching": nbad,
               "state_clean": state_clean(), "ok": nbad == 0 and state_clean()}
        if not rec["ok"]:
            nfail += 1
            ws[:65536].zero_()
        emit(rec)
        del lay, outs, graph
        torch.cuda.empty_cache()
    emit({"kind": "stress_summary", "failed": nfail})
    return nfail


def time_one(M, N, K, bits, dtype, shp, steps=200, tag=None):
    lay = bench.Layer(M, N, K, bits, 64, dtype, d, bench.copies_for(N, K, bits))
    lay.template_id = tid_of(bits, 32)
    rec = {"kind": "time", "bits": bits, "M": M, "N": N, "K": K, "dtype": str(dtype)[6:], "shape": shp}
    if tag:
        rec["tag"] = tag
    try:
        if shp is None:
            lay.tune()                                             # the shipped table / tuner: what flute.qgemm runs today
        else:
            lay.ovr = dev.Overrides(**shp)
        pl = dev.get_plan(M, N, K, bits, 64, lay.template_id, num_sms, dtype, lay.ovr)
        rec["plan"] = {k: pl[k] for k in ("family", "m_block", "m_tiles", "waves", "kw", "splitk", "grid")}
        ms = min(bench.time_graph(lay, steps, 5, torch.cuda.synchronize)[0] for _ in range(2))
        us = ms / steps * 1e3
        rec.update(us=round(us, 2), TFLOPs=round(lay.flops() / us / 1e6, 1), frac=round(lay.flops() / us / 1e6 / 2500, 3))
    except Exception as ex:  # noqa: BLE001
        rec["error"] = str(ex)[:200]
    emit(rec)
    del lay
    torch.cuda.empty_cache()


def time_mm(M, N, K):
    Wd = [torch.randn(K, N, device=d, dtype=f16) for _ in range(max(2, (300 << 20) // (K * N * 2) + 1))]
    Xd = torch.randn(M, K, device=d, dtype=f16)

    class Dense:
        def step(self, i):
            return torch.mm(Xd, Wd[i % len(Wd)])
    ms, _ = bench.time_graph(Dense(), 100, 5, torch.cuda.synchronize)
    us = ms / 100 * 1e3
    emit({"kind": "time", "M": M, "N": N, "K": K, "shape": "torch.mm fp16", "us": round(us, 2), "TFLOPs": round(2 * M * N * K / us / 1e6, 1)})
    del Wd, Xd
    torch.cuda.empty_cache()


def timing(more=False):
    shapes = [(256, 4096, 4096), (256, 11008, 4096)]
    if more:
        shapes += [(128, 4096, 4096), (512, 4096, 4096), (1024, 4096, 4096), (256, 14336, 4096), (256, 4096, 14336), (256, 8192, 8192),
                   (512, 11008, 4096), (128, 11008, 4096), (256, 28672, 8192)]
    for (M, N, K) in shapes:
        time_one(M, N, K, 4, f16, None)
        for sk in (1, 2, 4, 8, 16):
            try:
                dev.get_plan(M, N, K, 4, 64, tid_of(4, 32), num_sms, f16, dev.Overrides(family=6, splitk=sk))
            except RuntimeError:
                continue
            time_one(M, N, K, 4, f16, dict(family=6, splitk=sk))
        if M >= 256:
            time_one(M, N, K, 4, f16, dict(family=3, m_tiles=4))
        time_mm(M, N, K)
    if more:
        for (M, N, K) in ((2048, 4096, 4096), (4096, 4096, 4096), (1024, 11008, 4096), (2048, 11008, 4096), (1024, 8192, 8192), (512, 28672, 8192)):
            time_one(M, N, K, 4, f16, None, steps=100)
            time_one(M, N, K, 4, f16, dict(family=6, splitk=1), steps=100)
            time_one(M, N, K, 4, f16, dict(family=3, m_tiles=4), steps=100)
            time_one(M, N, K, 4, f16, dict(family=3, m_tiles=8), steps=100)
            time_mm(M, N, K)
    time_one(256, 4096, 4096, 4, bf16, dict(family=6, splitk=4))
    time_one(256, 4096, 4096, 2, f16, dict(family=6, splitk=4))


def check_skinny():
    """Skinny MFMA kernel (family 5) with a grid-level K split (xwg.h, L form)."""
    nfail = 0
    cases = [(32, 64, f16, 4096, 4096), (64, 64, bf16, 2048, 1024), (32, 128, f16, 4096, 512), (32, 32, f16, 1024, 256), (32, 64, f16, 8192, 2048),
             (32, 64, bf16, 14336, 4096)]
    for (tile_p, g, dtype, K, N) in cases:
        torch.manual_seed(K + N)
        W = torch.randint(0, 16, (K, N), dtype=torch.uint8, device=d)
        S = torch.randn(N, K // g, device=d).to(dtype)
        table = torch.randn(16, device=d).to(dtype)
        table2 = utils.make_qmap2_from_qmap(table)
        tid = tid_of(4, tile_p)
        Q = utils.pack(W, 4, [tid], num_sms)
        What = table[W.long()] * torch.repeat_interleave(S, g, dim=1).T
        tol = 1e-3 if dtype == f16 else 8e-3
        for M in (3, 5, 16):
            X = (torch.randn(M, K, device=d) / 100).to(dtype)
            ref = X.float() @ What.float()
            ks = torch.randint(0, K, (M,), device=d)
            E = torch.zeros(M, K, device=d, dtype=dtype)
            E[torch.arange(M, device=d), ks] = 1
            for sk in (1, 2, 4, 7, 8, 14, 16):
                for waves in (-1, 4):
                    rec = {"kind": "check_skinny", "tile_p": tile_p, "g": g, "dtype": str(dtype)[6:], "K": K, "N": N, "M": M, "splitk": sk, "waves": waves}
                    try:
                        ovr = dev.Overrides(family=5, splitk=sk, waves=waves)
                        try:
                            pl = dev.get_plan(M, N, K, 4, g, tid, num_sms, dtype, ovr)
                        except RuntimeError:
                            continue
                        if pl["family"] != 5 or pl["splitk"] != sk:
                            continue
                        o = dev.qgemm_planned(X, Q, S, table, table2, ws, 4, g, tid, num_sms, ovr)
                        o1 = dev.qgemm_planned(E, Q, S, table, table2, ws, 4, g, tid, num_sms, ovr)
                        o2 = dev.qgemm_planned(X, Q, S, table, table2, ws, 4, g, tid, num_sms, ovr)
                        torch.cuda.synchronize()
                        err = ((o.float() - ref).norm() / ref.norm()).item()
                        exact = bool(torch.equal(o1, What[ks])) if dtype == f16 else bool(((o1.float() - What[ks].float()).abs() <= 8e-3 * What[ks].float().abs() + 1e-6).all())
                        same = bool(torch.equal(o, o2))
                        clean = state_clean()
                        rec.update(err=err, onehot_exact=exact, repeat_identical=same, state_clean=clean, grid=pl["grid"], depth=pl["ring_depth"], kw=pl["kw"],
                                   ok=bool(err < tol and exact and same and clean))
                        if not clean:
                            ws[:65536].zero_()
                    except Exception as ex:  # noqa: BLE001
                        rec.update(ok=False, error=str(ex)[:300])
                    if not rec["ok"]:
                        nfail += 1
                    emit(rec)
        del W, S, Q, What
        torch.cuda.empty_cache()
    emit({"kind": "check_skinny_summary", "failed": nfail})
    return nfail


def time_skinny():
    for (M, N, K) in ((16, 4096, 4096), (4, 4096, 4096), (16, 8192, 4096), (16, 6144, 4096), (16, 11008, 4096), (16, 4096, 14336), (16, 8192, 8192), (16, 2048, 8192)):
        time_one(M, N, K, 4, f16, None)
        for sk in (1, 2, 4, 7, 8, 14):
            try:
                pl = dev.get_plan(M, N, K, 4, 64, tid_of(4, 32), num_sms, f16, dev.Overrides(family=5, splitk=sk))
            except RuntimeError:
                continue
            if pl["family"] == 5 and pl["splitk"] == sk:
                time_one(M, N, K, 4, f16, dict(family=5, splitk=sk))


def check_ldw():
    """The loader-wave variant (override waves = 12) on a subset of check()'s cases."""
    nfail = 0
    for (bits, tile_p, g, dtype, K, N) in [(4, 32, 64, f16, 4096, 4096), (4, 64, 64, bf16, 2048, 1024), (4, 32, 128, f16, 3072, 512),
                                           (2, 32, 64, f16, 4096, 2048), (2, 64, 128, bf16, 2048, 1024), (4, 32, 32, f16, 1024, 256)]:
        torch.manual_seed(K + N)
        W = torch.randint(0, 2 ** bits, (K, N), dtype=torch.uint8, device=d)
        S = torch.randn(N, K // g, device=d).to(dtype)
        table = torch.randn(2 ** bits, device=d).to(dtype)
        table2 = utils.make_qmap2_from_qmap(table)
        tid = tid_of(bits, tile_p)
        Q = utils.pack(W, bits, [tid], num_sms)
        What = table[W.long()] * torch.repeat_interleave(S, g, dim=1).T
        tol = 1e-3 if dtype == f16 else 8e-3
        for M in (1, 130, 256, 700):
            X = (torch.randn(M, K, device=d) / 100).to(dtype)
            ref = X.float() @ What.float()
            ks = torch.randint(0, K, (M,), device=d)
            E = torch.zeros(M, K, device=d, dtype=dtype)
            E[torch.arange(M, device=d), ks] = 1
            for sk in (1, 2, 3, 4, 8):
                rec = {"kind": "check_ldw", "bits": bits, "tile_p": tile_p, "g": g, "dtype": str(dtype)[6:], "K": K, "N": N, "M": M, "splitk": sk}
                try:
                    ovr = dev.Overrides(family=6, splitk=sk, waves=12)
                    try:
                        pl = dev.get_plan(M, N, K, bits, g, tid, num_sms, dtype, ovr)
                    except RuntimeError:
                        continue
                    assert pl["waves"] == 12 and pl["block"] == 768
                    o = dev.qgemm_planned(X, Q, S, table, table2, ws, bits, g, tid, num_sms, ovr)
                    o1 = dev.qgemm_planned(E, Q, S, table, table2, ws, bits, g, tid, num_sms, ovr)
                    o2 = dev.qgemm_planned(X, Q, S, table, table2, ws, bits, g, tid, num_sms, ovr)
                    torch.cuda.synchronize()
                    err = ((o.float() - ref).norm() / ref.norm()).item()
                    rec.update(err=err, onehot_exact=bool(torch.equal(o1, What[ks])), repeat_identical=bool(torch.equal(o, o2)), state_clean=state_clean())
                    rec["ok"] = bool(err < tol and rec["onehot_exact"] and rec["repeat_identical"] and rec["state_clean"])
                    if not rec["state_clean"]:
                        ws[:65536].zero_()
                except Exception as ex:  # noqa: BLE001
                    rec.update(ok=False, error=str(ex)[:300])
                if not rec["ok"]:
                    nfail += 1
                emit(rec)
        del W, S, Q, What
        torch.cuda.empty_cache()
    emit({"kind": "check_ldw_summary", "failed": nfail})
    return nfail


def time_ldw():
    for (M, N, K, sk) in ((256, 11008, 4096, 1), (1024, 4096, 4096, 1), (256, 4096, 4096, 1), (256, 4096, 4096, 2), (256, 4096, 4096, 4), (256, 14336, 4096, 1),
                          (256, 8192, 8192, 2), (512, 4096, 4096, 2), (128, 11008, 4096, 2)):
        for waves in (-1, 12):
            time_one(M, N, K, 4, f16, dict(family=6, splitk=sk, waves=waves))
    time_one(256, 11008, 4096, 4, bf16, dict(family=6, splitk=1, waves=12))
    time_one(256, 11008, 4096, 2, f16, dict(family=6, splitk=1, waves=12))


def time_had():
    """What the UNFUSED rotation costs at 5 <= M <= 16 (SURVEY f-3): flute.qgemm_hadamard(h = 512) against flute.qgemm on the
    Gemma-2-9B shapes - the difference is the separate flute_hadamard launch a fused prologue would save at most."""
    for (N, K) in ((4096, 3584), (14336, 3584), (3584, 14336), (4096, 4096)):
        for M in (1, 4, 16):
            us = {}
            for had in (0, 512):
                lay = bench.Layer(M, N, K, 4, 64, f16, d, bench.copies_for(N, K, 4), hadamard_size=had)
                lay.tune()
                ms = min(bench.time_graph(lay, 300, 5, torch.cuda.synchronize)[0] for _ in range(2))
                us[had] = round(ms / 300 * 1e3, 2)
                del lay
                torch.cuda.empty_cache()
            emit({"kind": "time_had", "M": M, "N": N, "K": K, "qgemm_us": us[0], "qgemm_hadamard_us": us[512], "rotation_us": round(us[512] - us[0], 2)})


def check_rt4():
    """64-row tiles (override m_tiles = 4) on a subset of check()'s cases."""
    nfail = 0
    for (bits, tile_p, g, dtype, K, N) in [(4, 32, 64, f16, 4096, 4096), (4, 64, 64, bf16, 2048, 1024), (4, 32, 128, f16, 3072, 512),
                                           (2, 32, 64, f16, 4096, 2048), (2, 64, 128, bf16, 2048, 1024), (4, 32, 32, f16, 1024, 256)]:
        torch.manual_seed(K + N)
        W = torch.randint(0, 2 ** bits, (K, N), dtype=torch.uint8, device=d)
        S = torch.randn(N, K // g, device=d).to(dtype)
        table = torch.randn(2 ** bits, device=d).to(dtype)
        table2 = utils.make_qmap2_from_qmap(table)
        tid = tid_of(bits, tile_p)
        Q = utils.pack(W, bits, [tid], num_sms)
        What = table[W.long()] * torch.repeat_interleave(S, g, dim=1).T
        tol = 1e-3 if dtype == f16 else 8e-3
        for M in (1, 60, 64, 130, 256, 700):
            X = (torch.randn(M, K, device=d) / 100).to(dtype)
            ref = X.float() @ What.float()
            ks = torch.randint(0, K, (M,), device=d)
            E = torch.zeros(M, K, device=d, dtype=dtype)
            E[torch.arange(M, device=d), ks] = 1
            for sk in (1, 2, 3, 4, 8):
                for waves in (-1,):
                    rec = {"kind": "check_rt4", "bits": bits, "tile_p": tile_p, "g": g, "dtype": str(dtype)[6:], "K": K, "N": N, "M": M, "splitk": sk, "waves": waves}
                    try:
                        ovr = dev.Overrides(family=6, splitk=sk, m_tiles=4, waves=waves, kw=2)
                        try:
                            pl = dev.get_plan(M, N, K, bits, g, tid, num_sms, dtype, ovr)
                        except RuntimeError:
                            continue
                        assert pl["m_tiles"] == 4 and pl["grid"] == -(-M // 64) * (N // 128) * sk, pl
                        o = dev.qgemm_planned(X, Q, S, table, table2, ws, bits, g, tid, num_sms, ovr)
                        o1 = dev.qgemm_planned(E, Q, S, table, table2, ws, bits, g, tid, num_sms, ovr)
                        o2 = dev.qgemm_planned(X, Q, S, table, table2, ws, bits, g, tid, num_sms, ovr)
                        torch.cuda.synchronize()
                        err = ((o.float() - ref).norm() / ref.norm()).item()
                        rec.update(err=err, onehot_exact=bool(torch.equal(o1, What[ks])), repeat_identical=bool(torch.equal(o, o2)), state_clean=state_clean())
                        rec["ok"] = bool(err < tol and rec["onehot_exact"] and rec["repeat_identical"] and rec["state_clean"])
                        if not rec["state_clean"]:
                            ws[:65536].zero_()
                    except Exception as ex:  # noqa: BLE001
                        rec.update(ok=False, error=str(ex)[:300])
                    if not rec["ok"]:
                        nfail += 1
                    emit(rec)
        del W, S, Q, What
        torch.cuda.empty_cache()
    emit({"kind": "check_rt4_summary", "failed": nfail})
    return nfail


def time_rt4():
    for (M, N, K) in ((256, 4096, 4096), (128, 4096, 4096), (512, 4096, 4096), (128, 11008, 4096), (256, 8192, 4096), (64, 4096, 4096), (256, 6144, 4096)):
        time_one(M, N, K, 4, f16, None)
        for sk in (1, 2, 4):
            for rt in (8, 4):
                try:
                    dev.get_plan(M, N, K, 4, 64, tid_of(4, 32), num_sms, f16, dev.Overrides(family=6, splitk=sk, m_tiles=rt))
                except RuntimeError:
                    continue
                time_one(M, N, K, 4, f16, dict(family=6, splitk=sk, m_tiles=rt))


def check_kp4():
    """Four K parts per workgroup (override kw = 4: 64 x 64 tiles, round 6) on a subset of check()'s cases, with and without K slices."""
    nfail = 0
    for (bits, tile_p, g, dtype, K, N) in [(4, 32, 64, f16, 4096, 4096), (4, 64, 64, bf16, 2048, 1024), (4, 32, 128, f16, 3072, 512),
                                           (2, 32, 64, f16, 4096, 2048), (2, 64, 128, bf16, 2048, 1024), (4, 32, 32, f16, 1024, 256),
                                           (4, 32, 64, f16, 4096, 11008), (4, 64, 256, bf16, 4096, 256)]:
        torch.manual_seed(K + N)
        W = torch.randint(0, 2 ** bits, (K, N), dtype=torch.uint8, device=d)
        S = torch.randn(N, K // g, device=d).to(dtype)
        table = torch.randn(2 ** bits, device=d).to(dtype)
        table2 = utils.make_qmap2_from_qmap(table)
        tid = tid_of(bits, tile_p)
        Q = utils.pack(W, bits, [tid], num_sms)
        What = table[W.long()] * torch.repeat_interleave(S, g, dim=1).T
        tol = 1e-3 if dtype == f16 else 4e-3
        for M in (1, 60, 64, 130, 256, 700):
            X = (torch.randn(M, K, device=d) / 100).to(dtype)
            ref = X.float() @ What.float()
            ks = torch.randint(0, K, (M,), device=d)
            E = torch.zeros(M, K, device=d, dtype=dtype)
            E[torch.arange(M, device=d), ks] = 1
            for sk in (1, 2, 3, 4, 8):
                rec = {"kind": "check_kp4", "bits": bits, "tile_p": tile_p, "g": g, "dtype": str(dtype)[6:], "K": K, "N": N, "M": M, "splitk": sk}
                try:
                    ovr = dev.Overrides(family=6, splitk=sk, kw=4)
                    try:
                        pl = dev.get_plan(M, N, K, bits, g, tid, num_sms, dtype, ovr)
                    except RuntimeError:
                        continue
                    assert pl["m_tiles"] == 4 and pl["kw"] == 4 and pl["grid"] == -(-M // 64) * (N // 64) * sk, pl
                    o = dev.qgemm_planned(X, Q, S, table, table2, ws, bits, g, tid, num_sms, ovr)
                    o1 = dev.qgemm_planned(E, Q, S, table, table2, ws, bits, g, tid, num_sms, ovr)
                    o2 = dev.qgemm_planned(X, Q, S, table, table2, ws, bits, g, tid, num_sms, ovr)
                    torch.cuda.synchronize()
                    err = ((o.float() - ref).norm() / ref.norm()).item()
                    rec.update(err=err, onehot_exact=bool(torch.equal(o1, What[ks])), repeat_identical=bool(torch.equal(o, o2)), state_clean=state_clean())
                    rec["ok"] = bool(err < tol and rec["onehot_exact"] and rec["repeat_identical"] and rec["state_clean"])
                    if not rec["ok"]:
                        bad = ((o.float() - ref).abs() > 0.05 * ref.abs().max()) | o.float().isnan()
                        rec["nbad"] = int(bad.sum().item())
                        rec["bad_rows"] = bad.any(1).nonzero().flatten()[:12].tolist()
                        rec["bad_cols"] = bad.any(0).nonzero().flatten()[:16].tolist()
                    if not rec["state_clean"]:
                        ws[:65536].zero_()
                except Exception as ex:  # noqa: BLE001
                    rec.update(ok=False, error=str(ex)[:300])
                if not rec["ok"]:
                    nfail += 1
                    emit(rec)
        del W, S, Q, What
        torch.cuda.empty_cache()
    emit({"kind": "check_kp4_summary", "failed": nfail})
    return nfail


def time_kp4():
    for (M, N, K) in ((256, 4096, 4096), (192, 4096, 4096), (128, 4096, 4096), (64, 4096, 4096), (256, 2048, 8192), (128, 8192, 4096), (512, 2048, 4096),
                      (64, 8192, 8192), (256, 11008, 4096)):
        time_one(M, N, K, 4, f16, None)
        for shp in (dict(family=6, splitk=1, kw=4), dict(family=6, splitk=2, kw=4), dict(family=6, splitk=2, kw=2, m_tiles=4), dict(family=6, splitk=1, kw=2, m_tiles=4),
                    dict(family=6, splitk=1, kw=2, m_tiles=8)):
            try:
                dev.get_plan(M, N, K, 4, 64, tid_of(4, 32), num_sms, f16, dev.Overrides(**shp))
            except RuntimeError:
                continue
            time_one(M, N, K, 4, f16, shp)
    time_one(256, 4096, 4096, 4, bf16, dict(family=6, splitk=1, kw=4))
    time_one(256, 4096, 4096, 4, bf16, None)
    time_one(256, 4096, 4096, 2, f16, dict(family=6, splitk=1, kw=4))
    time_one(256, 4096, 4096, 2, f16, None)


def main():
    rc = 0
    if "check_kp4" in what:
        rc |= check_kp4()
    if "time_kp4" in what:
        time_kp4()
    if "check_rt4" in what:
        rc |= check_rt4()
    if "time_rt4" in what:
        time_rt4()
    if "time_had" in what:
        time_had()
    if "check_ldw" in what:
        rc |= check_ldw()
    if "time_ldw" in what:
        time_ldw()
    if "check_skinny" in what:
        rc |= check_skinny()
    if "time_skinny" in what:
        time_skinny()
    if "time_abl" in what:                                             # ablation builds (tools/splitk_ablate.sh): loop rates only
        tag = os.path.basename(os.environ.get("FLUTE_AMD_LIB", "shipped"))
        for sk in (1, 4):
            time_one(256, 4096, 4096, 4, f16, dict(family=6, splitk=sk), tag=tag)
        time_one(256, 11008, 4096, 4, f16, dict(family=6, splitk=1), tag=tag)
        for sk in (2, 8):                                              # the seam's price: ablation 128 against the shipped library
            time_one(256, 4096, 4096, 4, f16, dict(family=6, splitk=sk), tag=tag)
        for (M, N, K, sk) in ((16, 4096, 4096, 4), (16, 4096, 4096, 2), (16, 8192, 4096, 2), (4, 4096, 4096, 4), (16, 2048, 8192, 8)):
            time_one(M, N, K, 4, f16, dict(family=5, splitk=sk), tag=tag, steps=500)
    if "check" in what:
        rc |= check()
    if "stress" in what:
        rc |= stress()
    if "time" in what:
        timing()
    if "time_more" in what:
        timing(True)
    sys.exit(1 if rc else 0)


if __name__ == "__main__":
    main()
