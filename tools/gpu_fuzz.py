"""Randomised parity sweep on the GPU (development aid): random layouts, group sizes, shapes, batch
sizes and TEMPLATE IDS (every id is a different launch plan) against a torch fp32 evaluation of the
reference's formula from the integer codes.  Non-stopping; exits 1 on any failure.

    python tools/gpu_fuzz.py [cases] [seed]
"""
import json
import os
import random
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import flute_amd  # noqa: E402
from flute_amd import dev as fdev, utils  # noqa: E402

dev = torch.device("cuda:0")
num_sms = utils.get_device_num_sms(dev)
ws = utils.get_workspace_streamk(dev)
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
fails = []
ran, maxerr, fam = 0, {}, {}
# FUZZ_OVR='{"one_shot": 2}': every call carries these launch-plan overrides (flute_overrides fields)
ovr = fdev.Overrides(**json.loads(os.environ["FUZZ_OVR"])) if os.environ.get("FUZZ_OVR") else None


def run(X, Q, S, table, table2, bits, g, tid):
    if ovr is None:
        return flute_amd.qgemm(X, Q, S, table, table2, ws, bits, g, tid, num_sms)
    return fdev.qgemm_planned(X, Q, S, table, table2, ws, bits, g, tid, num_sms, ovr)


t0 = time.time()
for case in range(ncases):
    bits = rng.choice([4, 4, 4, 2, 3])
    tids = [t for (b, t) in sorted(flute_amd.TEMPLATE_CONFIGS) if b == bits]
    tid = rng.choice(tids)
    tile_p = flute_amd.TEMPLATE_CONFIGS[(bits, tid)]["TileP"]
    if bits == 3 and tile_p != 32:
        continue
    g = rng.choice([32, 64, 64, 128, 256])
    dtype = rng.choice([torch.float16, torch.bfloat16])
    blk = tile_p * (16 if bits == 3 else 16 // bits)
    N = blk * rng.choice([1, 2, 3, 4, 7, 8, 16, 16, 32, 43, 64])
    K = g * rng.randint(1, max(1, 8192 // g))
    K = (K + 63) // 64 * 64
    if K % g:
        K = (K // g + 1) * g
        if K % 64:
            K = K * 64 // __import__("math").gcd(K, 64)
    M = rng.choice([1, 1, 2, 3, 4, 5, 7, 8, 13, 16, 17, 31, 32, 33, 48, 64, 65, 100, 128, 200, 256, 300, 512, 700, 1024, 2048, 2100])
    if os.environ.get("FUZZ_DECODE") == "1":                    # decode kernels only (one-shot and ring)
        M = rng.choice([1, 1, 1, 2, 2, 3, 4])
    if os.environ.get("FUZZ_DECODE") == "2":                    # one row, K in whole 1024-k steps (persistent one-shot kernel)
        M = 1
        K = max(1024, K // 1024 * 1024)
    if os.environ.get("FUZZ_DECODE") == "3":                    # round 5: the lean kernels' shapes (4 bits, K = 2048 / 3584 / 4096 / 8192, M <= 16; automatic ids often)
        if bits != 4:
            continue
        M = rng.choice([1, 1, 2, 3, 4, 5, 8, 9, 16])
        K = rng.choice([2048, 3584, 4096, 4096, 8192])
        g = rng.choice([64, 64, 128, 256])
        if rng.random() < 0.6:
            tid = rng.choice([t for t in tids if t % 4 == 0 and t < 32] or tids)
            tile_p = flute_amd.TEMPLATE_CONFIGS[(bits, tid)]["TileP"]
            blk = tile_p * (16 // bits)
        N = blk * rng.choice([4, 8, 16, 16, 32, 32, 43, 64])
    torch.manual_seed(case)
    W = torch.randint(0, 2 ** bits, (K, N), dtype=torch.uint8, device=dev)
    S = torch.randn(N, K // g, device=dev).to(dtype)
    table = torch.randn(2 ** bits, device=dev).to(dtype)
    table2 = utils.make_qmap2_from_qmap(table)
    Q = utils.pack(W, bits, [tid], num_sms)
    What = table[W.long()].float() * torch.repeat_interleave(S.float(), g, dim=1).T
    X = (torch.randn(M, K, device=dev) / 100).to(dtype)
    ref = X.float() @ What
    tag = f"case {case}: b{bits} tid{tid} tp{tile_p} g{g} {str(dtype)[6:]} K{K} N{N} M{M}"
    if ovr is not None:                                            # a forced plan that does not exist for this draw: skipped, not failed
        try:
            fdev.get_plan(M, N, K, bits, g, tid, num_sms, dtype, ovr)
        except RuntimeError:
            continue
    try:
        out = run(X, Q, S, table, table2, bits, g, tid)
        torch.cuda.synchronize()
        err = ((out.float() - ref).norm() / ref.norm()).item()
        ran += 1
        maxerr[str(dtype)] = max(maxerr.get(str(dtype), 0.0), err)
        pl = fdev.get_plan(M, N, K, bits, g, tid, num_sms, dtype, ovr)
        key = (pl['family'], pl['m_block'], pl['m_tiles'], pl.get('slabs_per_wave'), pl.get('one_shot'))
        fam[key] = fam.get(key, 0) + 1
        tol = 1e-3 if dtype == torch.float16 else 4e-3
        if not err < tol:
            fails.append((tag, err))
            print("FAIL", tag, f"err={err:.3e}", utils.get_plan(M, N, K, bits, g, tid, num_sms, dtype), flush=True)
        # one-hot rows are exact
        ks = torch.randint(0, K, (M,), device=dev)
        E = torch.zeros(M, K, device=dev, dtype=dtype)
        E[torch.arange(M, device=dev), ks] = 1
        o1 = run(E, Q, S, table, table2, bits, g, tid)
        if not torch.equal(o1.float(), What[ks].to(dtype).float()):
            fails.append((tag, "one-hot"))
            print("FAIL one-hot", tag, utils.get_plan(M, N, K, bits, g, tid, num_sms, dtype), flush=True)
    except Exception as ex:  # noqa: BLE001
        fails.append((tag, str(ex)[:200]))
        print("EXC ", tag, str(ex)[:200], flush=True)
print(f"fuzz: {ran - len(fails)}/{ran} executed cases ok ({ncases - ran} skipped draws) in {time.time() - t0:.1f}s; max rel err {maxerr}")
print("plans exercised (family, R or rows/pass, MT, SW, one_shot):", sorted(fam.items()))
os.makedirs("gpurun_out", exist_ok=True)
json.dump({"cases": ncases, "executed": ran, "fails": fails, "max_rel_err": maxerr,
           "plans": {str(k): v for k, v in sorted(fam.items())}}, open(os.environ.get("FUZZ_OUT", "gpurun_out/fuzz.json"), "w"), indent=1, default=str)
sys.exit(1 if fails else 0)
