"""Wide, non-stopping correctness sweep on the GPU (development aid; the formal
parity tests are tests/test_qgemm_gpu.py).  Checker = torch on the GPU evaluating
the reference's formula from the integer codes."""
import itertools
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import flute_amd  # noqa: E402
from flute_amd import utils, _lib  # noqa: E402

dev = torch.device("cuda:0")
num_sms = utils.get_device_num_sms(dev)
ws = utils.get_workspace_streamk(dev)
lib = _lib.get()
from flute_amd import dev as dev_mod  # noqa: E402
fails, total = [], 0
# SELFCHECK_FAMILY=2: run every automatic case (also M <= 4) through the MFMA kernel
FORCE_FAMILY = int(os.environ.get("SELFCHECK_FAMILY", "0"))


def tids(bits, tile_p):
    return [t for (b, t), c in sorted(flute_amd.TEMPLATE_CONFIGS.items()) if b == bits and c["TileP"] == tile_p]


def case(bits, tile_p, g, dtype, K, N, Ms, ovr_list, seed=0):
    global total
    torch.manual_seed(seed)
    W = torch.randint(0, 2 ** bits, (K, N), dtype=torch.uint8, device=dev)
    S = torch.randn(N, K // g, device=dev).to(dtype)
    table = torch.randn(2 ** bits, device=dev).to(dtype)
    table2 = utils.make_qmap2_from_qmap(table)
    tid = tids(bits, tile_p)[0]
    Q = utils.pack(W, bits, [tid], num_sms)
    What = table[W.long()] * torch.repeat_interleave(S, g, dim=1).T
    tol = 1e-3 if dtype == torch.float16 else 8e-3
    for M in Ms:
        X = (torch.randn(M, K, device=dev) / 100).to(dtype)
        ref = X.float() @ What.float()
        for ovr in ovr_list:
            if FORCE_FAMILY and ovr[0] == -1:
                ovr = (FORCE_FAMILY,) + tuple(ovr[1:])
            total += 1
            tag = f"b{bits} tp{tile_p} g{g} {str(dtype)[6:]} K{K} N{N} M{M} ovr{ovr}"
            try:
                out = dev_mod.qgemm_planned(X, Q, S, table, table2, ws, bits, g, tid, num_sms,
                                            dev_mod.overrides_from_tuple(ovr))
                torch.cuda.synchronize()
                err = ((out.float() - ref).norm() / ref.norm()).item()
                ok = err < tol
                if not ok:
                    bad = ((out.float() - ref).abs() > 0.05 * ref.abs().max()).nonzero()
                    fails.append((tag, err, bad[:6].tolist(), int(bad.shape[0])))
                    print("FAIL", tag, f"err={err:.3e} nbad={bad.shape[0]} first={bad[:6].tolist()}", flush=True)
            except Exception as ex:  # noqa: BLE001
                fails.append((tag, str(ex)[:200]))
                print("EXC ", tag, str(ex)[:200], flush=True)
    # one-hot exactness through both families
    ks = torch.randint(0, K, (16,), device=dev)
    X = torch.zeros(16, K, device=dev, dtype=dtype)
    X[torch.arange(16), ks] = 1
    for M in (1, 16):
        total += 1
        out = flute_amd.qgemm(X[:M], Q, S, table, table2, ws, bits, g, tid, num_sms)
        if not torch.equal(out, What[ks[:M]]):
            nb = (out != What[ks[:M]]).sum().item()
            fails.append((f"one-hot b{bits} tp{tile_p} g{g} K{K} N{N} M{M}", nb))
            print("FAIL one-hot", bits, tile_p, g, dtype, K, N, M, "mismatches", nb, flush=True)
    total += 1
    if not torch.equal(utils.unpack_codes(Q, bits, tid), W):
        fails.append((f"unpack b{bits} tp{tile_p}",))
        print("FAIL unpack", bits, tile_p, flush=True)


t0 = time.time()
AUTO = (-1, -1, -1, -1, -1, -1, -1)
for bits, tile_p in [(4, 32), (4, 64), (2, 32), (2, 64), (3, 32)]:
    for dtype in (torch.float16, torch.bfloat16):
        blk = tile_p * (16 if bits == 3 else 16 // bits)
        case(bits, tile_p, 64, dtype, 1024, 2 * blk, [1, 2, 3, 4, 5, 8, 9, 16, 17, 33, 64, 130], [AUTO, (-1, -1, -1, -1, -1, -1, 1)])
    case(bits, tile_p, 128, torch.float16, 4608, blk, [1, 4, 8, 16, 48, 70],
         [AUTO, (-1, -1, -1, 1, 1, 32, 0), (-1, -1, -1, 2, 2, 8, 1), (-1, -1, 4, 4, 1, 1, 0), (-1, -1, 16, 8, 1, 1, 1),
          (2, 1, 2, 2, 2, -1, -1), (2, 1, 1, 1, 1, 1, -1), (2, 2, 2, 1, 4, 2, -1),
          (2, 1, -1, -1, -1, -1, -1), (2, 2, 8, 4, 2, -1, -1), (2, 4, 4, 1, 1, -1, -1), (2, 4, 8, 8, 1, -1, -1),
          (2, 1, -1, -1, -1, 4, -1), (2, 2, 8, 2, 1, 2, -1), (2, 1, 4, 4, 2, 2, -1), (2, 2, -1, -1, -1, 4, -1),
          (2, 1, 8, 4, 1, 2, 2), (2, 1, 8, 8, 1, 4, 2), (2, 1, 4, 1, 2, 2, 2), (2, 1, 8, 2, 1, 4, 2)])
    case(bits, tile_p, 32, torch.float16, 256, blk, [1, 7, 20], [AUTO])
    case(bits, tile_p, 256, torch.bfloat16, 2048, blk, [1, 7, 20], [AUTO])
case(4, 32, 64, torch.float16, 4096, 4096, [1, 2, 4, 8, 16, 256], [AUTO])
case(4, 32, 64, torch.float16, 4096, 4096, [64, 256], [(2, 1, 8, 8, 1, 2, 2), (2, 1, 8, 8, 1, 4, 2)])
case(4, 64, 64, torch.bfloat16, 2048, 1024, [33, 100], [(2, 1, 8, 4, 1, 2, 2), (2, 1, 8, 8, 1, 2, 2)])
case(4, 64, 64, torch.float16, 4096, 11008, [1, 16, 256], [AUTO])
case(3, 32, 64, torch.bfloat16, 8192, 8192, [1, 4, 16, 64], [AUTO])
case(2, 32, 64, torch.float16, 14336, 4096, [1, 3, 32], [AUTO])
print(f"selfcheck: {total - len(fails)}/{total} ok in {time.time() - t0:.1f}s")
os.makedirs("gpurun_out", exist_ok=True)
json.dump({"total": total, "fails": fails}, open("gpurun_out/selfcheck.json", "w"), indent=1, default=str)
sys.exit(1 if fails else 0)
