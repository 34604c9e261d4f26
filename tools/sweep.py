"""Timing sweep over launch-plan overrides (development aid for the tuner).
Writes gpurun_out/sweep.json.  HBM-cold: rotating weight copies > 256 MiB."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from flute_amd import _lib, utils  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.get()
rows = []

from flute_amd.dev import overrides_from_tuple as ovr7  # noqa: E402


def run(M, N, K, bits, g, dtype, tid, ovr, steps=300, hot=False):
    lay = bench.Layer(M, N, K, bits, g, dtype, dev, 1 if hot else bench.copies_for(N, K, bits))
    lay.template_id = tid
    lay.ovr = ovr7(ovr)
    try:
        from flute_amd import dev as _dev
        plan = _dev.get_plan(M, N, K, bits, g, tid, lay.num_sms, dtype, lay.ovr)
        ms, _ = bench.time_graph(lay, steps, 10, torch.cuda.synchronize)
        us = ms / steps * 1e3
        r = {"M": M, "N": N, "K": K, "bits": bits, "g": g, "dtype": str(dtype), "tid": tid,
             "ovr": ovr, "hot": hot, "us": round(us, 3), "GBps": round(lay.bytes() / us / 1e3, 1),
             "TFLOPs": round(lay.flops() / us / 1e6, 2), "plan": plan}
    except Exception as ex:  # noqa: BLE001
        r = {"M": M, "N": N, "K": K, "bits": bits, "ovr": ovr, "error": str(ex)[:200]}
    rows.append(r)
    print(json.dumps(r), flush=True)
    del lay
    torch.cuda.empty_cache()


f16 = torch.float16
bf16 = torch.bfloat16
which = sys.argv[1] if len(sys.argv) > 1 else "decode"
if which == "decode":
    for waves in (8, 16):
        for kw in (1, 2, 4, 8):
            for pre in (0, 1):
                run(1, 4096, 4096, 4, 64, f16, 16, (0, -1, waves, kw, 1, -1, pre))
    run(1, 4096, 4096, 4, 64, f16, 16, (0, -1, 16, 4, 1, -1, 0), hot=True)
    run(1, 4096, 4096, 4, 64, bf16, 16, (0, -1, 16, 4, 1, -1, 0))
    for (n, k) in ((11008, 4096), (4096, 14336), (28672, 8192)):
        for waves, kw in ((16, 1), (16, 2), (8, 1), (8, 2), (16, 4)):
            run(1, n, k, 4, 64, f16, 16, (0, -1, waves, kw, 1, -1, 0))
    run(1, 28672, 8192, 4, 64, f16, 16, (0, -1, 16, 1, 1, -1, 1))
    for M in (2, 3, 4):
        run(M, 4096, 4096, 4, 64, f16, 16, (0, -1, -1, -1, -1, -1, 0))
        run(M, 28672, 8192, 4, 64, f16, 16, (0, -1, -1, -1, -1, -1, 0))
    for waves, kw in ((8, 1), (8, 2), (8, 4)):
        run(1, 8192, 8192, 3, 64, bf16, 4, (0, -1, waves, kw, 1, -1, 0))
    run(1, 28672, 8192, 3, 64, bf16, 4, (0, -1, 8, 1, 1, -1, 0))
    for waves, kw in ((16, 4), (16, 2), (8, 2)):
        run(1, 4096, 4096, 2, 64, f16, 4, (0, -1, waves, kw, 1, -1, 0))
    run(1, 28672, 8192, 2, 64, f16, 4, (0, -1, 16, 1, 1, -1, 0))
elif which == "kscale":
    # fixed launch overhead vs per-K slope of the column-per-lane MFMA kernel
    for kw in (8, 4, 2):
        for K in (512, 1024, 2048, 4096, 8192, 16384):
            if K // kw < 64:
                continue
            run(256, 4096, K, 4, 64, f16, 16, (2, 1, 8, kw, 1, 4, -1), steps=100, hot=True)
    for K in (1024, 4096, 16384):
        run(256, 4096, K, 4, 64, f16, 16, (2, 1, 8, 4, 1, 4, -1), steps=100)
        run(64, 4096, K, 4, 64, f16, 16, (2, 1, 8, 8, 1, 4, -1), steps=100, hot=True)
        run(16, 4096, K, 4, 64, f16, 16, (2, 4, 8, 8, 1, 1, -1), steps=100, hot=True)
elif which == "stride":
    # power-of-two row stride of A (L2 channel camping?) vs odd multiples of 64
    for K in (3968, 4032, 4096, 4160, 4224, 8192, 8256):
        run(256, 4096, K, 4, 64, f16, 16, (2, 1, 8, 8, 1, 4, -1), steps=100, hot=True)
        run(64, 4096, K, 4, 64, f16, 16, (2, 1, 8, 8, 1, 4, -1), steps=100, hot=True)
elif which == "tile":
    # LDS-DMA staged MFMA kernel (family 3) against the r01 column-per-lane kernel (family 2)
    for fam in ((int(sys.argv[2]),) if len(sys.argv) > 2 else (2, 3)):
        run(16, 4096, 4096, 4, 64, f16, 16, (fam, 4, 8, 8, 1, 1, -1))
        run(16, 4096, 4096, 4, 64, f16, 16, (fam, 2, 8, 8, 1, 1, -1))
        run(16, 11008, 4096, 4, 64, f16, 16, (fam, -1, -1, -1, -1, -1, -1))
        run(16, 28672, 8192, 4, 64, f16, 16, (fam, 1, 8, 4, 1, 1, -1))
        run(64, 4096, 4096, 4, 64, f16, 16, (fam, 1, 8, 8, 1, 1, -1))
        run(64, 4096, 4096, 4, 64, f16, 16, (fam, 1, 8, 8, 1, 4, -1))
        run(256, 4096, 4096, 4, 64, f16, 16, (fam, 1, 8, 8, 1, 4, -1))
        run(256, 4096, 4096, 4, 64, f16, 16, (fam, 1, 8, 4, 1, 4, -1))
        run(256, 4096, 4096, 4, 64, f16, 16, (fam, 1, 8, 8, 1, 2, -1))
        run(256, 4096, 4096, 4, 64, bf16, 16, (fam, 1, 8, 8, 1, 4, -1))
        run(256, 11008, 4096, 4, 64, f16, 16, (fam, 1, 8, 2, 1, 4, -1))
        run(256, 11008, 4096, 4, 64, f16, 16, (fam, 1, 8, 4, 1, 4, -1))
        run(1024, 4096, 4096, 4, 64, f16, 16, (fam, 1, 8, 4, 1, 4, -1))
        run(1024, 4096, 4096, 4, 64, f16, 16, (fam, 1, 8, 2, 1, 4, -1))
        run(4096, 4096, 4096, 4, 64, f16, 16, (fam, 1, 8, 1, 1, 4, -1), steps=50)
        run(64, 8192, 8192, 3, 64, bf16, 4, (fam, -1, -1, -1, -1, -1, -1))
        run(256, 4096, 4096, 2, 64, f16, 4, (fam, -1, -1, -1, -1, -1, -1))
elif which == "w3":
    for (n, k) in ((8192, 8192), (28672, 8192)):
        for waves, kw in ((16, 4), (16, 8), (16, 16), (8, 2), (8, 4), (8, 8), (4, 4), (4, 2)):
            run(1, n, k, 3, 64, bf16, 4, (0, -1, waves, kw, 1, -1, 0))
        run(1, n, k, 3, 64, bf16, 4, (0, -1, -1, -1, -1, -1, 0))
    run(2, 8192, 8192, 3, 64, bf16, 4, (0, -1, -1, -1, -1, -1, 0))
    run(1, 28672, 8192, 4, 64, f16, 16, (0, -1, 16, 1, 1, -1, 0))
    run(1, 28672, 8192, 4, 64, f16, 16, (0, -1, 16, 2, 1, -1, 0))
    run(1, 28672, 8192, 4, 64, f16, 16, (0, -1, 16, 4, 1, -1, 0))
elif which == "big":
    run(1, 28672, 8192, 4, 64, f16, 16, (0, -1, 16, 1, 1, -1, 0))
    run(1, 28672, 8192, 4, 64, f16, 16, (0, -1, 16, 2, 1, -1, 0))
    run(1, 28672, 8192, 4, 64, f16, 16, (0, -1, 8, 1, 1, -1, 0))
    run(1, 11008, 4096, 4, 64, f16, 16, (0, -1, 16, 2, 1, -1, 0))
    run(1, 4096, 4096, 4, 64, f16, 16, (0, -1, 8, 2, 1, -1, 0))
    run(1, 28672, 8192, 3, 64, bf16, 4, (0, -1, 8, 2, 1, -1, 0))
    run(1, 28672, 8192, 3, 64, bf16, 4, (0, -1, 4, 4, 1, -1, 0))
    run(1, 8192, 8192, 3, 64, bf16, 4, (0, -1, 8, 8, 1, -1, 0))
    run(4, 28672, 8192, 4, 64, f16, 16, (0, -1, -1, -1, -1, -1, 0))
elif which == "sw":
    # two slabs per wave against one (prescale slot of the overrides: 1 / 2)
    for swv in (1, 2):
        run(256, 4096, 4096, 4, 64, f16, 16, (2, 1, 8, 8, 1, 4, swv))
        run(256, 4096, 4096, 4, 64, f16, 16, (2, 1, 8, 8, 1, 2, swv))
        run(256, 11008, 4096, 4, 64, f16, 16, (2, 1, 8, 2, 1, 4, swv))
        run(256, 11008, 4096, 4, 64, f16, 16, (2, 1, 8, 4, 1, 4, swv))
        run(256, 11008, 4096, 4, 64, f16, 16, (2, 1, 8, 8, 1, 4, swv))
        run(256, 11008, 4096, 4, 64, f16, 16, (2, 1, 8, 8, 1, 2, swv))
        run(1024, 4096, 4096, 4, 64, f16, 16, (2, 1, 8, 2, 1, 4, swv))
        run(1024, 4096, 4096, 4, 64, f16, 16, (2, 1, 8, 4, 1, 4, swv))
        run(1024, 4096, 4096, 4, 64, f16, 16, (2, 1, 8, 8, 1, 4, swv))
        run(4096, 4096, 4096, 4, 64, f16, 16, (2, 1, 8, 1, 1, 4, swv), steps=50)
        run(4096, 4096, 4096, 4, 64, f16, 16, (2, 1, 8, 2, 1, 4, swv), steps=50)
        run(256, 4096, 4096, 4, 64, bf16, 16, (2, 1, 8, 8, 1, 2, swv))
elif which == "n11008":
    for waves, kw in ((16, 1), (16, 2), (16, 4), (16, 8), (8, 1), (8, 2), (8, 4), (8, 8), (4, 1), (4, 2), (4, 4)):
        run(1, 11008, 4096, 4, 64, f16, 16, (0, -1, waves, kw, 1, -1, 0))
    for waves, kw in ((16, 2), (16, 4), (8, 2), (8, 4)):
        run(1, 14336, 4096, 4, 64, f16, 16, (0, -1, waves, kw, 1, -1, 0))
        run(1, 4096, 14336, 4, 64, f16, 16, (0, -1, waves, kw, 1, -1, 0))
elif which == "calib":
    import time
    lib = _lib.get()
    sink = torch.zeros(16, dtype=torch.int32, device=dev)
    for total_mb, bpw in ((8, 8192), (8, 16384), (8, 32768), (22, 8192), (117, 16384), (117, 65536), (512, 65536)):
        nbytes = total_mb * (1 << 20) // bpw * bpw
        copies = max(1, (300 << 20) // nbytes + 1)
        bufs = [torch.randint(-2 ** 31, 2 ** 31 - 1, (nbytes // 4,), dtype=torch.int32, device=dev) for _ in range(copies)]
        for grid, block in ((256, 1024), (512, 512), (1024, 256), (2048, 256), (256, 512)):
            def step(i):
                rc = lib.flute_debug_stream_read(bufs[i % copies].data_ptr(), sink.data_ptr(), nbytes, bpw, grid,
                                                 block, torch.cuda.current_stream().cuda_stream)
                assert rc == 0, rc
            for i in range(5):
                step(i)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            steps = 200
            with torch.cuda.graph(g):
                for i in range(steps):
                    step(i)
            g.replay(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / steps * 1e3
            r = {"calib_read_MB": total_mb, "bytes_per_wave": bpw, "grid": grid, "block": block,
                 "us": round(us, 3), "GBps": round(nbytes / us / 1e3, 1)}
            rows.append(r); print(json.dumps(r), flush=True)
        del bufs
        torch.cuda.empty_cache()
elif which == "quick":
    run(1, 28672, 8192, 4, 64, f16, 16, (0, -1, 16, 1, 1, -1, 0), steps=100)
    run(1, 11008, 4096, 4, 64, f16, 16, (0, -1, 16, 1, 1, -1, 0))
    run(1, 4096, 4096, 4, 64, f16, 16, (0, -1, 8, 2, 1, -1, 0))
    run(16, 4096, 4096, 4, 64, f16, 16, (2, 4, 8, 8, 1, 1, -1))
    run(16, 28672, 8192, 4, 64, f16, 16, (2, 1, 8, 4, 1, 1, -1), steps=100)
    run(64, 4096, 4096, 4, 64, f16, 16, (2, 1, 8, 8, 1, 1, -1))
    run(256, 4096, 4096, 4, 64, f16, 16, (2, 1, 8, 8, 1, 4, -1), steps=200)
    run(256, 4096, 4096, 4, 64, f16, 16, (2, 1, 8, 4, 1, 2, -1), steps=200)
    run(256, 11008, 4096, 4, 64, f16, 16, (2, 1, 8, 2, 1, 4, -1), steps=100)
    run(1024, 4096, 4096, 4, 64, f16, 16, (2, 1, 8, 4, 1, 4, -1), steps=100)
elif which == "w4":         # fewer, fatter waves: less launch skew, less latency hiding
    for waves, kw in ((8, 2), (4, 1), (4, 2), (4, 4), (2, 1), (2, 2), (8, 2)):
        run(1, 4096, 4096, 4, 64, f16, 16, (0, -1, waves, kw, 1, -1, 0))
    for waves, kw in ((16, 1), (4, 1), (4, 2), (4, 4), (8, 1)):
        run(1, 11008, 4096, 4, 64, f16, 16, (0, -1, waves, kw, 1, -1, 0))
elif which == "batch":      # lane-sharing (R > 1) tile kernels: batched macro-step vs -DFLUTE_TILE_NO_BATCH
    for R in (4, 2):
        run(16, 4096, 4096, 4, 64, f16, 16, (2, R, 8, 8, 1, 1, -1))
        run(16, 4096, 4096, 4, 64, f16, 16, (2, R, 8, 4, 1, 1, -1))
        run(16, 11008, 4096, 4, 64, f16, 16, (2, R, 8, 8, 1, 1, -1))
        run(16, 28672, 8192, 4, 64, f16, 16, (2, R, 8, 8, 1, 1, -1), steps=100)
        run(32, 4096, 4096, 4, 64, f16, 16, (2, R, 8, 8, 1, 2, -1))
        run(64, 4096, 4096, 4, 64, f16, 16, (2, R, 8, 8, 1, 4, -1))
    run(8, 4096, 4096, 4, 64, f16, 16, (2, 4, 8, 8, 1, 1, -1))
    for R, MT in ((4, 1), (2, 1), (2, 2)):
        run(16 * MT, 4096, 4096, 4, 64, bf16, 16, (2, R, 8, 8, 1, MT, -1))
    run(16, 11008, 4096, 4, 64, bf16, 16, (2, 2, 8, 8, 1, 1, -1))
    run(16, 8192, 8192, 4, 128, bf16, 16, (2, 4, 8, 8, 1, 1, -1))
    run(16, 4096, 4096, 2, 64, f16, 4, (2, 4, -1, -1, -1, -1, -1))
    run(16, 8192, 8192, 4, 128, f16, 16, (2, 4, 8, 8, 1, 1, -1))
    run(16, 4096, 4096, 4, 64, f16, 16, (-1, -1, -1, -1, -1, -1, -1))
    run(16, 11008, 4096, 4, 64, f16, 16, (-1, -1, -1, -1, -1, -1, -1))
elif which == "ring":
    run(1, 28672, 8192, 4, 64, f16, 16, (0, -1, 16, 1, 1, -1, 0), steps=100)
    run(1, 28672, 8192, 4, 64, f16, 16, (0, -1, 8, 1, 1, -1, 0), steps=100)
    run(1, 11008, 4096, 4, 64, f16, 16, (0, -1, 16, 1, 1, -1, 0))
    run(1, 4096, 4096, 4, 64, f16, 16, (0, -1, 8, 2, 1, -1, 0))
    run(1, 4096, 4096, 4, 64, f16, 16, (0, -1, 16, 4, 1, -1, 0))
    run(1, 4096, 4096, 4, 64, f16, 16, (0, -1, 16, 2, 1, -1, 0))
elif which == "ablate":
    for pre in (0, 101, 102, 103):
        run(1, 28672, 8192, 4, 64, f16, 16, (0, -1, 16, 1, 1, -1, pre), steps=100)
        run(1, 4096, 4096, 4, 64, f16, 16, (0, -1, 8, 2, 1, -1, pre))
        run(1, 4096, 4096, 4, 64, f16, 16, (0, -1, 8, 2, 1, -1, pre), hot=True)
elif which == "mt":
    for M in (16, 32, 64, 256):
        for R, MT in ((1, 1), (2, 1), (4, 1), (1, 2), (2, 2), (1, 4), (2, 4)):
            if MT * 16 > max(M, 16):
                continue
            for kw in (8, 4, 2):
                run(M, 4096, 4096, 4, 64, f16, 16, (2, R, 8, kw, 1, MT, -1), steps=200)
    for M in (16, 256):
        for R, MT, kw in ((1, 1, 8), (1, 4, 8), (1, 4, 4), (2, 4, 4), (1, 4, 2), (1, 2, 4)):
            if MT * 16 > max(M, 16):
                continue
            run(M, 11008, 4096, 4, 64, f16, 16, (2, R, 8, kw, 1, MT, -1), steps=100)
    run(256, 4096, 4096, 4, 64, bf16, 16, (2, 1, 8, 8, 1, 4, -1), steps=200)
    run(1024, 4096, 4096, 4, 64, f16, 16, (2, 1, 8, 4, 1, 4, -1), steps=100)
elif which == "smallM":
    for M in (1, 4, 8, 16):
        for R in (1, 2, 4):
            for nw, kw in ((8, 8), (8, 4), (4, 4)):
                run(M, 4096, 4096, 4, 64, f16, 16, (2, R, nw, kw, 1, -1, -1))
    run(16, 4096, 4096, 4, 64, f16, 16, (2, 1, 8, 8, 4, -1, -1))
    run(16, 4096, 4096, 4, 64, bf16, 16, (2, 4, 8, 8, 1, -1, -1))
    for (n, k) in ((11008, 4096), (28672, 8192)):
        for R in (1, 2, 4):
            run(16, n, k, 4, 64, f16, 16, (2, R, 8, 8, 1, -1, -1))
        run(16, n, k, 4, 64, f16, 16, (2, 1, 8, 4, 1, -1, -1))
    run(16, 8192, 8192, 3, 64, bf16, 4, (2, 1, -1, -1, -1, -1, -1))
    run(16, 4096, 4096, 2, 64, f16, 4, (2, 4, -1, -1, -1, -1, -1))
else:
    raise SystemExit(f"unknown sweep mode {which!r}")
os.makedirs("gpurun_out", exist_ok=True)
json.dump(rows, open(f"gpurun_out/sweep_{which}.json", "w"), indent=1)
