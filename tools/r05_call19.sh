#!/bin/bash
# round 5, GPU calls 19 and 20: the lane-stage rewrite of the transform (DPP + lane swaps instead of ds_bpermute): tests, then the
# Hadamard timings of call 18 again (lean kernel with the fused rotation: ids 16/0; round-4 one-shot kernel: ids 17/1)
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -k "hadamard or higgs or lean or golden or fwht" 2>&1 | tail -5
python - <<'PY'
import torch, bench
from flute_amd import dev, utils
d = torch.device("cuda:0")
for (N, K) in ((4096, 3584), (3584, 4096), (4096, 4096)):
    for M in (1, 2):
        if M * K > 8192: continue
        for tid in (16, 17, 1, 0):
            lay = bench.Layer(M, N, K, 4, 64, torch.float16, d, bench.copies_for(N, K, 4), None, hadamard_size=512)
            lay.template_id = tid
            us = min(bench.time_graph(lay, 300, 20, torch.cuda.synchronize)[0] for _ in range(3)) / 300 * 1e3
            print("had512", M, N, K, "tid", tid, round(us, 3), flush=True)
            del lay; torch.cuda.empty_cache()
# the stand-alone transform
import flute_amd
for rows, n, h in ((1, 3584, 512), (1, 4096, 4096), (16, 4096, 512), (256, 4096, 4096), (4096, 4096, 4096)):
    x = torch.randn(rows, n, dtype=torch.float16, device=d)
    y = torch.empty_like(x)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        from flute_amd import ops as hm
        for _ in range(3): hm.hadamard_transform(x, h)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(100): hm.hadamard_transform(x, h)
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 10)
    print("fwht", rows, n, h, round(best, 2), "us", round(rows * n * 4 / best / 1e3, 1), "GB/s", flush=True)
PY
