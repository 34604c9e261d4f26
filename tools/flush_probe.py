"""Why is a 20-step replay right behind the cache flush slower per step than a 300-step one?  (development probe)"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from flute_amd.nf_utils import NF4_VALUES  # noqa: E402

d = torch.device("cuda:0")
lay = bench.Layer(1, 4096, 4096, 4, 64, torch.float16, d, bench.copies_for(4096, 4096, 4), NF4_VALUES)
lay.template_id = 16
out = {}
for steps in (20, 40, 80, 300):
    for i in range(5):
        lay.step(i)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(steps):
            lay.step(5 + i)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()

    def timed(pre):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        pre()
        s.record(); g.replay(); e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / steps * 1e3

    buf = torch.zeros(2 * bench.L3_BYTES // 4, dtype=torch.int32, device=d)
    sink = torch.zeros(1, dtype=torch.int64, device=d)
    big = torch.empty(3 * bench.L3_BYTES, dtype=torch.uint8, device=d)
    rec = {}
    rec["no_flush"] = min(timed(lambda: None) for _ in range(3))
    rec["flush_sum"] = min(timed(lambda: sink.add_(buf.sum())) for _ in range(3))
    rec["flush_sum_then_sync"] = min(timed(lambda: (sink.add_(buf.sum()), torch.cuda.synchronize())) for _ in range(3))
    rec["flush_memset"] = min(timed(lambda: big.zero_()) for _ in range(3))
    rec["flush_sum_then_one_launch"] = min(timed(lambda: (sink.add_(buf.sum()), lay.step(0))) for _ in range(3))
    rec["flush_sum_then_replay_untimed"] = min(timed(lambda: (sink.add_(buf.sum()), g.replay())) for _ in range(3))
    rec["two_replays_back_to_back_second_timed"] = min(timed(lambda: g.replay()) for _ in range(3))
    out[steps] = {k: round(v, 3) for k, v in rec.items()}
    print(steps, json.dumps(out[steps]), flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/flush_probe.json", "w"), indent=1)
