"""Run one (shape, overrides) case N times for rocprofv3 (development aid)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from flute_amd import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--M", type=int, default=1)
ap.add_argument("--N", type=int, default=4096)
ap.add_argument("--K", type=int, default=4096)
ap.add_argument("--bits", type=int, default=4)
ap.add_argument("--g", type=int, default=64)
ap.add_argument("--tid", type=int, default=16)
ap.add_argument("--bf16", action="store_true")
ap.add_argument("--ovr", type=str, default="-1,-1,-1,-1,-1,-1,-1,-1")
ap.add_argument("--steps", type=int, default=60)
ap.add_argument("--hot", action="store_true")
a = ap.parse_args()
dev = torch.device("cuda:0")
dtype = torch.bfloat16 if a.bf16 else torch.float16
lay = bench.Layer(a.M, a.N, a.K, a.bits, a.g, dtype, dev, 1 if a.hot else bench.copies_for(a.N, a.K, a.bits))
lay.template_id = a.tid
from flute_amd.dev import overrides_from_tuple  # noqa: E402
lay.ovr = overrides_from_tuple(a.ovr.split(","))
for i in range(a.steps):
    lay.step(i)
torch.cuda.synchronize()
