#!/bin/bash
set -u
cd "$(dirname "$0")/.."
python - <<'PY'
import torch, bench
from flute_amd import dev, utils
d = torch.device("cuda:0")
for (N, K) in ((4096, 3584), (3584, 4096)):
    for M in (1, 2):
        if M * K > 8192: continue
        for tid in (16, 17, 1, 0, 18):
            lay = bench.Layer(M, N, K, 4, 64, torch.float16, d, bench.copies_for(N, K, 4), None, hadamard_size=512)
            lay.template_id = tid
            us = min(bench.time_graph(lay, 300, 20, torch.cuda.synchronize)[0] for _ in range(3)) / 300 * 1e3
            print("had512", M, N, K, "tid", tid, round(us, 3), flush=True)
            del lay; torch.cuda.empty_cache()
        lay = bench.Layer(M, N, K, 4, 64, torch.float16, d, bench.copies_for(N, K, 4), None)
        lay.template_id = 16
        us = min(bench.time_graph(lay, 300, 20, torch.cuda.synchronize)[0] for _ in range(3)) / 300 * 1e3
        print("plain ", M, N, K, "tid 16", round(us, 3), flush=True)
        del lay; torch.cuda.empty_cache()
PY
