#!/bin/bash
# round 5, GPU call 22: (a) bench's 20-step line with the copies in one arena against separate tensors; (b) the Hadamard
# fuse threshold again now that a rotation costs half of what it did (M = 3, 4, 8: fused by override against the operator's two launches)
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for rep in 1 2; do
for a in 1 0; do
  FLUTE_BENCH_ARENA=$a python bench.py --steps 20 --warmup 5 --no-extras --no-cpu 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = j['roofline']
print('arena', $a, 'steps 20', j['value'], r.get('kernel_us'), r.get('kernel_us_hip_events'))"
done
done
FLUTE_BENCH_ARENA=1 python bench.py --steps 2000 --warmup 50 --no-extras --no-cpu 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = j['roofline']
print('arena 1 steps 2000', j['value'], r.get('kernel_us'), r.get('kernel_us_hip_events'))"
python - <<'PY'
import torch, bench
from flute_amd import dev
d = torch.device("cuda:0")
for (N, K) in ((4096, 3584), (4096, 4096), (3584, 4096), (14336, 3584)):
    for M in (2, 3, 4, 8):
        row = []
        for forced in (False, True):
            lay = bench.Layer(M, N, K, 4, 64, torch.float16, d, bench.copies_for(N, K, 4), None, hadamard_size=512)
            lay.template_id = 16
            if forced: lay.ovr = dev.Overrides(family=0)
            try:
                us = min(bench.time_graph(lay, 300, 20, torch.cuda.synchronize)[0] for _ in range(3)) / 300 * 1e3
            except Exception as e:
                us = float("nan")
            row.append(round(us, 3))
            del lay; torch.cuda.empty_cache()
        print("had512 M", M, N, K, "operator", row[0], "forced fused", row[1], flush=True)
PY
