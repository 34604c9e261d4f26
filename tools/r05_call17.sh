#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_qgemm_gpu.py tests/test_parity_sweep_gpu.py -x -q -m gpu -k "lean or hadamard or higgs or golden" > gpurun_out/r05/pytest_had.log 2>&1
tail -8 gpurun_out/r05/pytest_had.log | cut -c1-300
rm -f gpurun_out/r05/time_cases_k3584.jsonl
timeout 600 python tools/time_cases.py "4,1,4096,3584,f16;4,1,4096,3584,f16,one_shot=1;4,1,4096,3584,f16,one_shot=3;4,2,4096,3584,f16;4,2,4096,3584,f16,one_shot=1;4,4,4096,3584,f16;4,4,4096,3584,f16,family=0,one_shot=1;4,1,2048,3584,f16;4,1,8192,3584,f16;4,1,8192,3584,f16,one_shot=4;4,1,14336,3584,f16;4,1,14336,3584,f16,one_shot=4" --steps 300 --tag k3584 --out gpurun_out/r05/time_cases_k3584.jsonl > gpurun_out/r05/time_cases_k3584.log 2>&1
python - <<'PY'
import json
for l in open("gpurun_out/r05/time_cases_k3584.jsonl"):
    d = json.loads(l)
    print(d["bits"], d["M"], d["N"], d["K"], d["dtype"], d["ovr"], d["tid"], d["us"], "fam", d["plan"]["family"], "os", d["plan"]["one_shot"], d["plan"]["waves"], d["plan"]["grid"])
PY
python - <<'PY'
# configs[4]: pair codebook + Hadamard 512 on 3584 x 4096 (N = 4096, K = 3584), M = 1: the bench's line
import torch, bench
d = torch.device("cuda:0")
lay = bench.Layer(1, 4096, 3584, 4, 64, torch.float16, d, bench.copies_for(4096, 3584, 4), None, hadamard_size=512)
lay.tune()
us = min(bench.time_graph(lay, 300, 20, torch.cuda.synchronize)[0] for _ in range(3)) / 300 * 1e3
from flute_amd import utils
print("configs[4] 3584x4096 hadamard 512 M=1:", round(us, 3), "us tid", lay.template_id)
lay = bench.Layer(1, 3584, 4096, 4, 64, torch.float16, d, bench.copies_for(3584, 4096, 4), None, hadamard_size=512)
lay.tune()
us = min(bench.time_graph(lay, 300, 20, torch.cuda.synchronize)[0] for _ in range(3)) / 300 * 1e3
print("4096x3584 (N=3584, K=4096) hadamard 512 M=1:", round(us, 3), "us tid", lay.template_id)
PY
