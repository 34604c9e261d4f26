import json, os, sys, torch
sys.path.insert(0, os.getcwd())
import bench
from flute_amd import dev, utils
d = torch.device("cuda:0"); f16 = torch.float16
for (M, N, K) in ((64, 4096, 4096), (32, 4096, 4096), (16, 1024, 8192), (64, 1024, 8192), (128, 2048, 4096)):
    lay = bench.Layer(M, N, K, 4, 64, f16, d, bench.copies_for(N, K, 4)); lay.tune()
    pl = dev.get_plan(M, N, K, 4, 64, lay.template_id, lay.num_sms, f16)
    ms = min(bench.time_graph(lay, 300, 5, torch.cuda.synchronize)[0] for _ in range(2))
    print(json.dumps({"M": M, "N": N, "K": K, "tid": lay.template_id, "family": pl["family"], "splitk": pl["splitk"], "splitk_mode": pl["splitk_mode"], "us": round(ms / 300 * 1e3, 2)}), flush=True)
