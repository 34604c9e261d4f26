import sys, json, torch
sys.path.insert(0, '.')
import bench
from flute_amd import utils
dev = torch.device("cuda:0")
for M in (8, 32, 64, 128):
    for (n, k) in ((4096, 4096), (11008, 4096)):
        lay = bench.Layer(M, n, k, 4, 64, torch.float16, dev, bench.copies_for(n, k, 4), None)
        tid = lay.tune()
        ms, _ = bench.time_graph(lay, 300, 20, torch.cuda.synchronize)
        p = utils.get_plan(M, n, k, 4, 64, tid, lay.num_sms, torch.float16)
        print(M, n, "tid", tid, "us", round(ms / 300 * 1e3, 2), {k_: p[k_] for k_ in ("family", "m_block", "m_tiles", "slabs_per_wave", "waves", "kw", "splitk", "grid")}, flush=True)
        del lay; torch.cuda.empty_cache()
from flute_amd import _lib
lib = _lib.get()
for rep in range(3):
    lay = bench.Layer(32, 4096, 4096, 4, 64, torch.float16, dev, bench.copies_for(4096, 4096, 4))
    lay.template_id = 16
    from flute_amd import dev
    lay.ovr = dev.Overrides(family=2, m_block=2, waves=8, kw=4, splitk=1, m_tiles=1)
    ms, _ = bench.time_graph(lay, 300, 10, torch.cuda.synchronize)
    print("odd config rep", rep, round(ms / 300 * 1e3, 2), utils.get_plan(32, 4096, 4096, 4, 64, 16, lay.num_sms, torch.float16))
