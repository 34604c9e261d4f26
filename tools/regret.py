"""Planner regret sweep: for every (layer shape, bit width, M) time the AUTOMATIC plan (the shipped table's template id,
no override - what flute.qgemm launches) against every kernel family / launch shape that can be forced for it through
`flute_overrides`, and report how far the automatic plan is from the best one found.

    python tools/regret.py [--shapes baseline|supported] [--bits 4,3,2] [--ms 1,2,4,16,64,256,1024] [--budget-s 900] [--out FILE]

The role of flute/tune.py:205-257 (the reference times every template of its generated switch per shape): here the
template table is small and the planner's thresholds carry most of the decision, so what needs measuring is the planner.
HBM-cold (weight copies rotate through > 256 MiB), hipGraph replay, best of two.  Writes one JSON line per case and a
summary (worst cases first) to gpurun_out/planner_regret.json."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import flute_amd  # noqa: E402
from flute_amd import dev, tune, utils  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--shapes", default="baseline")
ap.add_argument("--bits", default="4,3,2")
ap.add_argument("--ms", default="1,2,4,16,64,256,1024")
ap.add_argument("--dtype", default="float16")
ap.add_argument("--budget-s", type=float, default=900.0)
ap.add_argument("--steps", type=int, default=100)
ap.add_argument("--out", default="gpurun_out/planner_regret.json")
a = ap.parse_args()

d = torch.device("cuda:0")
num_sms = utils.get_device_num_sms(d)
dtype = torch.float16 if a.dtype == "float16" else torch.bfloat16
BASELINE = [(4096, 4096), (11008, 4096), (4096, 11008), (8192, 8192), (28672, 8192), (8192, 28672), (3584, 8192), (14336, 3584), (3584, 14336),
            (14336, 4096), (4096, 14336), (6144, 4096), (10240, 8192)]
shapes = BASELINE if a.shapes == "baseline" else (tune.EXTRA_SHAPES + tune.SUPPORTED_SHAPES if a.shapes == "supported"
                                                  else [tuple(int(v) for v in s.split(",")) for s in a.shapes.split(";")])


def base_tid(bits, tile_p):
    return min(t for (b, t), c in flute_amd.TEMPLATE_CONFIGS.items() if b == bits and c["TileP"] == tile_p)


def candidates(bits, M):
    c = []
    if M <= 4:
        c += [dict(family=0, one_shot=0), dict(family=0, one_shot=1), dict(family=0, one_shot=1, ring_depth=4), dict(family=0, one_shot=3)]
    if M >= 3:
        for mb in (1, 2, 4):
            for sw in (1, 2):
                c.append(dict(family=2, m_block=mb, slabs_per_wave=sw))
        c += [dict(family=2, m_block=1, splitk=2), dict(family=2, m_block=1, slabs_per_wave=2, splitk=2), dict(family=2, m_block=2, splitk=2),
              dict(family=2, m_block=1, kw=4), dict(family=2, m_block=1, kw=2)]
    if 3 <= M <= 16 and bits == 4:
        c += [dict(family=5, splitk=sk) for sk in (1, 2, 4, 8)]
        c += [dict(family=7, slabs_per_wave=ng) for ng in (1, 2, 3)]          # lean MFMA decode kernel: column groups per workgroup (round 6)
    if 3 <= M <= 16 and bits in (2, 4):
        c += [dict(family=8, slabs_per_wave=ng) for ng in (1, 2, 3)]          # persistent MFMA decode kernel (4- and 2-bit members): column groups per set (round 6)
    if bits == 3 and 17 <= M <= 64:
        c += [dict(family=3, m_block=4), dict(family=3, m_block=2)]
    if bits == 3 and M > 64:
        c += [dict(family=3, m_tiles=4, splitk=sk) for sk in (2, 4)] + [dict(family=3, m_block=4, splitk=sk) for sk in (1, 2, 4)] + [dict(family=2)]
    if M >= 128:
        c += [dict(family=3, m_tiles=4), dict(family=3, m_tiles=8)]
        if bits != 3:
            c += [dict(family=6, splitk=sk) for sk in (1, 2, 4, 8)]
    if M >= 33 and bits != 3:                                                  # split-K block kernel: every tile form x K slices (round 6: 64 x 64 tiles)
        c += [dict(family=6, splitk=sk, kw=kw, m_tiles=rt) for (kw, rt) in ((4, 4), (2, 4), (2, 8)) for sk in (1, 2, 4)]
    return c


rows = []
t0 = time.time()
os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
skipped = 0
for (N, K) in shapes:
    for bits in [int(v) for v in a.bits.split(",")]:
        J = 16 if bits == 3 else 16 // bits
        if N % (J * 32):
            continue
        lay = None
        for M in [int(v) for v in a.ms.split(",")]:
            if time.time() - t0 > a.budget_s:
                skipped += 1
                continue
            try:
                if lay is None or lay.M != M:
                    del lay
                    torch.cuda.empty_cache()
                    lay = bench.Layer(M, N, K, bits, 64, dtype, d, bench.copies_for(N, K, bits))
                lay.ovr = None
                tid = lay.tune()
                tile_p = flute_amd.TEMPLATE_CONFIGS[(bits, tid)]["TileP"]
                steps = a.steps if M * N * K < (1 << 36) else max(20, a.steps // 4)
                auto_plan = dev.get_plan(M, N, K, bits, 64, tid, num_sms, dtype)
                auto_us = min(bench.time_graph(lay, steps, 3, torch.cuda.synchronize)[0] for _ in range(2)) / steps * 1e3
                best_us, best_ovr, tried = auto_us, None, 0
                seen = {json.dumps({k: auto_plan[k] for k in sorted(auto_plan)})}
                lay.template_id = base_tid(bits, tile_p)
                for shp in candidates(bits, M):
                    ovr = dev.Overrides(**shp)
                    try:
                        plan = dev.get_plan(M, N, K, bits, 64, lay.template_id, num_sms, dtype, ovr)
                    except RuntimeError:
                        continue
                    key = json.dumps({k: plan[k] for k in sorted(plan)})
                    if key in seen or ("family" in shp and plan["family"] != shp["family"] and not (shp["family"] in (1, 2) and plan["family"] == 2)):
                        continue
                    seen.add(key)
                    lay.ovr = ovr
                    try:
                        us = min(bench.time_graph(lay, steps, 3, torch.cuda.synchronize)[0] for _ in range(2)) / steps * 1e3
                    except RuntimeError:
                        continue
                    tried += 1
                    if us < best_us:
                        best_us, best_ovr = us, dict(shp, plan={k: plan[k] for k in ("family", "m_block", "m_tiles", "slabs_per_wave", "waves", "kw", "splitk", "grid", "one_shot")})
                rec = {"N": N, "K": K, "bits": bits, "M": M, "tid": tid, "auto_us": round(auto_us, 2),
                       "auto_plan": {k: auto_plan[k] for k in ("family", "m_block", "m_tiles", "slabs_per_wave", "waves", "kw", "splitk", "grid", "one_shot")},
                       "best_us": round(best_us, 2), "best": best_ovr, "regret_pct": round((auto_us / best_us - 1) * 100, 1), "plans_tried": tried}
            except Exception as ex:  # noqa: BLE001
                rec = {"N": N, "K": K, "bits": bits, "M": M, "error": str(ex)[:200]}
            rows.append(rec)
            print(json.dumps(rec), flush=True)
        del lay
        lay = None
        torch.cuda.empty_cache()
ok = [r for r in rows if "regret_pct" in r]
summary = {"cases": len(ok), "skipped_for_time": skipped, "seconds": round(time.time() - t0, 1),
           "over_10pct": sum(1 for r in ok if r["regret_pct"] > 10), "over_5pct": sum(1 for r in ok if r["regret_pct"] > 5),
           "mean_regret_pct": round(sum(r["regret_pct"] for r in ok) / max(1, len(ok)), 2),
           "worst": sorted(ok, key=lambda r: -r["regret_pct"])[:25]}
json.dump({"summary": summary, "rows": rows}, open(a.out, "w"), indent=1)
print(json.dumps({k: v for k, v in summary.items() if k != "worst"}))
