"""Where does the 20-step replay lose 15 % against the 2000-step one?  Times the headline launch
HBM-cold in several ways on one box and prints one JSON object:
  * outer events around ONE graph replay of K steps, K in {20, 40, 80, 300, 2000} (what bench.py does);
  * the same with the start / end events recorded INSIDE the captured graph (event-record nodes): the K kernels alone,
    without whatever a graph launch costs before its first and after its last node;
  * slope: (t(2K) - t(K)) / K of the outer-event times = per-step time with the fixed part cancelled.
Development tool (not part of bench.py's contract)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    from flute_amd.nf_utils import NF4_VALUES
    lay = bench.Layer(1, 4096, 4096, 4, 64, torch.float16, dev, bench.copies_for(4096, 4096, 4), NF4_VALUES)
    lay.tune()
    out = {"template_id": lay.template_id}
    sync = torch.cuda.synchronize

    def outer(steps):
        best = 1e9
        for _ in range(3):
            ms, _ = bench.time_graph(lay, steps, 5, sync)
            best = min(best, ms)
        return best * 1e3 / steps, best * 1e3

    per = {}
    tot = {}
    for k in (20, 40, 80, 300, 2000):
        per[k], tot[k] = outer(k)
    out["outer_events_us_per_step"] = {str(k): round(v, 3) for k, v in per.items()}
    out["outer_events_total_us"] = {str(k): round(v, 2) for k, v in tot.items()}
    out["slope_us_per_step"] = {"20->40": round((tot[40] - tot[20]) / 20, 3), "40->80": round((tot[80] - tot[40]) / 40, 3),
                                "80->300": round((tot[300] - tot[80]) / 220, 3), "300->2000": round((tot[2000] - tot[300]) / 1700, 3)}
    out["fixed_us_per_replay"] = {str(k): round(tot[k] - k * out["slope_us_per_step"]["300->2000"], 2) for k in tot}

    # events recorded inside the capture
    try:
        res = {}
        for k in (20, 80):
            e0 = torch.cuda.Event(enable_timing=True, external=True)
            e1 = torch.cuda.Event(enable_timing=True, external=True)
            for i in range(5):
                lay.step(i)
            sync()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                e0.record()
                for i in range(k):
                    lay.step(5 + i)
                e1.record()
            g.replay(); sync()
            best = 1e9
            for _ in range(5):
                bench.flush_l3(dev)
                g.replay()
                sync()
                best = min(best, e0.elapsed_time(e1) * 1e3 / k)
            res[str(k)] = round(best, 3)
        out["in_graph_events_us_per_step"] = res
    except Exception as ex:  # noqa: BLE001
        out["in_graph_events_us_per_step"] = f"unsupported: {type(ex).__name__}: {ex}"[:300]

    # back-to-back launches through the C ABI from one host loop (no torch dispatcher, no graph)
    try:
        from flute_amd import _lib
        lib = _lib.get()
        import ctypes
        st = torch.cuda.current_stream().cuda_stream
        D = torch.empty(1, 4096, dtype=torch.float16, device=dev)
        res = {}
        for k in (20, 200):
            def run(n):
                for i in range(n):
                    c = i % len(lay.Q)
                    lib.flute_qgemm(0, 4, 64, 1, 4096, 4096, 1024, lay.X.data_ptr(), lay.Q[c].data_ptr(), D.data_ptr(), lay.S[c].data_ptr(),
                                    lay.table.data_ptr(), lay.table2.data_ptr(), lay.ws.data_ptr(), lay.ws.numel(), lay.template_id, lay.num_sms,
                                    ctypes.c_void_p(st))
            run(10); sync()
            best = 1e9
            for _ in range(3):
                bench.flush_l3(dev)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); run(k); e1.record(); sync()
                best = min(best, e0.elapsed_time(e1) * 1e3 / k)
            res[str(k)] = round(best, 3)
        out["c_abi_loop_us_per_step"] = res
    except Exception as ex:  # noqa: BLE001
        out["c_abi_loop_us_per_step"] = f"failed: {type(ex).__name__}: {ex}"[:300]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
