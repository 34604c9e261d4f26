#!/usr/bin/env python
"""Benchmark of the qgemm hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

One "step" = one `flute.qgemm` launch (M=1, W4G64 NF4 table, fp16, K=N=4096 -
BASELINE.json configs[1], the shape the metric's target is quoted on) over one
packed weight copy.  Steps rotate over enough distinct (Q, S) copies to exceed
the 256 MiB Infinity Cache, so every step streams its weights from HBM; inputs
are resident in HBM before the timed region.  The K steps are enqueued as one
hipGraph replay (how a decode loop launches them; host launch overhead of the
Python op is reported separately as `eager_us_per_step`).

Rank 0 prints ONE JSON line (contract in the task description) carrying
`roofline` (HBM, algorithmic bytes / kernel time from HIP events on the launch
stream) and `cpu_baseline` (the CPU oracle = the reference's test formula,
dequant + torch.mm, timed on this host's cores).

`--gpus N` under a launcher (the driver's `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`)
must agree with WORLD_SIZE; with no launcher around it the script starts the N ranks itself (re-exec under
torch.distributed.run, rendezvous on 127.0.0.1).  The line carries the rank count summed over RCCL.
N > 1 (one process per GPU, RCCL): the headline stays the same workload - N
independent replicas, no collective, barrier + max-over-ranks, `value` = N x
bytes / time, "scaling": "weak" - so that the per-N values are comparable.
BASELINE.json configs[3] - the Llama-3-70B MLP pair 8192x28672 -> 28672x8192
sharded N-way (column-parallel up projection, row-parallel down projection,
ONE all-reduce of M*8192*2 B), launches AND collective captured in one
hipGraph - is reported under `tp_mlp_pair` (strong scaling; N = 1: TP = 1),
next to the other BASELINE configs, prefill shapes with torch.mm fp16 beside
them, and the CPU baseline.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
HBM_COPY_GBPS = 6290.0          # measured float4 copy
MFMA_PEAK_TFLOPS = 2500.0       # dense fp16/bf16
L3_BYTES = 256 * 1024 * 1024


def algorithmic_bytes(M, N, K, bits, g):
    """Q + S + X + Y + tables."""
    P = bits * N // 16
    return 2 * P * K + 2 * N * K // g + 2 * M * K + 2 * M * N + 2 * 2 ** bits + 4 * 4 ** bits


class Layer:
    """`copies` independent packed layers of one shape, all resident in HBM."""

    def __init__(self, M, N, K, bits, g, dtype, device, copies, table_values=None, seed=0,
                 hadamard_size=0):
        import flute_amd
        from flute_amd import utils
        self.M, self.N, self.K, self.bits, self.g, self.dtype = M, N, K, bits, g, dtype
        gen = torch.Generator(device="cpu").manual_seed(seed)
        if table_values is None:
            table = torch.randn(2 ** bits, generator=gen)
        else:
            table = torch.tensor(table_values)
        self.table = table.to(dtype).to(device)
        self.table2 = utils.make_qmap2_from_qmap(self.table)
        self.X = (torch.randn(M, K, generator=gen) / 100).to(dtype).to(device)
        self.num_sms = utils.get_device_num_sms(device)
        self.ws = utils.get_workspace_streamk(device)
        P = bits * N // 16
        self.Q, self.S = [], []
        gdev = torch.Generator(device=device).manual_seed(seed)
        # One allocation for all the copies of Q and one for S (a checkpoint's layers loaded into one arena), not `copies`
        # separate tensors from the caching allocator's 20-MB blocks: FLUTE_BENCH_ARENA=0 gives the latter (A/B)
        arena = os.environ.get("FLUTE_BENCH_ARENA", "1") != "0"
        if arena:
            q_all = torch.empty((copies, P, K), dtype=torch.int16, device=device)
            s_all = torch.empty((copies, N, K // g), dtype=dtype, device=device)
        for c in range(copies):
            # any bit pattern is a valid packed matrix (all codes reachable for b=2,4;
            # for b=3 too): uniform int16 == uniform codes
            q = torch.randint(-2 ** 15, 2 ** 15, (P, K), dtype=torch.int16, device=device, generator=gdev)
            sc = torch.randn(N, K // g, device=device, generator=gdev).to(dtype)
            if arena:
                q_all[c].copy_(q)
                s_all[c].copy_(sc)
                q, sc = q_all[c], s_all[c]
            self.Q.append(q)
            self.S.append(sc)
        self.template_id = None
        self.qgemm = flute_amd.qgemm
        self.hadamard_size = hadamard_size      # > 0: flute.qgemm_hadamard (rotation fused into the decode kernel)
        self.qgemm_hadamard = flute_amd.qgemm_hadamard
        self.ovr = None                         # development sweeps: per-call plan override (flute_amd.dev.Overrides)

    def bytes(self):
        return algorithmic_bytes(self.M, self.N, self.K, self.bits, self.g)

    def flops(self):
        return 2 * self.M * self.N * self.K

    def tune(self):
        from flute_amd import tune
        self.template_id = tune._tune(self.M, self.N, self.K, self.bits, self.g, self.num_sms,
                                      self.dtype, self.X.device, num_seeds=1, rep=120)
        return self.template_id

    def step(self, i):
        c = i % len(self.Q)
        if self.ovr is not None:
            from flute_amd import dev
            return dev.qgemm_planned(self.X, self.Q[c], self.S[c], self.table, self.table2, self.ws, self.bits,
                                     self.g, self.template_id, self.num_sms, self.ovr, self.hadamard_size)
        if self.hadamard_size:
            return self.qgemm_hadamard(self.X, self.Q[c], self.S[c], self.table, self.table2, self.ws,
                                       self.bits, self.g, self.hadamard_size, self.template_id, self.num_sms)
        return self.qgemm(self.X, self.Q[c], self.S[c], self.table, self.table2, self.ws,
                          self.bits, self.g, self.template_id, self.num_sms)


def copies_for(N, K, bits, cap_bytes=3 << 30):
    per = 2 * (bits * N // 16) * K
    n = L3_BYTES // per + 2
    return int(max(2, min(n, cap_bytes // per)))


_FLUSH = {}


def flush_l3(device):
    """Untimed: push everything out of the 256 MiB Infinity Cache (and the L2s) by READING a 512 MiB scratch
    buffer (read-only: no dirty lines whose write-back would compete with the timed reads), so that the timed
    replay streams its weights from HBM whatever --steps is (20 steps of the headline touch 178 MB: without this
    the warm replays would leave them cache-resident).  Enqueued on the replay stream right in front of the start
    event, no host synchronisation in between: an idle gap would let the chip clock down, and a 20-step replay
    (90 us) is shorter than the ramp back up."""
    buf = _FLUSH.get(device)
    if buf is None:
        buf = _FLUSH[device] = torch.zeros(2 * L3_BYTES // 4, dtype=torch.int32, device=device)
        _FLUSH[(device, "sink")] = torch.zeros(1, dtype=torch.int64, device=device)
        torch.cuda.synchronize()
    _FLUSH[(device, "sink")].add_(buf.sum())


LAST_TIMING = {}        # of the latest time_graph call: the HIP-event time of the same replay, the clock used


def _timestamp(ts, slot):
    """Enqueue (capture) a one-lane kernel that writes the chip-wide 100 MHz clock to ts[slot] (flute_debug_timestamp)."""
    from flute_amd import _lib
    import ctypes
    rc = _lib.get().flute_debug_timestamp(ctypes.c_void_p(ts.data_ptr() + 8 * slot),
                                          ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    if rc:
        raise RuntimeError(f"flute_debug_timestamp: {rc}")


def time_graph(layer, steps, warmup, sync, cold=True, replays=1):
    """Capture `steps` launches in one hipGraph, bracketed INSIDE the graph by two device-clock stamps (one-lane kernels
    writing the chip-wide 100 MHz clock: the first runs when everything before it has finished, the second when the last
    step has), replay once.  Returns (ms between the stamps, host wall ms around sync()).  HIP events around the same replay
    are recorded too (LAST_TIMING["events_ms"]): a graph launch bracketed by events carries a FIXED 16 - 19 us of launch /
    marker overhead per replay whatever it holds (profiles/r05/graph_replay_fixed_cost_probe.json: 97.6 us for 20 steps,
    183.0 for 40, 8190 for 2000 - slope 4.09 us per step at every length), i.e. +0.8 us per step at 20 steps and nothing
    at 2000; the stamps see the launches themselves.  cold: flush the caches (untimed) before the timed replay.
    replays > 1: that many timed replays of the same K steps, each behind its own flush and barrier; the MEDIAN replay is returned,
    every replay's time is kept in LAST_TIMING (median + min)."""
    for i in range(warmup):
        layer.step(i)
    dev = torch.device("cuda", torch.cuda.current_device())
    if cold:
        flush_l3(dev)     # first call: allocation + kernel load, out of the way
    ts = torch.zeros(2, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        _timestamp(ts, 0)
        for i in range(steps):
            layer.step(warmup + i)
        _timestamp(ts, 1)
    # untimed: the first replay pays the upload; then ~30 ms of replays so that the chip is at its sustained clocks
    # when the timed replay starts (a 20-step replay lasts 90 us - far shorter than a clock ramp; measured: 5.9 us
    # per step straight after the tuner's idle gap, 4.9 us after a sustained warm-up)
    graph.replay()
    torch.cuda.synchronize()
    w0, w1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    w0.record(); graph.replay(); w1.record()
    torch.cuda.synchronize()
    for _ in range(max(2, min(5000, int(30.0 / max(w0.elapsed_time(w1), 1e-3))))):
        graph.replay()
    torch.cuda.synchronize()
    runs = []
    for _ in range(max(1, replays)):
        start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        sync()
        t0 = time.perf_counter()
        if cold:
            flush_l3(dev)     # stream-ordered, in front of the start event
        else:
            graph.replay()                  # untimed spacer: the GPU stays busy while the host enqueues the timed replay
        start.record()
        graph.replay()
        end.record()
        torch.cuda.synchronize()
        sync()
        wall_ms = (time.perf_counter() - t0) * 1e3
        ev_ms = start.elapsed_time(end)
        t = ts.cpu()
        dev_ms = float(int(t[1]) - int(t[0])) * 1e-5            # ticks of 10 ns
        ok = 0.0 < dev_ms <= ev_ms * 1.001 + 0.01                # a stamp that did not run: fall back on the events, and say so
        runs.append((dev_ms if ok else ev_ms, wall_ms, ev_ms, dev_ms, ok))
    runs.sort(key=lambda r: r[0])
    best, wall_ms, ev_ms, dev_ms, ok = runs[len(runs) // 2]      # the median replay
    LAST_TIMING.clear()
    LAST_TIMING.update({"events_ms": ev_ms, "device_clock_ms": dev_ms, "replays_ms": [round(r[0], 6) for r in runs],
                        "min_ms": runs[0][0], "median_ms": best})
    LAST_TIMING["clock"] = "device clock stamps inside the graph" if ok else "hip events (device stamps implausible)"
    return best, wall_ms


def time_eager(layer, steps, warmup):
    for i in range(warmup):
        layer.step(i)
    torch.cuda.synchronize()
    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    for i in range(steps):
        layer.step(i)
    end.record()
    torch.cuda.synchronize()
    return start.elapsed_time(end)


def cpu_baseline(N, K, bits, g, tile_p=32, runs=36):
    """The reference's CPU-runnable case (BASELINE.json configs[0]): closed-form unpack
    of Q, LUT dequant, torch.matmul - i.e. the oracle, timed on the host cores.  Bounded
    sample: `runs` repetitions of the headline layer split over three thread counts (all
    cores, 32, 8 - a 256-core host runs this 16 M-element gather + GEMV several times SLOWER
    on all cores than on a few); the best one is reported with its thread count as `cores`
    (about 10 s of CPU work)."""
    from oracle import flute_oracle as O
    import numpy as np
    dtype = torch.float16
    rng = np.random.default_rng(0)
    W = rng.integers(0, 2 ** bits, size=(K, N), dtype=np.uint8)
    Q = O.pack(W, bits, tile_p)
    S = torch.randn(N, K // g).to(dtype)
    from flute_amd.nf_utils import NF4_VALUES
    table = torch.tensor(NF4_VALUES).to(dtype)
    X = (torch.randn(1, K) / 100).to(dtype)
    t0 = time.perf_counter()
    codes = torch.from_numpy(O.unpack(Q, bits, tile_p).astype(np.int64))
    t_unpack = time.perf_counter() - t0

    def run():
        W_ = table[codes]
        S_ = torch.repeat_interleave(S, g, dim=1).T
        return torch.mm(X, W_ * S_)      # tests/kernel.py:68-71

    # every host core is not the fastest way to run a 16 M-element gather + GEMV: take the best thread count
    ncpu = os.cpu_count() or 1
    per_threads = {}
    for nt in sorted({ncpu, min(ncpu, 32), min(ncpu, 8)}, reverse=True):
        torch.set_num_threads(nt)
        run()
        ts = []
        for _ in range(max(3, runs // 3)):
            t0 = time.perf_counter()
            run()
            ts.append(time.perf_counter() - t0)
        per_threads[nt] = min(ts)
    cores, t = min(per_threads.items(), key=lambda kv: kv[1])
    torch.set_num_threads(ncpu)
    return {
        "value": round(algorithmic_bytes(1, N, K, bits, g) / t / 1e9, 4), "unit": "GB/s",
        "cores": cores, "kind": "port",
        "sample": f"{max(3, runs // 3)} x (LUT dequant + torch.mm) per thread count, M=1 K={K} N={N} W{bits}G{g} fp16, "
                  f"codes pre-unpacked; best {t * 1e3:.1f} ms with {cores} threads of {ncpu} host cores ("
                  + ", ".join(f"{k} thr: {v * 1e3:.0f} ms" for k, v in per_threads.items())
                  + f"); closed-form unpack of Q alone {t_unpack * 1e3:.0f} ms",
        "ms": round(t * 1e3, 2), "unpack_ms": round(t_unpack * 1e3, 1),
    }


def tp_mlp_pair(world, rank, device, dist, pairs=50, warmup=5):
    """BASELINE.json configs[3] on `world` GPUs: the Llama-3-70B MLP up projection 8192 -> 28672 N-sharded
    (column-parallel, no collective) feeding the down projection 28672 -> 8192 K-sharded (row-parallel, ONE
    all-reduce of M*8192*2 B over RCCL/xGMI) - flute_amd/tp.py, reference contract vllm_utils.py:224-226,
    265-326.  Each rank builds its own shard directly (random packed data).  `pairs` pairs are captured in ONE
    hipGraph - launches AND the all-reduce - and the replay is timed (max over ranks); the kernels alone are
    timed the same way.  Falls back to eager launches if the collective cannot be captured."""
    bits, g, dtype, M, H, F = 4, 64, torch.float16, 1, 8192, 28672
    if F % (world * 256):
        return {"skipped": f"28672 does not split {world}-way on packed column blocks"}
    ncopies = max(2, (L3_BYTES // (2 * (bits * (F // world) // 16) * H)) // 2 + 2)
    up = Layer(M, F // world, H, bits, g, dtype, device, ncopies, None, seed=rank)
    down = Layer(M, H, F // world, bits, g, dtype, device, ncopies, None, seed=rank + 100)
    up.tune()
    down.tune()

    def body(collective, n):
        for i in range(n):
            up.step(i)
            y = down.step(i)
            if collective:
                dist.all_reduce(y)

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier(device_ids=[device.index])

    def run(collective):
        body(collective, max(1, warmup))
        sync()
        mode = "hipGraph"
        try:
            graph = torch.cuda.CUDAGraph()
            # thread_local: the process group's watchdog thread may touch the runtime while this thread captures
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                body(collective, pairs)
            graph.replay()
            sync()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            # ONE timed replay of exactly `pairs` steps, bracketed by barrier + synchronize on both sides
            s.record(); graph.replay(); e.record()
            torch.cuda.synchronize()
            best = s.elapsed_time(e) / pairs * 1e3
            sync()
        except Exception:                                   # noqa: BLE001 - capture of the collective unsupported
            torch.cuda.synchronize()
            mode = "eager"
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            sync()
            s.record(); body(collective, pairs); e.record()
            torch.cuda.synchronize()
            best = s.elapsed_time(e) / pairs * 1e3
        t = torch.tensor([best], device=device, dtype=torch.float64)
        if dist is not None:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return t.item(), mode

    k_us, k_mode = run(False)
    if dist is not None:
        kc_us, c_mode = run(True)
    else:
        kc_us, c_mode = k_us, k_mode
    nbytes = world * (up.bytes() + down.bytes())
    return {"workload": f"W4G64 fp16 M=1 Llama-3-70B MLP pair: 8192x28672 column-parallel -> 28672x8192 row-parallel, TP={world}",
            "kernels_us": round(k_us, 3), "kernels_plus_allreduce_us": round(kc_us, 3),
            "allreduce_bytes": 2 * M * H, "launch": f"kernels: {k_mode}; with all-reduce: {c_mode}",
            "whole_job_bytes": nbytes,
            "whole_job_GBps_kernels": round(nbytes / k_us / 1e3, 1),
            "whole_job_GBps_with_allreduce": round(nbytes / kc_us / 1e3, 1),
            "frac_hbm_8TBps_per_gpu_with_allreduce": round(nbytes / kc_us / 1e3 / world / HBM_PEAK_GBPS, 4),
            "template_ids": [up.template_id, down.template_id]}


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _self_launch(n):
    """`python bench.py --gpus N` with no launcher around it (WORLD_SIZE unset): start the N ranks here - replace this
    process by `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same arguments>`, one rank per
    GPU, rendezvous on 127.0.0.1 (the container's hostname may not resolve).  Under a launcher (the driver's torchrun
    line) WORLD_SIZE is set and this is never reached."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC only on this driver (RCCL needs it)
    env["FLUTE_BENCH_SELF_LAUNCHED"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    os.execve(sys.executable, cmd, env)


def _dry_run_ranks(args, world, rank):
    """FLUTE_BENCH_DRYRUN=1 (tests/test_host.py, no GPU): the rank plumbing only - every rank joins a gloo group, the
    rank count the process group reports and a sum of ones over the ranks go into the line.  No kernel runs and the line
    says so (`dry_run`); never a measurement."""
    import torch.distributed as dist
    dist.init_process_group("gloo")
    ones = torch.ones(1, dtype=torch.int64)
    dist.all_reduce(ones)
    assert dist.get_world_size() == args.gpus == world, (dist.get_world_size(), args.gpus, world)
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "ranks_in_group": dist.get_world_size(),
                          "ranks_counted_by_allreduce": int(ones.item()),
                          "self_launched": os.environ.get("FLUTE_BENCH_SELF_LAUNCHED") == "1"}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None,
                    help="ranks = GPUs of this node (default: WORLD_SIZE, else 1); with no launcher around it the script "
                         "starts the ranks itself")
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--template-id", type=int, default=-1,
                    help="skip the tuner and use this template id (profiling runs: keeps the tuner's "
                         "candidate launches out of the kernel trace)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and (args.gpus or 1) > 1:
        _self_launch(args.gpus)                      # does not return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus is None:
        args.gpus = world
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if os.environ.get("FLUTE_BENCH_DRYRUN") == "1":
        return _dry_run_ranks(args, world, rank)
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no CPU fallback)"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # FLUTE_BENCH_FORCE_DIST=1: take the process-group path on one GPU too (exercises the TP leg under torchrun)
    saved_stdout = None
    if world > 1 or os.environ.get("FLUTE_BENCH_FORCE_DIST") == "1":
        # RCCL prints a version banner on STDOUT when a communicator is created: route fd 1 to stderr until the one
        # JSON line is due, so that stdout carries that line and nothing else
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=device)
        assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)

        def sync():
            dist.barrier(device_ids=[local_rank])
            torch.cuda.synchronize()
    else:
        dist = None

        def sync():
            torch.cuda.synchronize()

    from flute_amd.nf_utils import NF4_VALUES        # the oracle is imported by the cpu_baseline leg only
    M, N, K, bits, g, dtype = 1, 4096, 4096, 4, 64, torch.float16
    layer = Layer(M, N, K, bits, g, dtype, device, copies_for(N, K, bits), NF4_VALUES, seed=rank)
    if args.template_id >= 0:
        tid = layer.template_id = args.template_id
    else:
        tid = layer.tune()
    ev_ms, wall_ms = time_graph(layer, args.steps, args.warmup, sync, replays=5)      # five timed replays of the K steps: the median is the line's figure
    headline_timing = dict(LAST_TIMING)
    eager_ms = time_eager(layer, min(args.steps, 500), 10)
    t = torch.tensor([ev_ms, wall_ms, headline_timing["events_ms"], headline_timing["min_ms"]], device=device, dtype=torch.float64)
    ranks_counted = 1
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ones = torch.ones(1, device=device, dtype=torch.int64)
        dist.all_reduce(ones)                           # every rank adds 1 over RCCL: the rank count the line reports
        ranks_counted = int(ones.item())
        assert ranks_counted == args.gpus, (ranks_counted, args.gpus)
    ev_ms, wall_ms, hip_events_ms, min_ms = t.tolist()
    ms_per_step = ev_ms / args.steps
    bytes_step = layer.bytes()
    value = world * bytes_step / (ms_per_step * 1e-3) / 1e9

    # cache-resident variant (one copy, served by L2 / Infinity Cache)
    hot = Layer(M, N, K, bits, g, dtype, device, 1, NF4_VALUES, seed=rank)
    hot.template_id = tid
    hot_ms, _ = time_graph(hot, args.steps, args.warmup, lambda: torch.cuda.synchronize(), cold=False)

    # Llama-3-70B MLP pair (configs[3]): with a process group it is the headline (strong scaling, one RCCL
    # all-reduce inside the captured graph); on one GPU it is an extra (TP = 1, no collective)
    try:
        tp_pair = (tp_mlp_pair(world, rank, device, dist, pairs=max(1, min(args.steps, 500)), warmup=min(args.warmup, 20))
                   if (dist is not None or not args.no_extras) else None)
    except Exception as exc:          # never lose the line to this leg  # noqa: BLE001
        tp_pair = {"error": f"{type(exc).__name__}: {exc}"[:300]}

    extras = []
    if rank == 0 and dist is None and not args.no_extras:
        for (n, k) in ((4096, 4096), (11008, 4096)):
            for m in (1, 16, 256, 1024, 2048, 4096):                      # >= 256: prefill, MFMA utilisation
                if (n, k, m) == (4096, 4096, 1):
                    continue
                lay = Layer(m, n, k, bits, g, dtype, device, copies_for(n, k, bits), NF4_VALUES)
                lay.tune()
                steps = 500 if m < 256 else (200 if m < 1024 else 60)
                # (best of two replays: one 200-step replay was once timed at twice its usual length on a fresh box)
                e_ms = min(time_graph(lay, steps, 20, lambda: torch.cuda.synchronize())[0] for _ in range(2))
                us = e_ms / steps * 1e3
                mm = {}
                if m >= 256:                                              # dense fp16 GEMM of the same shape beside it
                    wd = [torch.randn(k, n, device=device, dtype=dtype) for _ in range(max(2, L3_BYTES // (2 * k * n) + 2))]
                    xd = torch.randn(m, k, device=device, dtype=dtype)

                    class _MM:
                        def step(self, i, wd=wd, xd=xd):
                            return torch.mm(xd, wd[i % len(wd)])

                    d_ms, _ = time_graph(_MM(), steps, 10, lambda: torch.cuda.synchronize())
                    mm = {"torch_mm_fp16_us": round(d_ms / steps * 1e3, 3),
                          "speedup_vs_torch_mm": round(d_ms / e_ms, 3)}
                    del wd, xd
                extras.append({
                    **mm,
                    "workload": f"W4G64 fp16 M={m} K={k} N={n}", "template_id": lay.template_id,
                    "us": round(us, 3),
                    "GBps": round(lay.bytes() / us / 1e3, 1),
                    "TFLOPs": round(lay.flops() / us / 1e6, 2),
                    "frac_hbm_8TBps": round(lay.bytes() / us / 1e3 / HBM_PEAK_GBPS, 4),
                    "frac_mfma_2.5PF": round(lay.flops() / us / 1e6 / MFMA_PEAK_TFLOPS, 4),
                })
                del lay
                torch.cuda.empty_cache()
        # the other BASELINE.json configs, one line each (parity for them lives in tests/):
        # [2] W3G64 bf16 Llama-3-70B shapes, [3] the TP=8 column shard of 8192x28672 (what ONE of
        # eight GPUs runs; no collective for a column shard), [4] HIGGS pair codebook + Hadamard
        # pre-rotation on a Gemma-2-9B shape (one launch: the decode kernel rotates while staging X)
        bf16 = torch.bfloat16
        for (tag, m, n, k, b, dt, had) in (
                ("W3G64 bf16 M=1 K=8192 N=8192 (configs[2])", 1, 8192, 8192, 3, bf16, 0),
                ("W3G64 bf16 M=1 K=8192 N=28672 (configs[2])", 1, 28672, 8192, 3, bf16, 0),
                ("W4G64 fp16 M=1 K=8192 N=28672 full layer", 1, 28672, 8192, 4, dtype, 0),
                ("W4G64 fp16 M=2 K=8192 N=28672 (two rows: persistent one-shot kernel)", 2, 28672, 8192, 4, dtype, 0),
                ("W4G64 fp16 M=4 K=8192 N=28672 (persistent MFMA decode kernel)", 4, 28672, 8192, 4, dtype, 0),
                ("W4G64 fp16 M=4 K=28672 N=8192 (persistent MFMA decode kernel)", 4, 8192, 28672, 4, dtype, 0),
                ("W4G64 fp16 M=8 K=28672 N=8192 (persistent MFMA decode kernel)", 8, 8192, 28672, 4, dtype, 0),
                ("W4G64 fp16 M=4 K=14336 N=4096 (persistent MFMA decode kernel)", 4, 4096, 14336, 4, dtype, 0),
                ("W4G64 fp16 M=16 K=8192 N=28672 (persistent MFMA decode kernel)", 16, 28672, 8192, 4, dtype, 0),
                ("W4G64 fp16 M=8 K=4096 N=4096 (persistent MFMA decode kernel, activations resident in LDS)", 8, 4096, 4096, 4, dtype, 0),
                ("W2G64 fp16 M=4 K=4096 N=4096 (persistent MFMA decode kernel, 2-bit member)", 4, 4096, 4096, 2, dtype, 0),
                ("W2G64 fp16 M=16 K=11008 N=4096 (persistent MFMA decode kernel, 2-bit member)", 16, 4096, 11008, 2, dtype, 0),
                ("W4G64 fp16 M=64 K=8192 N=28672 (per-wave MFMA kernel, two slabs per wave)", 64, 28672, 8192, 4, dtype, 0),
                ("W4G64 fp16 M=1 K=8192 N=3584 = TP-8 column shard of 8192x28672 (configs[3])", 1, 3584, 8192, 4, dtype, 0),
                ("W4G64 fp16 M=1 K=3584 N=4096 pair codebook + hadamard_size=512 (configs[4], Gemma-2-9B)", 1, 4096, 3584, 4, dtype, 512)):
            lay = Layer(m, n, k, b, g, dt, device, copies_for(n, k, b), None, hadamard_size=had)
            lay.tune()
            e_ms, _ = time_graph(lay, 300, 20, lambda: torch.cuda.synchronize())
            us = e_ms / 300 * 1e3
            nbytes = algorithmic_bytes(m, n, k, b, g)
            extras.append({"workload": tag, "template_id": lay.template_id, "us": round(us, 3),
                           "GBps": round(nbytes / us / 1e3, 1),
                           "frac_hbm_8TBps": round(nbytes / us / 1e3 / HBM_PEAK_GBPS, 4)})
            del lay
            torch.cuda.empty_cache()

        # prefill of the 3- and 2-bit layers (block kernels qgemm_block3.h / qgemm_block2.h), torch.mm of the same
        # shape and dtype beside them
        for (tag, m, n, k, b, dt) in (
                ("W3G64 bf16 M=4096 K=4096 N=4096 prefill", 4096, 4096, 4096, 3, bf16),
                ("W3G64 bf16 M=1024 K=8192 N=28672 prefill (configs[2] layer)", 1024, 28672, 8192, 3, bf16),
                ("W3G64 bf16 M=256 K=8192 N=8192 (configs[2] layer; 128-row blocks x 4 K slices)", 256, 8192, 8192, 3, bf16),
                ("W3G64 bf16 M=1024 K=4096 N=4096 (128-row blocks x 2 K slices)", 1024, 4096, 4096, 3, bf16),
                ("W2G64 fp16 M=4096 K=4096 N=4096 prefill", 4096, 4096, 4096, 2, dtype)):
            lay = Layer(m, n, k, b, g, dt, device, copies_for(n, k, b), None)
            lay.tune()
            e_ms, _ = time_graph(lay, 30, 5, lambda: torch.cuda.synchronize())
            wd = [torch.randn(k, n, device=device, dtype=dt) for _ in range(max(2, L3_BYTES // (2 * k * n) + 2))]
            xd = torch.randn(m, k, device=device, dtype=dt)

            class _MM2:
                def step(self, i, wd=wd, xd=xd):
                    return torch.mm(xd, wd[i % len(wd)])

            d_ms, _ = time_graph(_MM2(), 30, 5, lambda: torch.cuda.synchronize())
            us = e_ms / 30 * 1e3
            extras.append({"workload": tag, "template_id": lay.template_id, "us": round(us, 3),
                           "TFLOPs": round(lay.flops() / us / 1e6, 2),
                           "frac_mfma_2.5PF": round(lay.flops() / us / 1e6 / MFMA_PEAK_TFLOPS, 4),
                           "torch_mm_us": round(d_ms / 30 * 1e3, 3), "speedup_vs_torch_mm": round(d_ms / e_ms, 3)})
            del lay, wd, xd
            torch.cuda.empty_cache()

        # the one speed figure the reference publishes for this path (intro-figure.jpg, README.md:135-137):
        # qgemm against torch.mm in fp16, W4G128, N = K = 8192, batch 1..32 (A100 ~2.05x, A6000 ~3.1x)
        n = k = 8192
        wcopies = [torch.randn(k, n, device=device, dtype=dtype) for _ in range(3)]      # 3 x 134 MB > L3

        class _Dense:
            def __init__(self, m):
                self.X = torch.randn(m, k, device=device, dtype=dtype)

            def step(self, i):
                return torch.mm(self.X, wcopies[i % 3])

        for m in (1, 16, 32):
            lay = Layer(m, n, k, 4, 128, dtype, device, copies_for(n, k, 4), None)
            lay.tune()
            q_ms, _ = time_graph(lay, 200, 10, lambda: torch.cuda.synchronize())
            d_ms, _ = time_graph(_Dense(m), 200, 10, lambda: torch.cuda.synchronize())
            extras.append({"workload": f"W4G128 fp16 M={m} K=8192 N=8192 vs torch.mm fp16 (reference's intro figure)",
                           "template_id": lay.template_id, "us": round(q_ms / 200 * 1e3, 3),
                           "torch_mm_us": round(d_ms / 200 * 1e3, 3), "speedup_vs_torch_mm": round(d_ms / q_ms, 2)})
            del lay
            torch.cuda.empty_cache()
        del wcopies
        torch.cuda.empty_cache()

    if rank == 0:
        import flute_amd
        achieved = bytes_step / (ms_per_step * 1e-3) / 1e9
        plan = flute_amd.utils.get_plan(M, N, K, bits, g, tid, layer.num_sms, dtype)
        # HBM bytes per launch from the committed PMC passes of THIS command (tools/prof_bench.sh writes the file
        # together with the plan it profiled): reported only while the plan is still the one that was profiled
        traffic, traffic_src, rocprof_us, rocprof_med_us = None, None, None, None
        import glob
        tpaths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_traffic.json")))     # newest round last
        if tpaths:
            tj = json.load(open(tpaths[-1]))
            if tj.get("plan") == plan:
                traffic, traffic_src = tj.get("hbm_bytes_per_launch"), "profiles/" + os.path.basename(tpaths[-1])
                rocprof_us, rocprof_med_us = tj.get("kernel_us_rocprof_avg"), tj.get("kernel_us_rocprof_median")
        # the floor of THIS launch shape: a pure read of the same packed bytes, requested the way the kernel requests them, in
        # the same kind of replayed graph (stand-alone harness, newest profiles/r*_calibration_stream_read.json)
        floor_us, floor_src = None, None
        cpaths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_calibration_stream_read.json")))
        if cpaths:
            reads = [r["us"] for r in json.load(open(cpaths[-1])).get("rows", [])
                     if str(r.get("variant", "")).startswith("read_") and r.get("N") == N and r.get("K") == K and "nt" not in r["variant"]]
            if reads:
                floor_us, floor_src = min(reads), "profiles/" + os.path.basename(cpaths[-1])
        replicas = {
            "metric": "qgemm effective GB/s, M=1, W4G64 NF4 fp16, K=N=4096 (Llama-3-8B linear), HBM-cold",
            "parallelism": f"{world} independent replica(s), no collective"}
        roofline = {
            "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS,
            "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
            "traffic_source": traffic_src,
            "frac_of_measured_copy_6.29TBps": round(achieved / HBM_COPY_GBPS, 4),
            "pure_read_floor_us": floor_us, "pure_read_floor_source": floor_src,
            "frac_of_pure_read_floor": None if floor_us is None else round(floor_us / (ms_per_step * 1e3), 4),
            "bytes_per_launch": bytes_step,
            "kernel_us": round(ms_per_step * 1e3, 3),
            "kernel_us_min_of_replays": round(min_ms / args.steps * 1e3, 3),
            "kernel_us_replays": [round(r / args.steps * 1e3, 3) for r in headline_timing.get("replays_ms", [])],
            "kernel_us_clock": headline_timing.get("clock"),
            "kernel_us_hip_events": round(hip_events_ms / args.steps * 1e3, 3),
            "kernel_us_rocprof": rocprof_us, "kernel_us_rocprof_median": rocprof_med_us,
            "kernel_us_rocprof_source": traffic_src,
            "achieved_from_rocprof_avg": None if not rocprof_us else round(bytes_step / rocprof_us / 1e3, 1),
            "note": "kernel_us = (second device-clock stamp - first) / steps: two one-lane stamp kernels captured in the graph "
                    "around the steps (period of back-to-back dependent launches, inter-kernel gaps included); "
                    "kernel_us_hip_events = HIP events around the same replay / steps: carries the fixed 16 - 19 us a graph launch "
                    "costs between its bracketing events whatever it holds, i.e. + 0.8 us per step at 20 steps, ~0 at 2000 "
                    "(profiles/r05/graph_replay_fixed_cost_probe.json); kernel_us_rocprof = average duration of the kernel in the "
                    "committed rocprofv3 --kernel-trace of this command (the tracer serialises replayed graph launches: an upper "
                    "bound, profiles/r03_rocprof/lab_traced_vs_untraced.txt); traffic = 2*FETCH_SIZE + WRITE_SIZE per launch "
                    "(gfx950 correction, MI355X_MICROARCH.md); a pure read of the same bytes, requested the same way, takes 2.8 us "
                    "per launch and an empty kernel 1.9 us in the stand-alone harness (profiles/r03_calibration_stream_read.json)",
        }
        # the second half of BASELINE.json's metric (TFLOP/s at M = 256) next to the roofline block: the extras entry of the same
        # shape, with the HBM traffic / MFMA-busy figures of the committed PMC passes while the plan is the one they were taken on
        m256 = None
        for e in extras:
            if e.get("workload") == "W4G64 fp16 M=256 K=4096 N=4096":
                m256 = {"workload": e["workload"], "us": e["us"], "TFLOPs": e["TFLOPs"], "bound": "mfma", "peak": MFMA_PEAK_TFLOPS,
                        "frac_mfma": e["frac_mfma_2.5PF"], "speedup_vs_torch_mm": e.get("speedup_vs_torch_mm"),
                        "template_id": e["template_id"], "algorithmic_bytes": algorithmic_bytes(256, 4096, 4096, bits, g),
                        "traffic": None, "mfma_busy_frac_chip": None, "pmc_source": None}
                mp = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_m256_pmc.json")))
                if mp:
                    mj = json.load(open(mp[-1]))
                    mplan = flute_amd.utils.get_plan(256, 4096, 4096, bits, g, e["template_id"], layer.num_sms, dtype)
                    if mj.get("plan") == mplan:
                        m256.update({"traffic": mj.get("hbm_bytes_per_launch"), "mfma_busy_frac_chip": mj.get("mfma_busy_frac_chip"),
                                     "kernel_us_rocprof_median": mj.get("kernel_us_rocprof_median"),
                                     "pmc_source": "profiles/" + os.path.basename(mp[-1])})
        out = {
            "metric": replicas["metric"],
            "value": round(value, 2), "unit": "GB/s", "n_gpus": world, "ranks_counted_over_rccl": ranks_counted,
            "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 6),
            "clock": headline_timing.get("clock"),            # what `value` was timed with; the HIP-event figure of the same replay follows
            "value_hip_events": round(world * bytes_step / (hip_events_ms / args.steps * 1e-3) / 1e9, 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16",
            "data": "synthetic (random codes/scales, NF4 table, X=randn/100; "
                    f"{len(layer.Q)} rotating weight copies > 256 MiB L3; caches flushed (untimed) before the timed replay)",
            "working_set_bytes": min(args.steps, len(layer.Q)) * (2 * (bits * N // 16) * K + 2 * N * K // g),
            "timed_replay_cold": "untimed 512 MiB read between the warm replays and the timed replay: every "
                                 "timed step streams its weights from HBM for any --steps",
            "config": {"workload": "W4G64 NF4 fp16 qgemm, M=1, K=4096, N=4096 (BASELINE configs[1])",
                       "template_id": tid, "plan": plan,
                       "launch": "hipGraph replay of all steps, timed by two device-clock stamps captured around them",
                       "parallelism": replicas["parallelism"]},
            "roofline": roofline,
            "m256": m256,
            "cache_resident": {"us": round(hot_ms / args.steps * 1e3, 3),
                               "GBps": round(bytes_step / (hot_ms / args.steps * 1e-3) / 1e9, 1)},
            "eager_us_per_step": round(eager_ms / min(args.steps, 500) * 1e3, 3),
            "wall_ms_timed_region": round(wall_ms, 3),
            "extras": extras,
        }
        # The headline metric is the SAME for every N (the driver computes scaling efficiency from the per-N values):
        # N independent replicas of the single-GPU workload, no collective, weak scaling.  The tensor-parallel
        # workload of BASELINE configs[3] - the column-parallel / row-parallel MLP pair with its ONE RCCL
        # all-reduce, captured in one hipGraph - is reported under its own key `tp_mlp_pair` (strong scaling).
        if tp_pair is not None:
            out["tp_mlp_pair"] = tp_pair
        if dist is None and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(N, K, bits, g)
        if saved_stdout is not None:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
        print(json.dumps(out), flush=True)
        if saved_stdout is not None:
            os.dup2(2, 1)                      # teardown chatter goes to stderr again
    if dist is not None:
        dist.barrier(device_ids=[local_rank])
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
