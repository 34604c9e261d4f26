"""RCCL (`nccl` backend) test of the tensor-parallel layers on real GPUs: each rank runs the HIP qgemm on ITS
shard cut out of the packed matrix, column-parallel with all_gather and row-parallel with ONE all-reduce, and
the result is compared with the unsharded HIP launch and with the oracle (flute_amd/tp.py; the contract of
flute/integrations/vllm_utils.py:224-226, 265-326).  The world-2 case needs two GPUs (the driver's multi-GPU
node); the world-1 case runs the same code over a one-rank RCCL group on the single-GPU box, including the
all-reduce captured in a hipGraph the way bench.py --gpus N replays it."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, bits, tile_p, g, dtype_name, result):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    d = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=d)
    try:
        import flute_amd
        from flute_amd import tp, utils
        from oracle import flute_oracle as O
        dtype = getattr(torch, dtype_name)
        torch.manual_seed(0)                                   # every rank builds the same full layer
        blk = tp.columns_per_block(bits, tile_p)
        K, N, M = 2048, 4 * world * blk, 3
        tid = min(t for (b, t), c in flute_amd.TEMPLATE_CONFIGS.items() if b == bits and c["TileP"] == tile_p)
        num_sms = utils.get_device_num_sms(d)
        ws = utils.get_workspace_streamk(d)
        W = torch.randint(0, 2 ** bits, (K, N), dtype=torch.uint8)
        S = torch.randn(N, K // g).to(dtype)
        table = torch.randn(2 ** bits).to(dtype)
        table2 = O.make_qmap2_from_qmap(table)
        X = (torch.randn(M, K) / 10).to(dtype)
        Q = torch.from_numpy(O.pack(W.numpy(), bits, tile_p))
        want = O.qgemm(X, Q.numpy(), S, table, table2, bits, g, tile_p).float()
        Qd, Sd, td, t2d, Xd = (t.to(d) for t in (Q, S, table, table2, X))
        full = flute_amd.qgemm(Xd, Qd, Sd, td, t2d, ws, bits, g, tid, num_sms)

        col = tp.ColumnParallelQLinear.from_full(Qd, Sd, td, t2d, bits, g, tid, tile_p, gather_output=True)
        assert col.weight.shape == (bits * (N // world) // 16, K)
        y_col = col(Xd)
        # a column shard computes exactly the columns of the unsharded launch that it owns... up to the plan
        # (split-K / k-wave order may differ between the shard's plan and the full layer's): oracle tolerance
        tol = 2e-3 if dtype == torch.float16 else 1.5e-2
        err = ((y_col.float().cpu() - want).norm() / want.norm()).item()
        assert err < tol, ("column-parallel", err)
        err = ((y_col.float() - full.float()).norm() / full.float().norm()).item()
        assert err < tol, ("column-parallel vs unsharded", err)

        row = tp.RowParallelQLinear.from_full(Qd, Sd, td, t2d, bits, g, tid)
        k0, k1 = rank * K // world, (rank + 1) * K // world
        xs = Xd[:, k0:k1].contiguous()
        y_row = row(xs)
        err = ((y_row.float().cpu() - want).norm() / want.norm()).item()
        assert err < tol, ("row-parallel", err)

        # the pair the bench replays: column shard -> row shard -> all-reduce, captured in ONE hipGraph
        torch.manual_seed(1 + rank)
        F = N // world                                          # the column shard's width = the row shard's K
        if F % max(64, g) == 0:
            W2 = torch.randint(0, 2 ** bits, (F, N), dtype=torch.uint8)
            Q2 = torch.from_numpy(O.pack(W2.numpy(), bits, tile_p)).to(d)
            S2 = torch.randn(N, F // g).to(dtype).to(d)
            col_s = tp.ColumnParallelQLinear.from_full(Qd, Sd, td, t2d, bits, g, tid, tile_p)
            down = tp.RowParallelQLinear(Q2, S2, td, t2d, bits, g, tid)
            eager = down(col_s(Xd))
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):      # (the watchdog thread stays legal)
                out = down(col_s(Xd))
            graph.replay()
            torch.cuda.synchronize()
            assert torch.equal(out, eager), "graph replay of kernels + all-reduce differs from eager"
        result[rank] = 1
    finally:
        dist.destroy_process_group()


def _run(world, bits, tile_p, dtype_name):
    port = _free_port()
    ctx = mp.get_context("spawn")
    result = ctx.Manager().dict()
    procs = [ctx.Process(target=_worker, args=(r, world, port, bits, tile_p, 64, dtype_name, result))
             for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
    for p in procs:
        if p.is_alive():
            p.kill()                                            # the exact processes started above
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert dict(result) == {r: 1 for r in range(world)}


@pytest.mark.parametrize("bits,tile_p,dtype_name", [(4, 32, "float16"), (3, 32, "bfloat16")])
def test_tp_rccl_world1(bits, tile_p, dtype_name):
    _run(1, bits, tile_p, dtype_name)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL over xGMI)")
@pytest.mark.parametrize("bits,tile_p,dtype_name", [(4, 32, "float16"), (4, 64, "bfloat16"), (2, 32, "float16"),
                                                    (3, 32, "bfloat16")])
def test_tp_rccl_world2(bits, tile_p, dtype_name):
    _run(2, bits, tile_p, dtype_name)


def test_simulated_tp8_on_one_gpu():
    """BASELINE configs[3] without eight GPUs: all 8 column shards of the Llama-3-70B up projection (8192 x 28672)
    and all 8 row shards of the down projection (28672 x 8192) are cut out of the packed matrices with
    tp.shard_columns / tp.shard_rows and run through the HIP kernels one after the other on this GPU; the column
    outputs are concatenated, the row partials are summed IN T in rank order (what the all-reduce does to the
    kernels' fp16 outputs), and both are compared with the unsharded HIP launch and with the fp32 product of the
    dequantised matrix (tests/kernel.py:68-71).  Contract: flute/integrations/vllm_utils.py:265-326.
    Tolerances: rel-Frobenius < 1e-3 (fp16) against the fp32 product - BASELINE's bar - for the column-parallel
    result and the unsharded launches; < 1.5e-3 for the sum of eight fp16-rounded partials (the reference accepts
    2.0e-3, tests/kernel.py:12)."""
    import flute_amd
    from flute_amd import tp, utils
    d = torch.device("cuda", 0)
    torch.manual_seed(0)
    bits, tile_p, g, dtype, world, M = 4, 32, 64, torch.float16, 8, 2
    H, F = 8192, 28672
    tid = min(t for (b, t), c in flute_amd.TEMPLATE_CONFIGS.items() if b == bits and c["TileP"] == tile_p)
    num_sms = utils.get_device_num_sms(d)
    ws = utils.get_workspace_streamk(d)
    table = torch.randn(2 ** bits, device=d).to(dtype)
    table2 = utils.make_qmap2_from_qmap(table)

    def layer(K, N):
        W = torch.randint(0, 2 ** bits, (K, N), dtype=torch.uint8, device=d)
        Q = utils.pack(W, bits, [tid], num_sms)
        S = (torch.randn(N, K // g, device=d) * 0.1).to(dtype)
        What = table.float()[W.long()] * torch.repeat_interleave(S.float(), g, dim=1).T      # [K, N] fp32
        return W, Q, S, What

    def relerr(a, b):
        return ((a.float() - b.float()).norm() / b.float().norm()).item()

    # ---- column-parallel up projection: N-shards, no collective ----
    X = (torch.randn(M, H, device=d) / 10).to(dtype)
    W, Q, S, What = layer(H, F)
    want = X.float() @ What
    full = flute_amd.qgemm(X, Q, S, table, table2, ws, bits, g, tid, num_sms)
    parts = []
    for r in range(world):
        q, s = tp.shard_columns(Q, S, bits, tile_p, world, r)
        assert q.shape == (bits * (F // world) // 16, H) and s.shape == (F // world, H // g)
        # the shard is a valid packed matrix of exactly this rank's columns
        codes = utils.unpack_codes(q, bits, tid)
        assert torch.equal(codes, W[:, r * F // world:(r + 1) * F // world])
        parts.append(flute_amd.qgemm(X, q, s, table, table2, ws, bits, g, tid, num_sms))
    y_col = torch.cat(parts, dim=-1)
    assert relerr(full, want) < 1e-3 and relerr(y_col, want) < 1e-3
    assert relerr(y_col, full) < 1e-3
    del W, Q, S, What, parts

    # ---- row-parallel down projection: K-shards, partials summed in T (the all-reduce) ----
    X2 = (torch.randn(M, F, device=d) / 10).to(dtype)
    W, Q, S, What = layer(F, H)
    want = X2.float() @ What
    full = flute_amd.qgemm(X2, Q, S, table, table2, ws, bits, g, tid, num_sms)
    acc = None
    for r in range(world):
        q, s = tp.shard_rows(Q, S, g, world, r)
        assert q.shape == (bits * H // 16, F // world) and s.shape == (H, F // world // g)
        part = flute_amd.qgemm(X2[:, r * F // world:(r + 1) * F // world].contiguous(), q, s, table, table2, ws, bits, g,
                               tid, num_sms)
        assert part.dtype == dtype
        acc = part if acc is None else acc + part              # fp16 + fp16 -> fp16, as RCCL sums them
    assert relerr(full, want) < 1e-3
    assert relerr(acc, want) < 1.5e-3, relerr(acc, want)
    assert relerr(acc, full) < 1.5e-3
