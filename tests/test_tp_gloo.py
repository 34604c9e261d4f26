"""gloo tests of the tensor-parallel sharding (CPU, world 2 and world 8).  Per-rank compute is
the oracle (the worker process replaces `tp.local_qgemm`, the one function that launches the
HIP kernel); what is under test is that shards cut out of the packed matrix are valid packed
matrices, that column-parallel (no collective) and row-parallel (one all-reduce) reproduce the
unsharded result, and how far eight T-rounded partial sums drift from it."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _oracle_fn(tile_p):
    from oracle import flute_oracle as O

    def fn(x, Q, S, table, table2, bits, g, tid):
        return O.qgemm(x, Q.numpy(), S, table, table2, bits, g, tile_p)
    return fn


def _worker(rank, world, port, bits, tile_p, g, result):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import flute_oracle as O
        from flute_amd import tp
        torch.manual_seed(0)
        blk = tp.columns_per_block(bits, tile_p)
        K, N, M, dtype = 512, 2 * world * blk, 3, torch.float32
        # fp32 "T" keeps gloo reductions exact enough to compare tightly; the layout
        # logic under test is dtype independent
        W = torch.randint(0, 2 ** bits, (K, N), dtype=torch.uint8)
        Q = torch.from_numpy(O.pack(W.numpy(), bits, tile_p))
        S16 = torch.randn(N, K // g).half()
        table16 = torch.randn(2 ** bits).half()
        table2 = O.make_qmap2_from_qmap(table16)
        X = (torch.randn(M, K) / 10).half()
        full = O.qgemm(X, Q.numpy(), S16, table16, table2, bits, g, tile_p).float()

        fn = _oracle_fn(tile_p)
        tp.local_qgemm = fn                      # this process only: the oracle stands in for the HIP kernel
        col = tp.ColumnParallelQLinear.from_full(Q, S16, table16, table2, bits, g, 0, tile_p,
                                                 gather_output=True)
        # the shard is itself a valid packed matrix of the right shape
        assert col.weight.shape == (bits * (N // world) // 16, K)
        Wshard = O.unpack(col.weight.numpy(), bits, tile_p)
        assert np.array_equal(Wshard, W.numpy()[:, rank * N // world:(rank + 1) * N // world])
        y_col = col(X).float()
        assert torch.equal(y_col, full), "column-parallel + all_gather"

        tp.local_qgemm = lambda *a: fn(*a).float()
        row = tp.RowParallelQLinear.from_full(Q, S16, table16, table2, bits, g, 0)
        k0, k1 = rank * K // world, (rank + 1) * K // world
        y_row = row(X[:, k0:k1].contiguous())
        err = ((y_row - full).norm() / full.norm()).item()
        assert err < 2e-3, err
        result[rank] = 1
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("bits,tile_p", [(4, 32), (4, 64), (2, 32), (3, 32)])
def test_tp_sharding_world2(bits, tile_p):
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    result = ctx.Manager().dict()
    procs = [ctx.Process(target=_worker, args=(r, world, port, bits, tile_p, 64, result))
             for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert dict(result) == {0: 1, 1: 1}


def _worker8(rank, world, port, result):
    """Row-parallel at TP = 8 the way the product runs it: every rank's partial is rounded to fp16 (the kernel's
    output type) BEFORE the all-reduce, which then sums eight fp16 addends (vllm_utils.py:265-326 contract)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import flute_oracle as O
        from flute_amd import tp
        torch.manual_seed(0)
        bits, tile_p, g = 4, 32, 64
        K, N, M = 8 * 512, 256, 4
        W = torch.randint(0, 2 ** bits, (K, N), dtype=torch.uint8)
        Q = torch.from_numpy(O.pack(W.numpy(), bits, tile_p))
        S16 = torch.randn(N, K // g).half()
        table16 = torch.randn(2 ** bits).half()
        table2 = O.make_qmap2_from_qmap(table16)
        X = (torch.randn(M, K) / 100).half()
        What = (table16[W.long()] * torch.repeat_interleave(S16, g, dim=1).T).float()
        full = X.float() @ What                               # fp32 ground truth (tests/kernel.py:68-71)
        tp.local_qgemm = lambda x, Qs, Ss, t, t2, b, gs, tid: O.qgemm(x, Qs.numpy(), Ss, t, t2, b, gs, tile_p).half()
        row = tp.RowParallelQLinear.from_full(Q, S16, table16, table2, bits, g, 0)
        assert row.weight.shape == (bits * N // 16, K // world)
        k0, k1 = rank * K // world, (rank + 1) * K // world
        y = row(X[:, k0:k1].contiguous())                    # fp16 partial -> all_reduce(sum) in fp16
        assert y.dtype == torch.float16
        err = ((y.float() - full).norm() / full.norm()).item()
        # eight fp16-rounded addends: stays inside the reference's fp16 acceptance (2.0e-3, tests/kernel.py:12) and
        # BASELINE's 1e-3 budget; measured here ~3e-4
        assert err < 1e-3, err
        result[rank] = err
    finally:
        dist.destroy_process_group()


def test_tp_row_parallel_world8_fp16_partials():
    world = 8
    port = _free_port()
    ctx = mp.get_context("spawn")
    result = ctx.Manager().dict()
    procs = [ctx.Process(target=_worker8, args=(r, world, port, result)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert len(result) == world and max(result.values()) < 1e-3


def _worker8_mlp(rank, world, port, result):
    """The MLP pair at TP = 8, end to end: a column-parallel layer (no collective, output stays sharded) feeds a
    row-parallel layer whose K shard is exactly that output shard; ONE all-reduce for the pair
    (flute/integrations/vllm_utils.py:224-226, 265-326)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import flute_oracle as O
        from flute_amd import tp
        torch.manual_seed(0)
        bits, tile_p, g = 4, 32, 64
        blk = tp.columns_per_block(bits, tile_p)              # 128 columns
        H, I, M = 512, world * blk, 3                         # hidden 512, intermediate 1024: 128 columns / 128 k per rank
        W1 = torch.randint(0, 2 ** bits, (H, I), dtype=torch.uint8)
        W2 = torch.randint(0, 2 ** bits, (I, H), dtype=torch.uint8)
        Q1 = torch.from_numpy(O.pack(W1.numpy(), bits, tile_p))
        Q2 = torch.from_numpy(O.pack(W2.numpy(), bits, tile_p))
        S1 = (torch.randn(I, H // g) / 4).half()
        S2 = (torch.randn(H, I // g) / 4).half()
        table16 = torch.randn(2 ** bits).half()
        table2 = O.make_qmap2_from_qmap(table16)
        X = (torch.randn(M, H) / 10).half()
        # unsharded pair through the oracle (fp16 hand-off between the layers, as the product)
        h_full = O.qgemm(X, Q1.numpy(), S1, table16, table2, bits, g, tile_p).half()
        y_full = O.qgemm(h_full, Q2.numpy(), S2, table16, table2, bits, g, tile_p).float()

        calls = {"all_reduce": 0, "all_gather": 0}
        real_ar, real_ag = dist.all_reduce, dist.all_gather
        dist.all_reduce = lambda *a, **k: (calls.__setitem__("all_reduce", calls["all_reduce"] + 1), real_ar(*a, **k))[1]
        dist.all_gather = lambda *a, **k: (calls.__setitem__("all_gather", calls["all_gather"] + 1), real_ag(*a, **k))[1]
        tp.local_qgemm = lambda x, Qs, Ss, t, t2, b, gs, tid: O.qgemm(x, Qs.numpy(), Ss, t, t2, b, gs, tile_p).half()
        up = tp.ColumnParallelQLinear.from_full(Q1, S1, table16, table2, bits, g, 0, tile_p, gather_output=False)
        down = tp.RowParallelQLinear.from_full(Q2, S2, table16, table2, bits, g, 0)
        h = up(X)                                             # [M, I / world], no collective
        assert h.shape == (M, I // world) and calls == {"all_reduce": 0, "all_gather": 0}
        assert torch.equal(h, h_full[:, rank * I // world:(rank + 1) * I // world]), "the column shard IS the row layer's K shard"
        y = down(h)                                           # fp16 partials, one all-reduce
        assert calls == {"all_reduce": 1, "all_gather": 0}
        err = ((y.float() - y_full).norm() / y_full.norm()).item()
        assert err < 1e-3, err
        result[rank] = err
    finally:
        dist.destroy_process_group()


def test_tp_mlp_pair_world8_one_allreduce():
    world = 8
    port = _free_port()
    ctx = mp.get_context("spawn")
    result = ctx.Manager().dict()
    procs = [ctx.Process(target=_worker8_mlp, args=(r, world, port, result)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert len(result) == world and max(result.values()) < 1e-3


def test_shard_bounds_rejected():
    from flute_amd import tp
    Q = torch.zeros((4 * 128 // 16, 128), dtype=torch.int16)
    S = torch.zeros((128, 2))
    with pytest.raises(ValueError):
        tp.shard_columns(Q, S, 4, 32, 2, 0)        # 128 columns = one block, cannot split
    with pytest.raises(ValueError):
        tp.shard_rows(Q, S, 64, 4, 0)              # 128 / 4 = 32 < 64
