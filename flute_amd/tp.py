"""Tensor-parallel use of a packed layer.

The path shards by output columns with NO collective (column-parallel: each GPU
holds Q[P/tp, K], S[N/tp, G] and computes D[:, N/tp]) and by input rows with ONE
all-reduce (row-parallel: Q[P, K/tp], S[N, G/tp], partial D summed over ranks) -
the Megatron / vLLM pairing the reference plugs into
(flute/integrations/vllm_utils.py:224-226, 265-326).

Unlike the reference, which gathers -> unpacks -> re-shards -> repacks at load
(vllm_utils.py:228-326, an all-gather of int16 cast to int32), shards are cut
directly out of the packed matrix: the wire format interleaves columns only
inside blocks of J*TileP columns and pairs of k, so any shard boundary that is a
multiple of that block (N) or of lcm(64, group_size) (K) is a plain slice.

Collectives go through torch.distributed (backend "nccl" == RCCL over xGMI on
ROCm; "gloo" in the CPU tests).  The decode-size message (M*N*2 B = 16 KB at
M=1, N=8192) is latency-bound: one all-reduce per row-parallel layer, never more.
"""
from typing import Tuple

import torch
import torch.distributed as dist


def columns_per_block(num_bits: int, tile_p: int) -> int:
    return tile_p * (16 if num_bits == 3 else 16 // num_bits)


def shard_columns(Q: torch.Tensor, S: torch.Tensor, num_bits: int, tile_p: int, world: int,
                  rank: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """N-shard `rank` of (Q[P,K], S[N,G]) -> (Q[P/world,K], S[N/world,G])."""
    P, K = Q.shape
    N = S.shape[0]
    blk = columns_per_block(num_bits, tile_p)
    if N % (world * blk):
        raise ValueError(f"N={N} cannot be split {world}-way on {blk}-column blocks")
    n0, n1 = rank * N // world, (rank + 1) * N // world
    if num_bits in (2, 4):
        J = 16 // num_bits
        q = Q[n0 // J: n1 // J]
    else:
        # plane 0 rows [N/16), then per 512-column block 32 rows of plane 1 and 32 of plane 2
        P1 = N // 16
        q = torch.cat([Q[n0 // 16: n1 // 16], Q[P1 + n0 // 8: P1 + n1 // 8]], dim=0)
    return q.contiguous(), S[n0:n1].contiguous()


def shard_rows(Q: torch.Tensor, S: torch.Tensor, group_size: int, world: int, rank: int
               ) -> Tuple[torch.Tensor, torch.Tensor]:
    """K-shard `rank` of (Q[P,K], S[N,G]) -> (Q[P,K/world], S[N,G/world])."""
    P, K = Q.shape
    step = max(64, group_size)
    if K % (world * step):
        raise ValueError(f"K={K} cannot be split {world}-way on multiples of {step}")
    k0, k1 = rank * K // world, (rank + 1) * K // world
    return Q[:, k0:k1].contiguous(), S[:, k0 // group_size: k1 // group_size].contiguous()


def local_qgemm(x, Q, S, table, table2, num_bits, group_size, template_id):
    """The per-rank product: flute.qgemm on this rank's shard (the HIP kernel; there is no CPU path)."""
    import flute_amd
    from flute_amd import utils
    return flute_amd.qgemm(x, Q, S, table, table2, utils.get_workspace_streamk(x.device), num_bits,
                           group_size, template_id, utils.get_device_num_sms(x.device))


class _ParallelQLinear(torch.nn.Module):
    def __init__(self, Q, S, table, table2, num_bits, group_size, template_id, group=None):
        super().__init__()
        self.register_buffer("weight", Q)
        self.register_buffer("scales", S)
        self.register_buffer("tables", table)
        self.register_buffer("tables2", table2)
        self.num_bits, self.group_size, self.template_id = num_bits, group_size, template_id
        self.group = group

    def _local(self, x):
        return local_qgemm(x, self.weight, self.scales, self.tables, self.tables2,
                           self.num_bits, self.group_size, self.template_id)


class ColumnParallelQLinear(_ParallelQLinear):
    """Holds the N-shard; output stays sharded (gather_output=False) or is
    all-gathered along the last dim."""

    @classmethod
    def from_full(cls, Q, S, table, table2, num_bits, group_size, template_id, tile_p,
                  gather_output=False, group=None):
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        q, s = shard_columns(Q, S, num_bits, tile_p, world, rank)
        m = cls(q, s, table, table2, num_bits, group_size, template_id, group)
        m.gather_output = gather_output
        return m

    def forward(self, x):
        y = self._local(x)
        if not getattr(self, "gather_output", False):
            return y
        world = dist.get_world_size(self.group)
        parts = [torch.empty_like(y) for _ in range(world)]
        dist.all_gather(parts, y.contiguous(), group=self.group)
        return torch.cat(parts, dim=-1)


class RowParallelQLinear(_ParallelQLinear):
    """Holds the K-shard; input is the matching K-slice (e.g. the sharded output
    of a column-parallel layer); ONE all-reduce(sum) of M*N*2 bytes."""

    @classmethod
    def from_full(cls, Q, S, table, table2, num_bits, group_size, template_id, group=None):
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        q, s = shard_rows(Q, S, group_size, world, rank)
        return cls(q, s, table, table2, num_bits, group_size, template_id, group)

    def forward(self, x_shard):
        y = self._local(x_shard)
        dist.all_reduce(y, op=dist.ReduceOp.SUM, group=self.group)
        return y
