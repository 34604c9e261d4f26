"""`flute_amd.integrations.vllm_utils.repack_loaded_shard` on tensors cut the way vLLM's weight loader cuts them
(flute/integrations/vllm_utils.py:228-326 is the reference's gather / unpack / re-shard / repack version): fused
partitions stacked along the packed dimension, column-parallel ranks holding a row slice of every partition,
row-parallel ranks a K slice.  vLLM itself is not installed; the classes that need it are import-guarded."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("bits,dtype", [(4, torch.float16), (2, torch.bfloat16), (3, torch.float16)])
def test_repack_of_loader_shards(bits, dtype):
    import flute_amd
    from flute_amd import utils
    from flute_amd.integrations import vllm_utils as V
    from oracle import flute_oracle as O
    d = torch.device("cuda:0")
    torch.manual_seed(bits)
    K, g, world, tile_p = 1024, 64, 2, 32
    parts = (1024, 512) if bits != 3 else (1024, 1024)          # fused partitions (e.g. gate / up)
    table = torch.randn(2 ** bits).to(dtype)
    table2 = O.make_qmap2_from_qmap(table)
    Ws = [torch.randint(0, 2 ** bits, (K, n), dtype=torch.uint8) for n in parts]
    Ss = [torch.randn(n, K // g).to(dtype) for n in parts]
    Qs = [torch.from_numpy(O.pack(W.numpy(), bits, tile_p)) for W in Ws]
    What = [table.float()[W.long()] * torch.repeat_interleave(S.float(), g, dim=1).T for W, S in zip(Ws, Ss)]
    X = (torch.randn(3, K) / 10).to(dtype)
    ws, num_sms = utils.get_workspace_streamk(d), utils.get_device_num_sms(d)
    tol = 2e-3 if dtype == torch.float16 else 1.5e-2

    def run(Q, S, x, tid):
        return flute_amd.qgemm(x.to(d), Q, S.to(d), table.to(d), table2.to(d), ws, bits, g, tid, num_sms).float().cpu()

    # unsharded fused layer (MergedColumnParallelLinear at TP = 1)
    Q, tid = V.repack_loaded_shard(torch.cat(Qs).to(d), torch.cat(Ss).to(d), parts, bits, g)
    want = X.float() @ torch.cat(What, dim=1)
    assert ((run(Q, torch.cat(Ss), X, tid) - want).norm() / want.norm()).item() < tol
    if bits == 3:
        return                                                  # bit planes: no contiguous row slice per rank
    for rank in range(world):
        # column-parallel: the loader keeps rows [rank * P_i / world, ...) of every partition's packed matrix
        q = torch.cat([Qi[rank * Qi.shape[0] // world:(rank + 1) * Qi.shape[0] // world] for Qi in Qs]).to(d)
        s = torch.cat([Si[rank * Si.shape[0] // world:(rank + 1) * Si.shape[0] // world] for Si in Ss])
        Q, tid = V.repack_loaded_shard(q, s.to(d), [n // world for n in parts], bits, g)
        want = X.float() @ torch.cat([Wi[:, rank * Wi.shape[1] // world:(rank + 1) * Wi.shape[1] // world] for Wi in What], dim=1)
        assert ((run(Q, s, X, tid) - want).norm() / want.norm()).item() < tol, ("column", rank)
        # row-parallel: columns [rank * K / world, ...) of the packed matrix, groups likewise
        k0, k1 = rank * K // world, (rank + 1) * K // world
        Q, tid = V.repack_loaded_shard(Qs[0][:, k0:k1].contiguous().to(d), Ss[0][:, k0 // g:k1 // g].contiguous().to(d),
                                       [parts[0]], bits, g)
        want = X[:, k0:k1].float() @ What[0][k0:k1]
        got = run(Q, Ss[0][:, k0 // g:k1 // g].contiguous(), X[:, k0:k1].contiguous(), tid)
        assert ((got - want).norm() / want.norm()).item() < tol, ("row", rank)
