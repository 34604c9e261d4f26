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


def _install_vllm_stub():
    """The four names flute_amd.integrations.vllm_utils takes from vLLM, with vLLM's semantics for them."""
    import sys
    import types

    def set_weight_attrs(weight, attrs):
        for k, v in (attrs or {}).items():
            setattr(weight, k, v)

    class LinearBase(torch.nn.Module):
        pass

    class LinearMethodBase:
        pass

    class QuantizationConfig:
        def __init__(self):
            pass

        @staticmethod
        def get_from_keys(config, keys):
            for k in keys:
                if k in config:
                    return config[k]
            raise ValueError(keys)

        @staticmethod
        def get_from_keys_or(config, keys, default):
            try:
                return QuantizationConfig.get_from_keys(config, keys)
            except ValueError:
                return default

    names = ["vllm", "vllm.model_executor", "vllm.model_executor.layers", "vllm.model_executor.layers.linear",
             "vllm.model_executor.layers.quantization", "vllm.model_executor.layers.quantization.base_config"]
    mods = {n: types.ModuleType(n) for n in names}
    mods["vllm.model_executor.layers.linear"].LinearBase = LinearBase
    mods["vllm.model_executor.layers.linear"].LinearMethodBase = LinearMethodBase
    mods["vllm.model_executor.layers.linear"].set_weight_attrs = set_weight_attrs
    mods["vllm.model_executor.layers.quantization.base_config"].QuantizationConfig = QuantizationConfig
    saved = {n: sys.modules.get(n) for n in names}
    sys.modules.update(mods)
    return saved, LinearBase


@pytest.mark.parametrize("bits,dtype", [(4, torch.float16), (2, torch.bfloat16)])
def test_flute_linear_method_runs_against_a_vllm_stub(bits, dtype):
    """FluteConfig / FluteLinearMethod (flute/integrations/vllm_utils.py:42-349) executed end to end: create_weights
    registers every tensor a FLUTE checkpoint holds (incl. `tables2`), a vLLM-style `params_dict[name]` loop loads a
    FluteLinear state dict packed for ANOTHER TileP, process_weights_after_loading re-tunes / repacks for this GPU and
    apply() launches the HIP kernel."""
    import importlib
    import sys
    import flute_amd
    from flute_amd import utils
    from flute_amd.integrations import vllm_utils as V0
    from oracle import flute_oracle as O
    saved, LinearBase = _install_vllm_stub()
    try:
        V = importlib.reload(V0)
        assert V._HAVE_VLLM
        d = torch.device("cuda:0")
        torch.manual_seed(10 + bits)
        K, g = 1024, 64
        parts = [512, 1024]                                     # a fused gate / up projection
        N = sum(parts)
        tile_p = V.reference_packed_tile_p()
        table = torch.randn(2 ** bits).to(dtype)
        W = torch.randint(0, 2 ** bits, (K, N), dtype=torch.uint8)
        S = torch.randn(N, K // g).to(dtype)
        # the checkpoint: every partition packed on its own (what prepare_model_flute writes per projection)
        ckpt = {}
        n0 = 0
        Qs, Ss = [], []
        for n in parts:
            Qs.append(torch.from_numpy(O.pack(W[:, n0:n0 + n].numpy(), bits, tile_p)))
            Ss.append(S[n0:n0 + n])
            n0 += n
        ckpt["proj.weight"] = torch.cat(Qs)
        ckpt["proj.scales"] = torch.cat(Ss)
        ckpt["proj.tables"] = table
        ckpt["proj.tables2"] = O.make_qmap2_from_qmap(table)

        cfg = V.FluteConfig.from_config({"num_bits": bits, "group_size": g, "num_sms": 108})
        layer = LinearBase()
        method = cfg.get_quant_method(layer, "proj")
        assert isinstance(method, V.FluteLinearMethod)
        method.create_weights(layer, K, parts, K, N, dtype)
        params_dict = {"proj." + n: p for n, p in layer.named_parameters()}
        assert set(params_dict) == set(ckpt), "every checkpoint tensor needs a parameter (vLLM: params_dict[name])"
        for name, tensor in ckpt.items():                      # vLLM's load_weights loop with the default loader
            param = params_dict[name]
            assert param.shape == tensor.shape and param.dtype == tensor.dtype, name
            param.data.copy_(tensor)
        method.process_weights_after_loading(layer)
        assert layer.flute_template_id is not None and not layer.needs_repacking
        X = (torch.randn(5, K) / 10).to(dtype).to(d)
        bias = torch.randn(N).to(dtype).to(d)
        got = method.apply(layer, X, bias).float().cpu()
        What = table.float()[W.long()] * torch.repeat_interleave(S.float(), g, dim=1).T
        want = X.float().cpu() @ What + bias.float().cpu()
        tol = 2e-3 if dtype == torch.float16 else 1.5e-2
        assert ((got - want).norm() / want.norm()).item() < tol
        # a second process_weights_after_loading is a no-op
        q_before = layer.weight.data.clone()
        method.process_weights_after_loading(layer)
        assert torch.equal(q_before, layer.weight.data)
    finally:
        for n, m in saved.items():
            if m is None:
                sys.modules.pop(n, None)
            else:
                sys.modules[n] = m
        importlib.reload(V0)
