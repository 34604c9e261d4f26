"""Load-time path of a FLUTE checkpoint (flute_amd/integrations/huggingface.py; reference
flute/integrations/huggingface.py:84-236): module replacement and the transformers registration on CPU,
the repack of a foreign-GPU checkpoint on the GPU."""
import os

import flute_amd
import pytest
import torch

from flute_amd.integrations import huggingface as hf
from flute_amd.integrations.base import FluteLinear


class _Block(torch.nn.Module):
    def __init__(self, d, f, dtype):
        super().__init__()
        self.up_proj = torch.nn.Linear(d, f, bias=False, dtype=dtype)
        self.down_proj = torch.nn.Linear(f, d, bias=True, dtype=dtype)

    def forward(self, x):
        return self.down_proj(torch.nn.functional.silu(self.up_proj(x)))


class _Tiny(torch.nn.Module):
    def __init__(self, d=512, f=1024, dtype=torch.float16):
        super().__init__()
        self.layers = torch.nn.ModuleList([_Block(d, f, dtype), _Block(d, f, dtype)])
        self.lm_head = torch.nn.Linear(d, 64, bias=False, dtype=dtype)

    def forward(self, x):
        for blk in self.layers:
            x = x + blk(x)
        return x


def test_replace_with_flute_linear_swaps_everything_but_the_skipped_modules():
    model = _Tiny()
    model, replaced = hf.replace_with_flute_linear(model, num_bits=4, group_size=64)
    assert replaced
    assert isinstance(model.lm_head, torch.nn.Linear) and not isinstance(model.lm_head, FluteLinear)
    for blk in model.layers:
        for name, (k, n, has_bias) in {"up_proj": (512, 1024, False), "down_proj": (1024, 512, True)}.items():
            lin = getattr(blk, name)
            assert isinstance(lin, FluteLinear) and lin.needs_repacking and lin.template_id is None
            assert lin.source_cls is torch.nn.Linear
            assert lin.weight.shape == (4 * n // 16, k) and lin.weight.dtype == torch.int16
            assert lin.weight.device.type == "meta" and lin.scales.shape == (n, k // 64)
            assert (lin.bias is not None) == has_bias
            assert not any(p.requires_grad for p in lin.parameters())
    # dotted paths and prefixes are honoured (reference :103-108)
    model2, _ = hf.replace_with_flute_linear(_Tiny(), 4, 64, modules_to_not_convert=["layers.0", "lm_head"])
    assert not isinstance(model2.layers[0].up_proj, FluteLinear)
    assert isinstance(model2.layers[1].up_proj, FluteLinear)
    # the checkpoint's extra state supplies the id the weights were packed with
    lin = model.layers[0].up_proj
    lin.set_extra_state({"num_bits": 4, "group_size": 64, "template_id": 7})
    assert lin.template_id == 7
    with pytest.raises(ValueError):
        lin.set_extra_state({"num_bits": 4, "group_size": 64, "template_id": 8})


def test_template_id_callback_and_nothing_to_replace():
    seen = []

    def tid(N, K, dtype):
        seen.append((N, K, dtype))
        return 3
    model, _ = hf.replace_with_flute_linear(_Tiny(), 4, 64, template_id_of=tid)
    assert model.layers[1].down_proj.template_id == 3
    assert (1024, 512, torch.float16) in seen and (512, 1024, torch.float16) in seen
    _, replaced = hf.replace_with_flute_linear(torch.nn.Sequential(torch.nn.ReLU()), 4, 64)
    assert not replaced


def test_transformers_registration_and_quantizer_hooks():
    transformers = pytest.importorskip("transformers")
    from transformers.quantizers.auto import AUTO_QUANTIZATION_CONFIG_MAPPING, AUTO_QUANTIZER_MAPPING
    assert AUTO_QUANTIZER_MAPPING["flute"] is hf.FluteHfQuantizer
    assert AUTO_QUANTIZATION_CONFIG_MAPPING["flute"] is hf.FluteConfig
    cfg = hf.FluteConfig(num_bits=4, group_size=64, num_sms_packed=108, example_batch_size=1)
    assert hf.FluteConfig.from_dict(cfg.to_dict()).num_sms_packed == 108
    with pytest.raises(ValueError):
        hf.FluteConfig(num_bits=5)
    q = hf.FluteHfQuantizer(cfg, pre_quantized=True)
    assert q.is_trainable is False and q.is_serializable() is True
    with pytest.raises(TypeError):
        q.update_dtype(None)
    with pytest.raises((NotImplementedError, ValueError)):
        hf.FluteHfQuantizer(cfg, pre_quantized=False)
    llama_cfg = transformers.LlamaConfig(hidden_size=512, intermediate_size=1024, num_hidden_layers=1,
                                         num_attention_heads=4, num_key_value_heads=4, vocab_size=128)
    with torch.device("meta"):
        model = transformers.LlamaForCausalLM(llama_cfg).to(torch.float16)
    q._process_model_before_weight_loading(model)
    layer = model.model.layers[0]
    for lin in (layer.self_attn.q_proj, layer.self_attn.o_proj, layer.mlp.gate_proj, layer.mlp.down_proj):
        assert isinstance(lin, FluteLinear) and lin.needs_repacking
    assert not isinstance(model.lm_head, FluteLinear)
    assert model.config.quantization_config is cfg


@pytest.mark.gpu
def test_checkpoint_packed_for_another_gpu_is_repacked_on_load():
    """A state dict packed with TileP=64 template ids for a 108-SM GPU loads into the replaced model and is
    re-laid-out for this GPU: same codes, outputs equal to a model packed natively."""
    import flute_amd
    from flute_amd import utils
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    dtype, bits, g = torch.float16, 4, 64
    foreign_tid = 0                                    # b=4: ids with (id % 48) < 16 are TileP = 64
    assert utils.get_template_config(bits, foreign_tid, 108)["tileP"] == 64
    table = torch.randn(2 ** bits).to(dtype)
    ref_model = _Tiny(dtype=dtype)                     # only its shapes and biases are used
    sd, native = {}, {}
    for name, mod in ref_model.named_modules():
        if isinstance(mod, torch.nn.Linear) and name != "lm_head":
            K, N = mod.in_features, mod.out_features
            codes = torch.randint(0, 2 ** bits, (K, N), dtype=torch.uint8)
            scales = (torch.randn(N, K // g) / 8).to(dtype)
            sd[f"{name}.weight"] = utils.pack(codes, bits, [foreign_tid], 108)
            sd[f"{name}.scales"] = scales
            sd[f"{name}.tables"] = table
            sd[f"{name}.tables2"] = utils.make_qmap2_from_qmap(table)
            sd[f"{name}._extra_state"] = {"num_bits": bits, "group_size": g, "template_id": foreign_tid}
            if mod.bias is not None:
                sd[f"{name}.bias"] = mod.bias.detach().clone()
            native[name] = (codes, scales)
    sd["lm_head.weight"] = ref_model.lm_head.weight.detach().clone()

    with torch.device("meta"):
        model = _Tiny(dtype=dtype)
    model, replaced = hf.replace_with_flute_linear(model, bits, g)
    assert replaced
    model.load_state_dict(sd, assign=True)
    model = model.to(dev)
    assert model.layers[0].up_proj.template_id == foreign_tid
    n = hf.repack_flute_linear(model, num_sms_packed=108, example_batch_size=1)
    assert n == 4

    x = (torch.randn(3, 512) / 4).to(dtype).to(dev)
    with torch.no_grad():
        y = model(x)
    for name, (codes, scales) in native.items():
        lin = model.get_submodule(name)
        assert not lin.needs_repacking
        got = utils.unpack_codes(lin.weight, bits, lin.template_id).cpu()
        assert torch.equal(got, codes), name
        assert torch.equal(lin.tables2.cpu(), utils.make_qmap2_from_qmap(table))
    # same model built directly for this GPU
    direct = _Tiny(dtype=dtype).to(dev)
    for name, (codes, scales) in native.items():
        bias = sd.get(f"{name}.bias")
        lin = model.get_submodule(name)
        new = FluteLinear.from_codes(codes.to(dev), scales.to(dev), table.to(dev), bits, g, lin.template_id,
                                     bias=None if bias is None else bias.to(dev))
        parent, leaf = name.rsplit(".", 1)
        setattr(direct.get_submodule(parent), leaf, new)
    with torch.no_grad():
        y_direct = direct(x)
    assert torch.equal(y, y_direct)
    assert flute_amd.__name__ == "flute_amd"


@pytest.mark.gpu
def test_checkpoint_without_extra_state_loads_as_reference_tilep():
    """Safetensors checkpoints carry no `_extra_state`: the layer has no template id after loading and is
    unpacked with the TileP of the reference's bundled tuned table (32 for every entry), then repacked."""
    from flute_amd import utils
    dev = torch.device("cuda:0")
    torch.manual_seed(1)
    dtype, bits, g = torch.bfloat16, 3, 64
    ref_tid = hf.template_id_with_tile_p(bits, hf.reference_packed_tile_p())
    table = torch.randn(2 ** bits).to(dtype)
    ref_model = _Tiny(dtype=dtype)
    sd, native = {}, {}
    for name, mod in ref_model.named_modules():
        if isinstance(mod, torch.nn.Linear) and name != "lm_head":
            K, N = mod.in_features, mod.out_features
            codes = torch.randint(0, 2 ** bits, (K, N), dtype=torch.uint8)
            sd[f"{name}.weight"] = utils.pack(codes, bits, [ref_tid], 108)
            sd[f"{name}.scales"] = (torch.randn(N, K // g) / 8).to(dtype)
            sd[f"{name}.tables"] = table
            sd[f"{name}.tables2"] = utils.make_qmap2_from_qmap(table)
            if mod.bias is not None:
                sd[f"{name}.bias"] = mod.bias.detach().clone()
            native[name] = codes
    sd["lm_head.weight"] = ref_model.lm_head.weight.detach().clone()
    with torch.device("meta"):
        model = _Tiny(dtype=dtype)
    model, _ = hf.replace_with_flute_linear(model, bits, g)
    missing = model.load_state_dict(sd, assign=True, strict=False)
    assert all(k.endswith("_extra_state") for k in missing.missing_keys), missing
    model = model.to(dev)
    assert model.layers[0].up_proj.template_id is None
    assert hf.repack_flute_linear(model, num_sms_packed=108, example_batch_size=1) == 4
    for name, codes in native.items():
        lin = model.get_submodule(name)
        assert torch.equal(utils.unpack_codes(lin.weight, bits, lin.template_id).cpu(), codes), name
    y = model((torch.randn(2, 512) / 4).to(dtype).to(dev))
    assert torch.isfinite(y).all()


def test_reference_tilep_table_is_shipped():
    assert hf.reference_packed_tile_p() == 32
    for bits in (2, 3, 4):
        tid = hf.template_id_with_tile_p(bits, 32)
        import flute_amd
        assert flute_amd.TEMPLATE_CONFIGS[(bits, tid)]["TileP"] == 32


@pytest.mark.gpu
def test_transformers_higgs_linear_forward_runs_on_flute_amd():
    """BASELINE.json configs[4] through its real caller: transformers' `HiggsLinear.forward`
    (integrations/higgs.py) pads x to the Hadamard block and calls `flute.tune.qgemm_v2(x, weight, scales, tables,
    tables2.view(float32), workspace, tune_metadata, hadamard_size=...)`, with the buffers produced by
    `flute.integrations.higgs.prepare_data_transposed`.  transformers binds those names at import time only when a
    pip-installed `flute` distribution is present, so they are bound by hand here."""
    pytest.importorskip("transformers")
    import transformers.integrations.higgs as H
    import flute_amd
    from flute_amd import tune, utils
    from flute_amd.integrations import higgs
    from oracle import flute_oracle as O
    H.qgemm_v2, H.TuneMetaData = tune.qgemm_v2, tune.TuneMetaData
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    dtype, bits, g, had, vec = torch.float16, 4, 256, 512, 2
    K, N = 1024, 512                                               # K a multiple of the Hadamard block
    layer = H.HiggsLinear(K, N, num_bits=bits, bias=False, dtype=dtype, device=dev, group_size=g,
                          hadamard_size=had)
    codes = torch.randint(0, 2 ** (bits * vec), (N, K // vec), dtype=torch.uint8, device=dev)
    scales = (torch.randn(N, K // g, device=dev) / 4).to(dtype)
    grid = torch.randn(2 ** (bits * vec), vec, device=dev).to(dtype)
    Q, S, tables, tables2, meta = higgs.prepare_data_transposed(
        codes, scales, grid, bits, g, vec, dtype, dev, example_batch_size=1, check_correctness=False)
    layer.weight.data = Q
    layer.scales.data = S
    layer.tables.data = tables
    layer.tables2.data = tables2.view(dtype=dtype).view(2 ** bits, 2 ** bits, 2)     # how transformers stores it
    layer.workspace = utils.get_workspace_streamk(dev)
    layer.tune_metadata = meta
    x = (torch.randn(5, K, device=dev) / 8).to(dtype)
    y = layer(x)
    assert y.shape == (5, N) and y.dtype == dtype
    # reference: rotate x blockwise, then x @ dequant(W)^T with the vector codebook (tests/higgs.py:7-17)
    What = O.vector_dequantize_higgs(codes.cpu(), scales.cpu(), grid.cpu())           # [N, K] in T
    xr = O.hadamard_transform(x.cpu(), had)
    ref = (xr.double() @ What.double().T)
    err = ((y.cpu().double() - ref).norm() / ref.norm()).item()
    assert err < 3e-3, err
    assert flute_amd.qgemm_hadamard is not None


@pytest.mark.gpu
def test_quantize_hf_model_cli_round_trip(tmp_path):
    """flute/integrations/base.py:329-388: quantize a (tiny, random) Llama checkpoint with the command-line
    quantizer, load the saved checkpoint into a model whose linears were replaced by empty FluteLinear layers,
    repack for this GPU, and compare the logits with the kernel-faithful fake quantization of the same model."""
    transformers = pytest.importorskip("transformers")
    import json
    import subprocess
    import sys
    from safetensors.torch import load_file
    from flute_amd.integrations import base
    torch.manual_seed(0)
    cfg = transformers.LlamaConfig(hidden_size=512, intermediate_size=1024, num_hidden_layers=2,
                                   num_attention_heads=4, num_key_value_heads=4, vocab_size=256,
                                   max_position_embeddings=128)
    src, out, fake = tmp_path / "fp16", tmp_path / "flute", tmp_path / "fake"
    transformers.LlamaForCausalLM(cfg).to(torch.float16).save_pretrained(src)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "flute_amd.integrations.base", "--pretrained_model_name_or_path", str(src),
                        "--save_directory", str(out), "--num_bits", "4", "--group_size", "64", "--torch_dtype", "float16",
                        "--example_batch_size", "1"], cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert json.load(open(out / base.FLUTE_CONFIG_FILE_NAME)) == {"version": flute_amd.__version__, "num_bits": 4, "group_size": 64}
    base.quantize_hf_model(str(src), str(fake), 4, 64, "float16", 1, fake=True)

    dev = torch.device("cuda:0")
    state = {}
    for f in sorted(os.listdir(out)):
        if f.endswith(".safetensors"):
            state.update(load_file(str(out / f)))
    assert not any(k.endswith("_extra_state") for k in state)
    assert state["model.layers.0.mlp.down_proj.weight"].dtype == torch.int16
    model = transformers.LlamaForCausalLM(cfg).to(torch.float16)
    hf.replace_with_flute_linear(model.model.layers, 4, 64, modules_to_not_convert=[])
    for m in model.modules():                                   # materialise the empty layers, then load
        if isinstance(m, FluteLinear):
            m.to_empty(device=dev)
    model.to(dev)
    missing, unexpected = model.load_state_dict(state, strict=False)
    assert not unexpected and all(k.endswith("_extra_state") or "lm_head" in k for k in missing), (missing, unexpected)
    assert hf.repack_flute_linear(model, num_sms_packed=108) == 14          # 2 layers x 7 linears
    ref = transformers.AutoModelForCausalLM.from_pretrained(str(fake), torch_dtype=torch.float16).to(dev)
    ids = torch.randint(0, 256, (2, 16), device=dev)
    with torch.no_grad():
        a = model(ids).logits.float()
        b = ref(ids).logits.float()
    err = ((a - b).norm() / b.norm()).item()
    assert err < 5e-3, err
