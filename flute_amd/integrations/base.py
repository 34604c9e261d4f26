"""`FluteLinear` - the drop-in quantized linear layer (flute/integrations/base.py:203-326).

Same constructor, buffers (`weight[P,K] int16`, `scales[N,G]`, `tables[2^b]`,
`tables2`), extra-state and forward semantics (in-place bias add) as the
reference.  `prepare_model_flute` is the model-walking NormalFloat quantizer of
base.py:44-200 for plain `nn.Linear` layers; `quantize_hf_model` and the module's
command line are the checkpoint quantizer of base.py:329-388 (the bitsandbytes /
learnable-scale sources stay outside the hot path).

    python -m flute_amd.integrations.base --pretrained_model_name_or_path DIR --save_directory OUT \
        --num_bits 4 --group_size 64 --torch_dtype float16 --example_batch_size 1
"""
import argparse
import json
import os
import warnings
from typing import Dict, Optional

import torch

import flute_amd
import flute_amd.utils

FLUTE_CONFIG_FILE_NAME = "flute_config.json"        # flute/integrations/base.py:24


class FluteLinear(torch.nn.Module):
    __constants__ = ["in_features", "out_features", "num_bits", "group_size", "template_id",
                     "num_sms", "workspace_lazy_init"]

    def __init__(self, in_features: int, out_features: int, num_bits: int, group_size: int,
                 template_id: int, workspace_lazy_init: bool = False, bias: bool = False,
                 device: Optional[torch.device] = None, dtype: Optional[torch.dtype] = None) -> None:
        if dtype not in [torch.float16, torch.bfloat16]:
            raise NotImplementedError
        if not isinstance(device, torch.device):
            raise NotImplementedError
        super().__init__()
        K, N = in_features, out_features
        P = int(N / 16 * num_bits)
        G = int(K / group_size)
        tables = torch.arange(2 ** num_bits, dtype=dtype, device=device)
        if workspace_lazy_init:
            num_sms, workspace = None, None
        else:
            num_sms = flute_amd.utils.get_device_num_sms(device)
            workspace = flute_amd.utils.get_workspace_streamk(device)
        self.in_features = in_features
        self.out_features = out_features
        self.num_bits = num_bits
        self.group_size = group_size
        self.template_id = template_id
        self.num_sms = num_sms
        self.workspace = workspace
        self.workspace_lazy_init = workspace_lazy_init
        self.register_buffer("weight", torch.empty((P, K), dtype=torch.int16, device=device))
        self.register_buffer("scales", torch.ones((N, G), dtype=dtype, device=device))
        self.register_buffer("tables", tables)
        self.register_buffer("tables2", flute_amd.utils.make_qmap2_from_qmap(tables))
        if bias:
            self.bias = torch.nn.Parameter(torch.empty(out_features, device=device, dtype=dtype))
        else:
            self.register_parameter("bias", None)

    def forward(self, inputs: torch.Tensor) -> torch.Tensor:
        if self.workspace_lazy_init:
            num_sms = flute_amd.utils.get_device_num_sms(inputs.device)
            workspace = flute_amd.utils.get_workspace_streamk(inputs.device)
        else:
            num_sms, workspace = self.num_sms, self.workspace
        output = flute_amd.qgemm(inputs, self.weight, self.scales, self.tables, self.tables2,
                                 workspace, self.num_bits, self.group_size, self.template_id,
                                 num_sms)
        if self.bias is not None:
            output.add_(self.bias)   # in place, base.py:298-299
        return output

    def extra_repr(self) -> str:
        return (f"in_features={self.in_features}, out_features={self.out_features}, "
                f"bias={self.bias is not None}, num_bits={self.num_bits}, "
                f"group_size={self.group_size}")

    def get_extra_state(self) -> Dict:
        return {"num_bits": self.num_bits, "group_size": self.group_size,
                "template_id": self.template_id}

    def set_extra_state(self, state: Dict) -> None:
        if self.num_bits != state["num_bits"] or self.group_size != state["group_size"]:
            raise ValueError
        if self.template_id is None:
            self.template_id = state["template_id"]
        if self.template_id != state["template_id"]:
            raise ValueError

    @classmethod
    def from_codes(cls, codes: torch.Tensor, scales: torch.Tensor, tables: torch.Tensor,
                   num_bits: int, group_size: int, template_id: int,
                   bias: Optional[torch.Tensor] = None) -> "FluteLinear":
        """Build a layer from integer codes [K, N], scales [N, G] and a table."""
        K, N = codes.shape
        layer = cls(K, N, num_bits, group_size, template_id, bias=bias is not None,
                    device=scales.device, dtype=scales.dtype)
        num_sms = flute_amd.utils.get_device_num_sms(scales.device)
        layer.weight.copy_(flute_amd.utils.pack(codes.to(scales.device), num_bits,
                                                [template_id], num_sms))
        layer.scales.copy_(scales)
        layer.tables.copy_(tables)
        layer.tables2.copy_(flute_amd.utils.make_qmap2_from_qmap(tables.to(scales.dtype)))
        if bias is not None:
            with torch.no_grad():
                layer.bias.copy_(bias)
        return layer


@torch.no_grad()
def prepare_model_flute(name: str, module: torch.nn.Module, num_bits: int, group_size: int,
                        example_batch_size: int, fake: bool = False, handle_hooks: bool = False,
                        custom_scales_dict: Optional[Dict[str, torch.Tensor]] = None) -> None:
    """Replace every fp16/bf16 `nn.Linear` under `module` by a `FluteLinear` holding its
    NormalFloat-quantized weights, tuned and packed for this GPU (flute/integrations/base.py:44-200).

    `fake=True` keeps the layers and overwrites their weights with the kernel-faithful fake
    quantization (base.py:84-100), which is what the quantized model must reproduce.
    accelerate hooks on a replaced layer are moved to the new layer when `handle_hooks` is set."""
    import flute_amd.nf_utils
    import flute_amd.tune

    def _replace(_name: str, _module: torch.nn.Module) -> None:
        for child_name, child in _module.named_children():
            full = f"{_name}.{child_name}"
            if not isinstance(child, torch.nn.Linear):
                _replace(full, child)
                continue
            if child.weight.dtype not in (torch.float16, torch.bfloat16):
                raise NotImplementedError(f"{full}: only fp16 / bf16 layers are quantized")
            if child.in_features % group_size or child.in_features % 64 or \
                    child.out_features % (512 if num_bits == 3 else 128):
                raise ValueError(f"{full}: shape {tuple(child.weight.shape)} is not packable with group size {group_size}")
            dev = child.weight.device
            if dev.type != "cuda":
                raise ValueError(f"{full}: quantization and tuning run on the layer's GPU")
            if fake:
                new_weight = flute_amd.nf_utils.nf_quantize_2(child.weight, num_bits, group_size, child.weight.dtype)
                child.weight = torch.nn.Parameter(new_weight, requires_grad=False)     # assignment: no casts
                continue
            hook = None
            if handle_hooks:
                if child._backward_hooks or child._forward_hooks or child._forward_pre_hooks:
                    raise NotImplementedError(f"{full}: PyTorch hooks are not carried over")
                hook = getattr(child, "_hf_hook", None)
            elif getattr(child, "_hf_hook", None) is not None:
                raise ValueError(f"`{full}` has an accelerate hook (pass handle_hooks=True)")
            custom = custom_scales_dict[full] if custom_scales_dict is not None else None
            _, codes, scales, qmap = flute_amd.nf_utils.nf_quantize(child.weight, num_bits, group_size, custom)
            example = torch.randn(example_batch_size, child.in_features, dtype=child.weight.dtype, device=dev)
            Q, meta = flute_amd.tune.tune_and_pack(example, codes.to(torch.uint8).T.contiguous(), num_bits, group_size)
            new = FluteLinear(child.in_features, child.out_features, num_bits, group_size, meta.template_id,
                              workspace_lazy_init=False, bias=child.bias is not None, device=dev,
                              dtype=child.weight.dtype)
            new.weight.copy_(Q)
            new.scales.copy_(scales.view(new.scales.shape).to(new.scales.dtype))
            new.tables.copy_(qmap.to(new.tables.dtype))
            new.tables2.copy_(flute_amd.utils.make_qmap2_from_qmap(new.tables))
            if new.bias is not None:
                new.bias.copy_(child.bias)
            setattr(_module, child_name, new)
            if hook is not None:
                from accelerate.hooks import add_hook_to_module
                add_hook_to_module(new, hook)

    if not fake:
        warnings.warn("prepare_model_flute tunes every distinct layer shape on the GPU (a few seconds each)")
    _replace(name, module)


def quantize_hf_model(pretrained_model_name_or_path: str, save_directory: str, num_bits: int, group_size: int,
                      torch_dtype: str = "auto", example_batch_size: int = 1, fake: bool = False,
                      device: Optional[torch.device] = None) -> None:
    """NormalFloat-quantize the decoder layers of a Hugging Face causal LM and save the checkpoint plus
    `flute_config.json` (flute/integrations/base.py:329-367).  The reference restricts itself to Llama and Gemma-2
    (`isinstance` check, :343); here any `*ForCausalLM` that keeps its blocks in `model.model.layers` is accepted.
    Quantization, tuning and packing run on the GPU (`prepare_model_flute`); the embedding and `lm_head` stay as
    they are, as in the reference (only `model.model.layers` is walked)."""
    from transformers import AutoModelForCausalLM

    dtype = torch_dtype if torch_dtype == "auto" else getattr(torch, torch_dtype)
    model = AutoModelForCausalLM.from_pretrained(pretrained_model_name_or_path, device_map="cpu", torch_dtype=dtype)
    layers = getattr(getattr(model, "model", None), "layers", None)
    if layers is None:
        raise NotImplementedError(f"{type(model).__name__}: no `model.layers` to quantize")
    dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
    layers.to(dev)
    prepare_model_flute(name="model.model.layers", module=layers, num_bits=num_bits, group_size=group_size,
                        example_batch_size=example_batch_size, fake=fake)
    # A safetensors checkpoint cannot carry the layers' extra state (the template id = the TileP the weights were
    # packed with).  Checkpoints are therefore stored in the layout every entry of the reference's tuned table uses
    # (TileP 32, flute_amd/data/ref_packed_tilep.json): loadable with or without extra state, here
    # (integrations/huggingface.py re-tunes and repacks on load) and by the reference.
    if not fake:
        from flute_amd.integrations.huggingface import reference_packed_tile_p, template_id_with_tile_p
        canon = template_id_with_tile_p(num_bits, reference_packed_tile_p())
        for m in layers.modules():
            if isinstance(m, FluteLinear) and flute_amd.TEMPLATE_CONFIGS[(num_bits, m.template_id)]["TileP"] != reference_packed_tile_p():
                codes = flute_amd.utils.unpack_codes(m.weight, num_bits, m.template_id)
                m.weight.copy_(flute_amd.utils.pack(codes, num_bits, [canon], flute_amd.utils.get_device_num_sms(m.weight.device)))
                m.template_id = canon
    state = {k: v for k, v in model.state_dict().items() if not k.endswith("_extra_state")}
    model.save_pretrained(save_directory, state_dict=state)
    with open(os.path.join(save_directory, FLUTE_CONFIG_FILE_NAME), "w") as f:
        json.dump({"version": flute_amd.__version__, "num_bits": num_bits, "group_size": group_size}, f)


if __name__ == "__main__":
    parser = argparse.ArgumentParser()
    parser.add_argument("--pretrained_model_name_or_path", type=str, required=True)
    parser.add_argument("--save_directory", type=str, required=True)
    parser.add_argument("--num_bits", type=int, required=True)
    parser.add_argument("--group_size", type=int, required=True)
    parser.add_argument("--torch_dtype", type=str, default="auto")
    parser.add_argument("--example_batch_size", type=int, default=1)
    parser.add_argument("--fake", action="store_true")
    args = parser.parse_args()
    quantize_hf_model(args.pretrained_model_name_or_path, args.save_directory, args.num_bits, args.group_size,
                      args.torch_dtype, args.example_batch_size, args.fake)
