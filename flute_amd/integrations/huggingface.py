"""Load-time path of a FLUTE checkpoint (flute/integrations/huggingface.py:84-236, 239-352).

A checkpoint on the Hub holds, per linear layer, `weight[P,K] int16` packed for the
GPU it was tuned on (an A100/A6000 template id), `scales`, `tables`, `tables2` and
the layer's extra state `{num_bits, group_size, template_id}`.  Loading is three steps:

1. `replace_with_flute_linear`: every `nn.Linear` outside `modules_to_not_convert`
   becomes an empty `FluteLinear` flagged `needs_repacking` (reference `:84-170`);
2. the framework loads the state dict into those buffers;
3. `repack_flute_linear`: each flagged layer is re-laid-out for THIS GPU
   (reference `:173-236`).  The reference reconstructs the codes with an identity
   qgemm of the foreign layout; here the native unpacker reads them straight out
   of `Q` (template id -> TileP is all that matters of the foreign layout), the
   gfx950 tuner picks a plan and the codes are packed again.

What differs from the reference, on purpose:
* the packed template id comes from the checkpoint's own extra state when it has one.  Safetensors
  checkpoints on the Hub do not: the reference then takes the id from its bundled per-GPU table
  `qgemm_kernel_raw_tuned_configs.no-M.pth` (`:53-83`).  Only the TileP of that id matters to the native
  unpacker, and EVERY entry of that table (3 816: A100 / A6000 / RTX 4090, 2-4 bits, g 32-256, fp16 / bf16)
  is a TileP = 32 template (`flute_amd/data/ref_packed_tilep.json`, generated and asserted by
  `tools/make_ref_tilep_table.py`): a layer without a template id is unpacked as TileP = 32.  The caller
  may still point `legacy_template_table=` at a `.pth` table of ids;
* one distinct (N, K) shape is tuned once per model, not once per layer, and shapes found in the shipped
  gfx950 table (`flute_amd/data/gfx950_tuned.json`) are not timed at all.

`FluteConfig` / `FluteHfQuantizer` register the method with `transformers`
(written against the 5.x quantizer API of this image); they are defined only if
`transformers` imports, the two functions above need torch alone.
"""
import json
import os
import warnings
from typing import Callable, Dict, List, Optional, Tuple

import torch

import flute_amd
import flute_amd.utils
from flute_amd import tune
from flute_amd.integrations.base import FluteLinear

FLUTE_CONFIG_FILE_NAME = "flute_config.json"        # flute/integrations/base.py:21
_REF_TILEP_FILE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "data",
                               "ref_packed_tilep.json")


def reference_packed_tile_p() -> int:
    """TileP of every template the reference's bundled tuned table assigns (see the module docstring)."""
    with open(_REF_TILEP_FILE) as f:
        return int(json.load(f)["default_tile_p"])


def template_id_with_tile_p(num_bits: int, tile_p: int) -> int:
    """A template id of this library's table with the given packed layout (id -> TileP is the reference's map)."""
    return min(t for (b, t), c in flute_amd.TEMPLATE_CONFIGS.items() if b == num_bits and c["TileP"] == tile_p)


def _skip(name: str, path: str, modules_to_not_convert: List[str]) -> bool:
    # reference :103-108 - by leaf name, by dotted path, or by a dotted-path prefix
    return name in modules_to_not_convert or any(
        (key + "." in path) or (key == path) for key in modules_to_not_convert)


def replace_with_flute_linear(model: torch.nn.Module, num_bits: int, group_size: int,
                              modules_to_not_convert: Optional[List[str]] = None,
                              template_id_of: Optional[Callable[[int, int, torch.dtype], int]] = None,
                              _prefix: str = "") -> Tuple[torch.nn.Module, bool]:
    """Swap `nn.Linear` children for empty (meta-device) `FluteLinear` layers awaiting
    a state dict.  `template_id_of(N, K, dtype)` supplies the id the checkpoint was
    packed with when its extra state does not carry one."""
    skip = list(modules_to_not_convert) if modules_to_not_convert is not None else ["lm_head"]
    replaced = False
    for name, module in model.named_children():
        path = f"{_prefix}.{name}" if _prefix else name
        if isinstance(module, torch.nn.Linear) and not _skip(name, path, skip):
            layer = FluteLinear(
                in_features=module.in_features, out_features=module.out_features,
                num_bits=num_bits, group_size=group_size,
                template_id=(template_id_of(module.out_features, module.in_features, module.weight.dtype)
                             if template_id_of is not None else None),
                workspace_lazy_init=True, bias=module.bias is not None,
                device=torch.device("meta"), dtype=module.weight.dtype)
            layer.source_cls = type(module)
            layer.requires_grad_(False)
            layer.needs_repacking = True
            model._modules[name] = layer
            replaced = True
        elif len(list(module.children())) > 0:
            _, r = replace_with_flute_linear(module, num_bits, group_size, skip, template_id_of, path)
            replaced = replaced or r
    return model, replaced


@torch.no_grad()
def repack_flute_linear(model: torch.nn.Module, num_sms_packed: int, example_batch_size: int = 1,
                        _cache: Optional[Dict] = None) -> int:
    """Re-lay-out every `FluteLinear` flagged `needs_repacking` for the local GPU; returns
    how many layers were repacked.  Layers of one (N, K, bits, group size, dtype) share
    one tuner run."""
    cache = {} if _cache is None else _cache
    count = 0
    for name, module in model.named_children():
        if isinstance(module, FluteLinear) and getattr(module, "needs_repacking", False):
            if module.template_id is None:
                # no extra state (safetensors) and no table of ids: the reference's own table only ever
                # assigns TileP = 32 templates, and TileP is all the unpacker needs
                module.template_id = template_id_with_tile_p(module.num_bits, reference_packed_tile_p())
            home = module.weight.device
            device = home if home.type == "cuda" else torch.device("cuda")
            if home.type != "cuda":
                warnings.warn(f"[FLUTE]: moving {name} to {device} for repacking")
            Q = module.weight.to(device)
            N, K = module.out_features, module.in_features
            key = (N, K, module.num_bits, module.group_size, module.scales.dtype, example_batch_size)
            codes = flute_amd.utils.unpack_codes(Q, module.num_bits, module.template_id)      # [K, N]
            if key not in cache:
                example = torch.randn(example_batch_size, K, dtype=module.scales.dtype, device=device)
                Q_new, meta = tune.tune_and_pack(example, codes, module.num_bits, module.group_size)
                cache[key] = meta.template_id
            else:
                Q_new = flute_amd.utils.pack(codes, module.num_bits, [cache[key]],
                                             flute_amd.utils.get_device_num_sms(device))
            if Q_new.shape != module.weight.shape or Q_new.dtype != module.weight.dtype:
                raise ValueError
            # the loader may have cast `tables2` (a 32-bit container): regenerate it (reference :221-232)
            tables2 = flute_amd.utils.make_qmap2_from_qmap(module.tables)
            if tables2.shape != module.tables2.shape:
                raise ValueError
            module.weight = Q_new.to(home)
            module.tables2 = tables2.to(module.tables2.device)
            module.template_id = cache[key]
            module.needs_repacking = False
            count += 1
        elif len(list(module.children())) > 0:
            count += repack_flute_linear(module, num_sms_packed, example_batch_size, cache)
    return count


def legacy_template_lookup(table_path: str, num_sms_packed: int, num_bits: int, group_size: int
                           ) -> Callable[[int, int, torch.dtype], int]:
    """`template_id_of` backed by the reference's per-GPU table of tuned ids
    (`qgemm_kernel_raw_tuned_configs.no-M.pth`, keyed `(num_sms, bits, g, N, K, str(dtype))`,
    reference :70-83) for checkpoints whose extra state is missing."""
    table = torch.load(table_path, weights_only=True)

    def lookup(N: int, K: int, dtype: torch.dtype) -> int:
        return int(table[(num_sms_packed, num_bits, group_size, N, K, str(dtype))])
    return lookup


try:                                                   # transformers is optional plumbing
    from transformers.quantizers.auto import register_quantization_config, register_quantizer
    from transformers.quantizers.base import HfQuantizer
    from transformers.utils.quantization_config import QuantizationConfigMixin
    _HAVE_TRANSFORMERS = True
except Exception:                                      # pragma: no cover - image without transformers
    _HAVE_TRANSFORMERS = False


if _HAVE_TRANSFORMERS:

    @register_quantization_config("flute")
    class FluteConfig(QuantizationConfigMixin):
        """reference :31-83"""

        def __init__(self, num_bits: int = 4, group_size: int = 64, num_sms_packed: int = 108,
                     example_batch_size: int = 1, modules_to_not_convert: Optional[List[str]] = None,
                     legacy_template_table: Optional[str] = None, **kwargs) -> None:
            if num_bits not in (2, 3, 4):
                raise ValueError
            self.quant_method = "flute"
            self.num_bits = num_bits
            self.group_size = group_size
            self.num_sms_packed = num_sms_packed
            self.example_batch_size = example_batch_size
            self.modules_to_not_convert = modules_to_not_convert
            self.legacy_template_table = legacy_template_table

        def template_id_of(self) -> Optional[Callable[[int, int, torch.dtype], int]]:
            if self.legacy_template_table is None:
                return None
            return legacy_template_lookup(self.legacy_template_table, self.num_sms_packed,
                                          self.num_bits, self.group_size)

    @register_quantizer("flute")
    class FluteHfQuantizer(HfQuantizer):
        """reference :239-318: pre-quantized checkpoints only."""
        requires_calibration = True

        def __init__(self, quantization_config, **kwargs) -> None:
            super().__init__(quantization_config, **kwargs)
            if not self.pre_quantized:
                raise NotImplementedError("FLUTE loads pre-quantized checkpoints only")

        def validate_environment(self, *args, **kwargs) -> None:
            flute_amd._lib.get()                       # raises if libflute_amd.so is missing

        def update_dtype(self, dtype):
            if dtype is None:
                raise TypeError("specify `dtype` (float16 or bfloat16) in `from_pretrained`")
            return dtype

        def _process_model_before_weight_loading(self, model, keep_in_fp32_modules=None, **kwargs) -> None:
            cfg = self.quantization_config
            skip = self.get_modules_to_not_convert(model, cfg.modules_to_not_convert, keep_in_fp32_modules)
            _, replaced = replace_with_flute_linear(model, cfg.num_bits, cfg.group_size, skip,
                                                    cfg.template_id_of())
            if not replaced:
                warnings.warn("FLUTE quantization requested but the model has no linear layer to convert")
            model.config.quantization_config = cfg

        def _process_model_after_weight_loading(self, model, **kwargs):
            cfg = self.quantization_config
            repack_flute_linear(model, cfg.num_sms_packed, cfg.example_batch_size)
            return model

        @property
        def is_trainable(self) -> bool:
            return False

        def is_serializable(self, **kwargs) -> bool:
            return True
