"""vLLM glue (flute/integrations/vllm_utils.py:42-349, vllm.py:9-46) against the CURRENT operator signature.

vLLM is not part of the build image, so the two vLLM-facing classes (`FluteConfig`, `FluteLinearMethod`) are
only defined when `vllm` imports.  They are executed on the GPU against a stub of the four vLLM names they use
(tests/test_vllm_utils_gpu.py: create_weights -> a `params_dict[name]` load of a FluteLinear state dict ->
process_weights_after_loading -> apply); the part that carries the logic - `repack_loaded_shard`, what
`process_weights_after_loading` does to the tensors vLLM's loader produced - is a plain function and is tested
by loading shards the way vLLM cuts them.

Difference from the reference: it all-gathers every shard (int16 cast to int32 for NCCL), unpacks the full
matrix with an identity-matrix qgemm, re-shards and repacks on the CPU (vllm_utils.py:228-326).  The packed format
interleaves columns only inside blocks of (16 / bits) x TileP columns and pairs of k, so the row / column slices
vLLM's loader cuts ARE valid packed matrices (flute_amd/tp.py): each rank unpacks its own shard with the native
unpacker, concatenates the fused partitions (q/k/v, gate/up) as codes, tunes and packs for this GPU.  No
collective, no CPU round trip.  3-bit layers interleave three bit planes over the whole column range and cannot
be N-sharded by a contiguous row slice: their loader must call `flute_amd.tp.shard_columns` (raised here otherwise).
"""
from typing import Any, Dict, List, Optional, Sequence, Tuple

import torch

import flute_amd
import flute_amd.utils
from flute_amd import tune
from flute_amd.integrations.huggingface import reference_packed_tile_p, template_id_with_tile_p


def packed_rows(num_columns: int, num_bits: int) -> int:
    """P of a layer with N columns (the reference's `PackFactor`, vllm_utils.py:23-39: N / 16 * bits)."""
    if num_columns % 16:
        raise ValueError
    return num_columns // 16 * num_bits


@torch.no_grad()
def repack_loaded_shard(weight: torch.Tensor, scales: torch.Tensor, output_partition_sizes: Sequence[int],
                        num_bits: int, group_size: int, example_batch_size: int = 1,
                        packed_tile_p: Optional[int] = None) -> Tuple[torch.Tensor, int]:
    """`weight[P, K]` is what vLLM's loader assembled for this rank: the packed rows of every fused partition
    (this rank's row slice of each, for column-parallel layers; its K slice, for row-parallel layers), stacked in
    `output_partition_sizes` order; `scales[N, G]` likewise.  Returns the layer re-tuned and re-packed for this GPU
    and the template id to launch it with.  `packed_tile_p`: TileP of the checkpoint (default: the reference's)."""
    if weight.device.type != "cuda":
        raise ValueError("repacking runs on the GPU")
    tile_p = reference_packed_tile_p() if packed_tile_p is None else packed_tile_p
    tid_in = template_id_with_tile_p(num_bits, tile_p)
    block = tile_p * (16 if num_bits == 3 else 16 // num_bits)
    Ps = [packed_rows(n, num_bits) for n in output_partition_sizes]
    if sum(Ps) != weight.shape[0] or sum(output_partition_sizes) != scales.shape[0]:
        raise ValueError("partition sizes do not match the loaded tensors")
    codes = []
    for Q, n in zip(torch.split(weight, Ps, dim=0), output_partition_sizes):
        if n % block:
            raise NotImplementedError(f"a partition of {n} columns is not a whole number of {block}-column blocks")
        codes.append(flute_amd.utils.unpack_codes(Q.contiguous(), num_bits, tid_in))          # [K, n]
    codes = torch.cat(codes, dim=1).contiguous()
    example = torch.randn(example_batch_size, codes.shape[0], dtype=scales.dtype, device=weight.device)
    Q_new, meta = tune.tune_and_pack(example, codes, num_bits, group_size)
    if Q_new.shape != weight.shape or Q_new.dtype != weight.dtype:
        raise ValueError
    return Q_new, meta.template_id


try:                                                   # pragma: no cover - vLLM is not in the build image
    from torch.nn.parameter import Parameter
    from vllm.model_executor.layers.linear import LinearBase, LinearMethodBase, set_weight_attrs
    from vllm.model_executor.layers.quantization.base_config import QuantizationConfig
    _HAVE_VLLM = True
except Exception:
    _HAVE_VLLM = False


if _HAVE_VLLM:                                         # pragma: no cover

    class _PackFactor:
        """`x // pack_factor` = packed rows of x columns, also for 3 bits (vllm_utils.py:23-39)."""

        def __init__(self, num_bits: int) -> None:
            self.num_bits = num_bits

        def __rfloordiv__(self, other: int) -> int:
            return packed_rows(other, self.num_bits)

    class FluteConfig(QuantizationConfig):
        """vllm_utils.py:42-107"""

        def __init__(self, num_bits: int, group_size: int, num_sms_packed: int = 108) -> None:
            super().__init__()
            if num_bits not in (2, 3, 4):
                raise ValueError
            self.num_bits, self.group_size, self.num_sms_packed = num_bits, group_size, num_sms_packed
            self.pack_factor = _PackFactor(num_bits)

        def __repr__(self) -> str:
            return f"FluteConfig(num_bits={self.num_bits}, group_size={self.group_size}, num_sms_packed={self.num_sms_packed})"

        @classmethod
        def get_name(cls) -> str:
            return "flute"

        @classmethod
        def get_supported_act_dtypes(cls) -> List[torch.dtype]:
            return [torch.float16, torch.bfloat16]

        @classmethod
        def get_min_capability(cls) -> int:
            return 0                                   # ROCm: the capability gate is CUDA-specific

        @classmethod
        def get_config_filenames(cls) -> List[str]:
            return ["flute_config.json"]

        @classmethod
        def from_config(cls, config: Dict[str, Any]) -> "FluteConfig":
            return cls(num_bits=cls.get_from_keys(config, ["num_bits"]), group_size=cls.get_from_keys(config, ["group_size"]),
                       num_sms_packed=cls.get_from_keys_or(config, ["num_sms"], 108))

        def get_quant_method(self, layer: torch.nn.Module, prefix: str) -> Optional["FluteLinearMethod"]:
            return FluteLinearMethod(self) if isinstance(layer, LinearBase) else None

        def get_scaled_act_names(self) -> List[str]:
            return []

    class FluteLinearMethod(LinearMethodBase):
        """vllm_utils.py:110-349"""

        def __init__(self, quant_config: FluteConfig) -> None:
            self.quant_config = quant_config

        def create_weights(self, layer: torch.nn.Module, input_size_per_partition: int,
                           output_partition_sizes: List[int], input_size: int, output_size: int,
                           params_dtype: torch.dtype, **extra_weight_attrs) -> None:
            if params_dtype not in (torch.float16, torch.bfloat16):
                raise TypeError
            cfg = self.quant_config
            K, N = input_size_per_partition, sum(output_partition_sizes)
            if cfg.num_bits == 3 and N != output_size:
                raise NotImplementedError("3-bit layers: shard columns with flute_amd.tp.shard_columns in the weight loader")
            dev = torch.device("cuda", torch.cuda.current_device())
            weight = Parameter(torch.empty((packed_rows(N, cfg.num_bits), K), dtype=torch.int16, device=dev), requires_grad=False)
            set_weight_attrs(weight, {**extra_weight_attrs, "input_dim": 1, "output_dim": 0, "packed_dim": 0,
                                      "pack_factor": cfg.pack_factor})
            scales = Parameter(torch.empty((N, K // cfg.group_size), dtype=params_dtype, device=dev), requires_grad=False)
            set_weight_attrs(scales, {**extra_weight_attrs, "input_dim": 1, "output_dim": 0})
            tables = Parameter(torch.arange(2 ** cfg.num_bits, dtype=params_dtype, device=dev), requires_grad=False)
            set_weight_attrs(tables, {**extra_weight_attrs, "input_dim": None, "output_dim": None, "ignore_warning": True})
            # the pair table is part of every FLUTE checkpoint (`...tables2`, vllm_utils.py:196-218): vLLM's loader
            # looks every checkpoint tensor up in params_dict, so it must exist as a parameter; the loaded value
            # is replaced by make_qmap2_from_qmap(tables) after loading (loaders may have cast it, huggingface.py:221-232)
            tables2 = Parameter(torch.zeros((2 ** cfg.num_bits, 2 ** cfg.num_bits, 1), dtype=torch.float32, device=dev),
                                requires_grad=False)
            set_weight_attrs(tables2, {**extra_weight_attrs, "input_dim": None, "output_dim": None, "ignore_warning": True})
            layer.register_parameter("weight", weight)
            layer.register_parameter("scales", scales)
            layer.register_parameter("tables", tables)
            layer.register_parameter("tables2", tables2)
            layer.flute_output_partition_sizes = list(output_partition_sizes)
            layer.flute_template_id = None
            layer.needs_repacking = True

        def process_weights_after_loading(self, layer: torch.nn.Module) -> None:
            if not getattr(layer, "needs_repacking", False):
                return
            cfg = self.quant_config
            Q, tid = repack_loaded_shard(layer.weight.data, layer.scales.data, layer.flute_output_partition_sizes,
                                         cfg.num_bits, cfg.group_size)
            layer.weight = Parameter(Q, requires_grad=False)
            layer.flute_template_id = tid
            layer.tables2 = Parameter(flute_amd.utils.make_qmap2_from_qmap(layer.tables.data), requires_grad=False)
            layer.flute_workspace = flute_amd.utils.get_workspace_streamk(Q.device)
            layer.flute_num_sms = flute_amd.utils.get_device_num_sms(Q.device)
            layer.needs_repacking = False

        def apply(self, layer: torch.nn.Module, x: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
            out = flute_amd.qgemm(x, layer.weight, layer.scales, layer.tables, layer.tables2, layer.flute_workspace,
                                  self.quant_config.num_bits, self.quant_config.group_size, layer.flute_template_id,
                                  layer.flute_num_sms)
            return out if bias is None else out.add_(bias)


def patch_vllm() -> None:                              # pragma: no cover
    """flute/integrations/vllm.py:9-27: register the method with a running vLLM."""
    if not _HAVE_VLLM:
        raise ImportError("vllm is not installed")
    from vllm.model_executor.layers.quantization import QUANTIZATION_METHODS
    if isinstance(QUANTIZATION_METHODS, dict):
        QUANTIZATION_METHODS.setdefault("flute", FluteConfig)
    else:                                              # newer vLLM: a list of names + a registration decorator
        from vllm.model_executor.layers.quantization import register_quantization_config
        register_quantization_config("flute")(FluteConfig)
