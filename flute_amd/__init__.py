"""flute_amd - MI355X (gfx950) implementation of the `flute.qgemm` hot path.

Same operator surface as HanGuo97/flute v0.4.2 (flute/__init__.py:12-69):
`qgemm`, `qgemm_hadamard` (handles to `torch.ops.flute.qgemm_raw_simple[_hadamard]`)
and `TEMPLATE_CONFIGS` keyed `(num_bits, template_id)`.  The device code is a
hand-written HIP library behind a C ABI (include/flute_amd.h); importing this
package fails loudly if that library has not been built.
"""
from typing import Callable, cast

import torch

from . import _lib
from . import ops

__version__ = "0.1.0"

_lib.get()   # fail at import time, not at first call, when the HIP library is missing

qgemm = cast(Callable[..., torch.Tensor], torch.ops.flute.qgemm_raw_simple.default)
qgemm_hadamard = cast(Callable[..., torch.Tensor], torch.ops.flute.qgemm_raw_simple_hadamard.default)
hadamard_transform = ops.hadamard_transform

_QUANT_MAP_MODE = {1: "kVectorized   ", 32: "kVectorized_32", 16: "kVectorized_16", 8: "kVectorized_8 "}


def _load_template_configs():
    """Built from the library's own table (single source of truth) in the
    reference's format (flute/codegen_utils.py:110-152)."""
    lib = _lib.get()
    configs = {}
    for bits in (4, 3, 2):
        for tid in range(lib.flute_num_templates(bits)):
            t = _lib.TemplateInfo()
            _lib.check(lib.flute_get_template_info(bits, tid, t))
            configs[(bits, tid)] = {
                "SMs_Multiple": t.sms_multiple,
                "Threads": t.threads,
                "TileM": t.tile_m,
                "TileK": t.tile_k,
                "TileP": t.tile_p,
                "Stages": t.stages,
                "QuantMapMode": _QUANT_MAP_MODE[t.lut_copies] if bits == 4 else "kVectorized   ",
                "AccumulationMode": "kMixed",
                "DecompositionMode": "kSplitK",
                "LutCopies": t.lut_copies,
            }
    return configs


TEMPLATE_CONFIGS = _load_template_configs()

from . import utils  # noqa: E402
from . import tune   # noqa: E402


def install_as_flute() -> None:
    """Make `import flute` resolve to this package (drop-in for integrations that
    import the reference by name, e.g. transformers' HIGGS support)."""
    import sys
    from . import integrations, nf_utils
    sys.modules.setdefault("flute", sys.modules[__name__])
    for name, mod in (("utils", utils), ("tune", tune), ("ops", ops), ("nf_utils", nf_utils),
                      ("integrations", integrations)):
        sys.modules.setdefault(f"flute.{name}", mod)
    from .integrations import base, higgs
    sys.modules.setdefault("flute.integrations.base", base)
    sys.modules.setdefault("flute.integrations.higgs", higgs)
