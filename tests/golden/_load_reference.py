"""Load the reference's pure-torch packers by file path (no install, no `_C`).

Test infrastructure only.  Used by make_golden.py (in the build container,
where /root/reference exists) to produce the committed fixtures.  Follows
SURVEY.md Appendix B: stub `jaxtyping` and a stub `flute` package exposing
`qgemm=None` and `TEMPLATE_CONFIGS`.
"""
import importlib.util
import os
import sys
import types

import torch

REF = os.environ.get("FLUTE_REFERENCE", "/root/reference")


def load_reference():
    if not os.path.isdir(REF):
        raise FileNotFoundError(REF)

    class _Any:
        def __class_getitem__(cls, item):
            return cls

    jt = types.ModuleType("jaxtyping")
    for name in ("Float", "UInt8", "Int16", "Int32", "Bool"):
        setattr(jt, name, _Any)
    sys.modules["jaxtyping"] = jt

    pkg = types.ModuleType("flute")
    pkg.__path__ = [os.path.join(REF, "flute")]
    pkg.qgemm = None
    pkg.TEMPLATE_CONFIGS = torch.load(
        os.path.join(REF, "flute/data/qgemm_kernel_raw_generated_configs.pth"),
        weights_only=True)
    sys.modules["flute"] = pkg

    mods = {}
    for name in ("packbits_utils", "utils"):
        spec = importlib.util.spec_from_file_location(
            f"flute.{name}", os.path.join(REF, "flute", f"{name}.py"))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[f"flute.{name}"] = mod
        setattr(pkg, name, mod)
        spec.loader.exec_module(mod)
        mods[name] = mod
    return pkg, mods["utils"], mods["packbits_utils"]
