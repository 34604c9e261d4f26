"""ctypes binding of the C-ABI library (include/flute_amd.h).

The HIP library is the product: there is no Python/CPU fallback for any device
entry point.  If `libflute_amd.so` is missing this module raises at import of
the first symbol, loudly, with the build command.
"""
import ctypes
import os
from ctypes import c_char_p, c_int, c_size_t, c_uint32, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# FLUTE_AMD_LIB: development builds of the same ABI (e.g. the phase-timestamp build)
LIB_PATH = os.environ.get("FLUTE_AMD_LIB") or os.path.join(_HERE, "csrc", "libflute_amd.so")


class TemplateInfo(ctypes.Structure):
    _fields_ = [(n, c_int) for n in (
        "num_bits", "template_id", "sms_multiple", "threads", "tile_m", "tile_k",
        "tile_p", "stages", "lut_copies")]


class Plan(ctypes.Structure):
    _fields_ = [
        ("family", c_int), ("m_block", c_int), ("m_tiles", c_int), ("slabs_per_wave", c_int), ("waves", c_int),
        ("kw", c_int),
        ("splitk", c_int), ("k_per_split", c_int), ("lut_copies", c_int),
        ("grid", ctypes.c_uint), ("block", ctypes.c_uint),
        ("lds_bytes", c_size_t), ("workspace_needed", c_size_t),
        ("ring_depth", c_int), ("visits", c_int), ("k_chunks", c_int), ("one_shot", c_int), ("splitk_mode", c_int)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class Overrides(ctypes.Structure):
    """Per-call launch-plan overrides (include/flute_amd.h `flute_overrides`); -1 = automatic."""
    _fields_ = [(n, c_int) for n in (
        "family", "m_block", "waves", "kw", "splitk", "m_tiles", "slabs_per_wave", "ring_depth", "one_shot")]

    def __init__(self, family=-1, m_block=-1, waves=-1, kw=-1, splitk=-1, m_tiles=-1, slabs_per_wave=-1,
                 ring_depth=-1, one_shot=-1):
        super().__init__(family, m_block, waves, kw, splitk, m_tiles, slabs_per_wave, ring_depth, one_shot)


# every symbol include/flute_amd.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "flute_abi_version": (c_int, []),
    "flute_strerror": (c_char_p, [c_int]),
    "flute_num_templates": (c_int, [c_int]),
    "flute_get_template_info": (c_int, [c_int, c_int, ctypes.POINTER(TemplateInfo)]),
    "flute_qgemm_plan": (c_int, [c_int] * 8 + [c_size_t, ctypes.POINTER(Plan)]),
    "flute_qgemm_plan_ex": (c_int, [c_int] * 8 + [c_size_t, ctypes.POINTER(Overrides), ctypes.POINTER(Plan)]),
    "flute_qgemm_ex": (c_int, [c_int] * 8 + [c_void_p] * 8 + [c_size_t, c_int, c_int, ctypes.POINTER(Overrides),
                               c_void_p]),
    "flute_qgemm": (c_int, [c_int] * 7 + [c_void_p] * 7 + [c_size_t, c_int, c_int, c_void_p]),
    "flute_qgemm_hadamard": (c_int, [c_int] * 8 + [c_void_p] * 8 + [c_size_t, c_int, c_int, c_void_p]),
    "flute_qgemm_hadamard_fused": (c_int, [c_int] * 9 + [c_size_t]),
    "flute_hadamard": (c_int, [c_int, c_void_p, c_void_p, c_size_t, c_uint32, c_void_p]),
    "flute_unpack": (c_int, [c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "flute_debug_stream_read": (c_int, [c_void_p, c_void_p, c_size_t, c_int, c_int, c_int, c_void_p]),
    "flute_debug_timestamp": (c_int, [c_void_p, c_void_p]),
}

_lib = None


def get() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"flute_amd: HIP library not built: {LIB_PATH} is missing. "
                f"Run `python -c 'import __graft_entry__ as g; g.build()'` or "
                f"`make -C {os.path.join(_HERE, 'csrc')} -j`. There is no fallback path.")
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)   # AttributeError if the ABI is incomplete
            fn.restype = res
            fn.argtypes = args
        if lib.flute_abi_version() != 8:
            raise ImportError("flute_amd: ABI version mismatch, rebuild libflute_amd.so")
        _lib = lib
    return _lib


def strerror(code: int) -> str:
    return get().flute_strerror(code).decode()


def check(code: int) -> None:
    """Error convention of the reference: RuntimeError with its message prefixes
    (AT_ERROR in qgemm.cpp:153,171 / qgemm_kernel_raw_generated.cu:205)."""
    if code != 0:
        raise RuntimeError(strerror(code))
