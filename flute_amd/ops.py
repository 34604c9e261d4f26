"""torch operator surface: `flute::qgemm_raw_simple[_hadamard]`.

Schemas are the reference's, verbatim (flute/csrc/qgemm.cpp:251-254); the
implementation is registered for the `CUDA` dispatch key (HIP tensors use it on
PyTorch-ROCm, as qgemm.cpp:257-260 does for CUDA) by a compiled C++ binding and
forwards to the C ABI.  The fake (meta) implementations restate flute/ops.py:4-83
so that torch.compile / opcheck see the same validation.  No CPU kernel is registered.
"""
import os

import torch

from . import _lib

# The operators are DEFINED and IMPLEMENTED (dispatch key CUDA) by the compiled binding
# flute_amd/csrc/torch_binding.cpp (TORCH_LIBRARY(flute) - the role of flute/csrc/qgemm.cpp:246-260):
# validation, flatten, at::empty, device guard, current stream and ONE call into the C ABI, all in C++.
# This module loads it, fails loudly when it has not been built, and adds the fake (meta) implementations.
TORCH_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libflute_amd_torch.so")
_lib.get()                      # libflute_amd.so first: the binding links against it
if not os.path.exists(TORCH_LIB_PATH):
    raise ImportError(
        f"flute_amd: torch binding not built: {TORCH_LIB_PATH} is missing. Run "
        f"`python -c 'import __graft_entry__ as g; g.build()'` or `make -C {os.path.dirname(TORCH_LIB_PATH)} -j`. "
        f"There is no Python fallback for the operators.")
torch.ops.load_library(TORCH_LIB_PATH)

_DTYPE_ID = {torch.float16: 0, torch.bfloat16: 1}


def _validate(input, weight, scales, table, table2, workspace, num_bits, group_size):
    # flute/ops.py:17-49 (the reference validates only in the fake impl; the
    # real one trusts raw pointers, qgemm.cpp:71-77 - we check in both)
    if not all([input.ndim >= 2, weight.ndim == 2, scales.ndim == 2, table.ndim == 1,
                table2.ndim == 3, workspace.ndim == 1]):
        raise ValueError
    dtype = input.dtype
    if dtype not in _DTYPE_ID:
        raise TypeError
    if not all([weight.dtype == torch.int16, scales.dtype == dtype, table.dtype == dtype,
                table2.dtype == torch.float32, workspace.dtype == torch.uint8]):
        raise TypeError
    if not all([
        weight.shape[1] == input.shape[-1],
        weight.shape[1] == scales.shape[1] * group_size,
        weight.shape[0] == int(num_bits * (scales.shape[0] / 16)),
        table.shape[0] == 2 ** num_bits,
        table2.shape[0] == 2 ** num_bits,
        table2.shape[1] == 2 ** num_bits,
        table2.shape[2] == 1,
    ]):
        raise ValueError


def _stream_ptr(device):
    return torch.cuda.current_stream(device).cuda_stream


def hadamard_transform(input: torch.Tensor, hadamard_size: int) -> torch.Tensor:
    """apply_hadamard (qgemm.cpp:201-211): out-of-place FWHT over
    input.reshape(-1, hadamard_size), orthonormal."""
    if input.dtype not in _DTYPE_ID:
        raise TypeError("Only fp16 and bf16 supported currently")
    if not input.is_cuda:
        raise RuntimeError("flute_amd.hadamard_transform: tensor must live on the GPU")
    if input.shape[-1] % hadamard_size and input.numel() % hadamard_size:
        raise RuntimeError(f"shape {tuple(input.shape)} is invalid for hadamard_size {hadamard_size}")
    x = input.contiguous()
    out = torch.empty_like(x)
    with torch.cuda.device(x.device):
        _lib.check(_lib.get().flute_hadamard(
            _DTYPE_ID[x.dtype], x.data_ptr(), out.data_ptr(), x.numel(), hadamard_size,
            _stream_ptr(x.device)))
    return out.view(input.shape)


@torch.library.register_fake("flute::qgemm_raw_simple")
def _qgemm_raw_simple_abstract(input, weight, scales, table, table2, workspace, num_bits,
                               group_size, template_id, num_sms):
    _validate(input, weight, scales, table, table2, workspace, num_bits, group_size)
    N = scales.shape[0]
    return torch.empty(input.shape[:-1] + (N,), dtype=input.dtype, device=input.device)


@torch.library.register_fake("flute::qgemm_raw_simple_hadamard")
def _qgemm_raw_simple_hadamard_abstract(input, weight, scales, table, table2, workspace,
                                        num_bits, group_size, hadamard_size, template_id,
                                        num_sms):
    return _qgemm_raw_simple_abstract(input, weight, scales, table, table2, workspace,
                                      num_bits, group_size, template_id, num_sms)
