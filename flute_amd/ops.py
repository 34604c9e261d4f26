"""torch operator surface: `flute::qgemm_raw_simple[_hadamard]`.

Schemas are the reference's, verbatim (flute/csrc/qgemm.cpp:251-254); the
implementation is registered for the `CUDA` dispatch key (HIP tensors use it on
PyTorch-ROCm, as qgemm.cpp:257-260 does for CUDA) and forwards to the C ABI.
The fake (meta) implementations restate flute/ops.py:4-83 so that
torch.compile / opcheck see the same validation.  No CPU kernel is registered.
"""
import torch

from . import _lib

_SCHEMA_QGEMM = (
    "qgemm_raw_simple(Tensor input, Tensor weight, Tensor scales, Tensor table, "
    "Tensor table2, Tensor(a!) workspace, int num_bits, int group_size, "
    "int template_id, int num_sms) -> Tensor")
_SCHEMA_QGEMM_HADAMARD = (
    "qgemm_raw_simple_hadamard(Tensor input, Tensor weight, Tensor scales, Tensor table, "
    "Tensor table2, Tensor(a!) workspace, int num_bits, int group_size, "
    "int hadamard_size, int template_id, int num_sms) -> Tensor")

_DEF = torch.library.Library("flute", "DEF")
_DEF.define(_SCHEMA_QGEMM)
_DEF.define(_SCHEMA_QGEMM_HADAMARD)
_IMPL = torch.library.Library("flute", "IMPL", "CUDA")

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


def _qgemm_raw_simple(input, weight, scales, table, table2, workspace, num_bits, group_size,
                      template_id, num_sms, hadamard_size=0):
    _validate(input, weight, scales, table, table2, workspace, num_bits, group_size)
    K = input.shape[-1]
    N = scales.shape[0]
    x2d = input.reshape(-1, K)
    if not x2d.is_contiguous():
        x2d = x2d.contiguous()
    for t in (weight, scales, table, table2, workspace):
        if not t.is_contiguous():
            raise RuntimeError("flute::qgemm_raw_simple: weight/scales/tables/workspace must be contiguous")
        if t.device != input.device:
            raise RuntimeError("flute::qgemm_raw_simple: all tensors must be on the input's device")
    M = x2d.shape[0]
    out = torch.empty((M, N), dtype=input.dtype, device=input.device)
    if M > 0:
        lib = _lib.get()
        scratch = None
        if hadamard_size:
            if hadamard_size < 1 or hadamard_size & (hadamard_size - 1) or hadamard_size > 2 ** 15:
                _lib.check(-8)
            if K % hadamard_size and (M * K) % hadamard_size:
                raise RuntimeError(f"shape {tuple(input.shape)} is invalid for hadamard_size {hadamard_size}")
            # decode-kernel launches rotate the activations while staging them; every other plan
            # (and blocks that span rows) rotates into a scratch tensor first (qgemm.cpp:201-244)
            if not lib.flute_qgemm_hadamard_fused(_DTYPE_ID[input.dtype], num_bits, group_size,
                                                  hadamard_size, M, N, K, template_id, num_sms,
                                                  workspace.numel()):
                scratch = torch.empty_like(x2d)
        with torch.cuda.device(input.device):          # qgemm.cpp:101 OptionalCUDAGuard
            rc = lib.flute_qgemm_hadamard(
                _DTYPE_ID[input.dtype], num_bits, group_size, hadamard_size, M, N, K, weight.shape[0],
                x2d.data_ptr(), weight.data_ptr(), out.data_ptr(), scales.data_ptr(),
                table.data_ptr(), table2.data_ptr(),
                scratch.data_ptr() if scratch is not None else None,
                workspace.data_ptr(), workspace.numel(),
                template_id, num_sms, _stream_ptr(input.device))   # qgemm.cpp:105 current stream
        _lib.check(rc)
    return out.reshape(input.shape[:-1] + (N,))


def _qgemm_raw_simple_hadamard(input, weight, scales, table, table2, workspace, num_bits,
                               group_size, hadamard_size, template_id, num_sms):
    # qgemm.cpp:214-244: rotate, then the plain op - fused into one launch where the plan allows
    if input.dtype not in _DTYPE_ID:
        raise TypeError("Only fp16 and bf16 supported currently")
    return _qgemm_raw_simple(input, weight, scales, table, table2, workspace, num_bits,
                             group_size, template_id, num_sms, hadamard_size=hadamard_size)


_IMPL.impl("qgemm_raw_simple", _qgemm_raw_simple)
_IMPL.impl("qgemm_raw_simple_hadamard", _qgemm_raw_simple_hadamard)


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
