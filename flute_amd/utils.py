"""Host-side helpers with the reference's names and signatures (flute/utils.py).

Packing is the reference's wire format (flute/utils.py:59-253) written as a
closed form over bit fields instead of the reference's bool-tensor pipeline; it
runs on whatever device `W` lives on.  Unpacking of device tensors is a native
HIP kernel (`flute_unpack`), not the identity-matmul trick of utils.py:347-407.
"""
import math
import warnings
from functools import lru_cache
from typing import Dict, List

import torch

from . import _lib

_WORKSPACES = {}


def make_qmap2_from_qmap(qmap: torch.Tensor) -> torch.Tensor:
    """flute/utils.py:15-33: (table[i], table[j]) pairs viewed as float32 [n,n,1]."""
    if qmap.ndim != 1:
        raise ValueError
    if qmap.dtype not in [torch.float16, torch.bfloat16]:
        raise TypeError
    n = qmap.shape[0]
    qmap2 = torch.stack([qmap[:, None].expand(n, n), qmap[None, :].expand(n, n)], dim=-1)
    return qmap2.contiguous().view(dtype=torch.float32)


@lru_cache(maxsize=8)
def get_device_num_sms(device: torch.device) -> int:
    """Compute units of the device (256 on MI355X); flute/utils.py:410-412."""
    return torch.cuda.get_device_properties(device).multi_processor_count


def make_workspace_streamk(device: torch.device) -> torch.Tensor:
    """Zero-filled scratch for the grid-level K split (flute/utils.py:36-45).
    Layout (csrc/api.hip, csrc/xwg.h): bytes [0, 64 KB) hold two state words per
    output tile for the in-launch reductions - zero when a call starts and zero
    again when it ends, as the reference's barrier region
    (tile_scheduler_utils.hpp:196) -, the fp32 split-K slabs follow
    (splitk*M*N*4 bytes for the two-launch form; for the in-launch form
    [splitk][tile] slabs in MFMA-fragment order - 64 / 32 / 16 KB per 128 x 128 /
    64 x 128 / 64 x 64 tile of the split-K block kernel, 128 KB per 128 x 256 block
    of the 3-bit block kernel; the planner only takes splits that fit).  64 MiB replaces the reference's blocks*threads*2048-byte
    formula (537 MB for 256 CUs)."""
    return torch.zeros(64 * 1024 * 1024, dtype=torch.uint8, device=device)


def get_workspace_streamk(device: torch.device) -> torch.Tensor:
    """One workspace per device shared by all layers (flute/utils.py:49-56):
    calls on a device must be stream-ordered."""
    device = torch.device(device)
    if device.type != "cuda":
        warnings.warn(f"Only CUDA devices are supported, but got: {device} ({device.type})")
    if device.type == "cuda" and device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    if device not in _WORKSPACES:
        _WORKSPACES[device] = make_workspace_streamk(device)
    return _WORKSPACES[device]


def safe_cast(tensor: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """flute/utils.py:256-266"""
    if tensor.dtype == dtype:
        return tensor
    tensor_casted = tensor.to(dtype=dtype)
    if not (tensor_casted == tensor).all():
        raise ValueError
    return tensor_casted


# ---------------------------------------------------------------------------
# packing
# ---------------------------------------------------------------------------


def _cols_per_block(num_bits: int, tile_P: int) -> int:
    return tile_P * (16 if num_bits == 3 else 16 // num_bits)


def _fields(W: torch.Tensor, num_bits: int) -> torch.Tensor:
    W = safe_cast(W, torch.uint8).to(torch.int64)
    if W.numel() and int(W.max()) >= (1 << num_bits):
        raise OverflowError          # packbits_utils.py:31-32
    return (W[0::2] << num_bits) | W[1::2]       # high bits: even k (utils.py:77-84)


def _words_to_int16(q32: torch.Tensor, K: int) -> torch.Tensor:
    """[P, K/2] int64 holding uint32 values -> [P, K] int16 (little-endian halves)."""
    lo = q32 & 0xFFFF
    hi = (q32 >> 16) & 0xFFFF
    q16 = torch.stack([lo, hi], dim=-1).reshape(q32.shape[0], K)
    q16 = torch.where(q16 >= 0x8000, q16 - 0x10000, q16)
    return q16.to(torch.int16).contiguous()


def _pack_pow2(W: torch.Tensor, num_bits: int, tile_P: int) -> torch.Tensor:
    K, N = W.shape
    J = 16 // num_bits
    if K % 2 or N % (J * tile_P):
        raise ValueError(f"cannot pack K={K} N={N} for num_bits={num_bits} tile_P={tile_P}")
    f = _fields(W, num_bits).reshape(K // 2, N // (J * tile_P), J, tile_P)
    q32 = torch.zeros((K // 2, N // (J * tile_P), tile_P), dtype=torch.int64, device=W.device)
    for j in range(J):
        q32 |= f[:, :, j, :] << (2 * num_bits * j)
    return _words_to_int16(q32.reshape(K // 2, N // J).T, K)


def _pack_4bit(W: torch.Tensor, tile_P: int) -> torch.Tensor:
    """flute/utils.py:59-91"""
    return _pack_pow2(W, 4, tile_P)


def _pack_2bit(W: torch.Tensor, tile_P: int) -> torch.Tensor:
    """flute/utils.py:94-134"""
    return _pack_pow2(W, 2, tile_P)


def _pack_3bit(W: torch.Tensor, tile_P: int) -> torch.Tensor:
    """flute/utils.py:137-253: three 32-bit planes per (block, t, k-pair)."""
    if tile_P != 32:
        raise NotImplementedError
    K, N = W.shape
    if K % 2 or N % 512:
        raise ValueError(f"cannot pack K={K} N={N} for num_bits=3")
    K2, NB, P1 = K // 2, N // 512, N // 16
    f = _fields(W, 3).reshape(K2, NB, 16, 32)
    planes = [torch.zeros((K2, NB, 32), dtype=torch.int64, device=W.device) for _ in range(3)]
    for j in range(15):
        planes[j % 3] |= f[:, :, j, :] << (6 * (j // 3))
    for s in range(3):
        planes[s] |= ((f[:, :, 15, :] >> (2 * s)) & 3) << 30
    q32 = torch.empty((3 * P1, K2), dtype=torch.int64, device=W.device)
    q32[:P1] = planes[0].reshape(K2, P1).T
    q32[P1:] = torch.stack([planes[1], planes[2]], dim=2).reshape(K2, 2 * P1).T
    return _words_to_int16(q32, K)


def pack(W: torch.Tensor, num_bits: int, template_ids: List[int], num_sms: int) -> torch.Tensor:
    """flute/utils.py:269-299: codes W[K,N] -> Q[P,K] int16 for the templates'
    (common) TileP."""
    if W.ndim != 2:
        raise NotImplementedError
    tile_Ps = [get_template_config(num_bits, t, num_sms)["tileP"] for t in template_ids]
    if len(set(tile_Ps)) != 1:
        raise ValueError
    tile_P = tile_Ps[0]
    if num_bits == 4:
        return _pack_4bit(W, tile_P=tile_P)
    if num_bits == 2:
        return _pack_2bit(W, tile_P=tile_P)
    if num_bits == 3:
        return _pack_3bit(W, tile_P=tile_P)
    raise ValueError


# ---------------------------------------------------------------------------
# template table
# ---------------------------------------------------------------------------


def get_template_config(num_bits: int, template_id: int, num_sms: int) -> Dict:
    """flute/utils.py:302-309"""
    from . import TEMPLATE_CONFIGS
    config = TEMPLATE_CONFIGS[(num_bits, template_id)]
    return {
        "tileM": config["TileM"],
        "tileK": config["TileK"],
        "tileP": config["TileP"],
        "blocks": config["SMs_Multiple"] * num_sms,
    }


def get_template_ids(num_bits: int) -> List[int]:
    """flute/utils.py:312-316"""
    from . import TEMPLATE_CONFIGS
    return [i for b, i in TEMPLATE_CONFIGS.keys() if b == num_bits]


def is_template_supported(M: int, N: int, K: int, num_bits: int, template_id: int,
                          num_sms: int, group_size: int = 64,
                          dtype: torch.dtype = torch.float16) -> bool:
    """flute/utils.py:320-344.  The reference rejects templates with fewer tiles
    than CTAs (a Stream-K constraint); here a template is supported when the
    library can plan a launch for it (layout divisibility, LDS budget)."""
    plan = _lib.Plan()
    rc = _lib.get().flute_qgemm_plan(
        0 if dtype == torch.float16 else 1, num_bits, group_size, M, N, K, template_id,
        num_sms, 1 << 40, plan)
    return rc == 0


def get_plan(M: int, N: int, K: int, num_bits: int, group_size: int, template_id: int,
             num_sms: int, dtype: torch.dtype = torch.float16,
             workspace_bytes: int = 64 * 1024 * 1024) -> Dict:
    """Launch plan the library would use (kernel family, K split, LDS, grid)."""
    plan = _lib.Plan()
    _lib.check(_lib.get().flute_qgemm_plan(
        0 if dtype == torch.float16 else 1, num_bits, group_size, M, N, K, template_id,
        num_sms, workspace_bytes, plan))
    return plan.as_dict()


# ---------------------------------------------------------------------------
# unpacking
# ---------------------------------------------------------------------------


def unpack_codes(weight: torch.Tensor, num_bits: int, template_id: int) -> torch.Tensor:
    """Q[P,K] int16 on the GPU -> integer codes W[K,N] uint8 (native kernel)."""
    if weight.dtype != torch.int16 or weight.ndim != 2:
        raise TypeError
    if not weight.is_cuda:
        raise RuntimeError("flute_amd.utils.unpack_codes needs a GPU tensor (native HIP unpacker)")
    weight = weight.contiguous()
    P, K = weight.shape
    N = P * 16 // num_bits
    W = torch.empty((K, N), dtype=torch.uint8, device=weight.device)
    with torch.cuda.device(weight.device):
        _lib.check(_lib.get().flute_unpack(
            num_bits, template_id, N, K, weight.data_ptr(), W.data_ptr(),
            torch.cuda.current_stream(weight.device).cuda_stream))
    return W


def reconstruct(weight, scales, tables, tables2, workspace, num_bits, group_size,
                template_id, num_sms) -> torch.Tensor:
    """flute/utils.py:347-376: dequantized weight [N, K] via qgemm(I)."""
    from . import qgemm
    inputs = torch.eye(weight.shape[1], dtype=scales.dtype, device=scales.device)
    return qgemm(inputs, weight, scales, tables, tables2, workspace, num_bits, group_size,
                 template_id, num_sms).T


def unpack(weight, scales, workspace, num_bits, group_size, template_id_packed,
           num_sms_packed) -> torch.Tensor:
    """flute/utils.py:379-407: integer codes as a [N, K] tensor of scales.dtype
    (same return convention), computed by the native unpacker."""
    return unpack_codes(weight, num_bits, template_id_packed).T.to(dtype=scales.dtype)
