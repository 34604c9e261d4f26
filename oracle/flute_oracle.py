"""CPU oracle for the flute.qgemm hot path.  TEST INFRASTRUCTURE ONLY.

This module is a restatement, in numpy/torch-CPU, of what the reference
(HanGuo97/flute v0.4.2) computes on its qgemm path.  It is the checker for the
HIP kernels: only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s
`cpu_baseline` leg may import it.  Nothing under `flute_amd/` imports it and the
product path has no CPU fallback.

Pinning status
--------------
* pack / unpack / make_qmap2: PINNED.  `tests/golden/*.npz` were produced by the
  reference's own pure-torch packers (`flute/utils.py:59-253`,
  `flute/utils.py:15-33`) imported in the build container by
  `tests/golden/make_golden.py`; `tests/test_oracle.py` checks this module
  against every fixture bit-for-bit.
* qgemm: PINNED to the reference's ground-truth formula
  (`tests/kernel.py:68-71` == `flute/tune.py:332-335`) evaluated by that same
  script with torch on CPU; the reference CUDA kernel itself cannot be built
  here (no nvcc / CUTLASS / NVIDIA GPU - SURVEY.md 8c).
* vector (HIGGS) dequant: PINNED to `tests/higgs.py:7-17` the same way.
* hadamard: PARITY UNPINNED.  No reference test exercises
  `hadamard_transform` / `qgemm_hadamard`; the oracle is the mathematical
  definition (orthonormal Sylvester Hadamard, `hadamard_transform_cuda.cu:141-154`).
  `hadamard_transform_staged` additionally emulates the reference kernel's STAGED
  arithmetic (one 16x16 factor per tensor-core pass, every pass rounded to T, the
  scale constants rounded to T - `hadamard_transform_cuda.cu:56-73,141-154`), so
  that tests can report how far the HIP kernel (fp32 butterflies, one rounding)
  is from what the CUDA kernel would return.

Notation (the reference's): W[K,N] integer codes, Q[P,K] int16 packed,
S[N,G] scales, table[2^b], table2[2^b,2^b,1] fp32-viewed pair table, X[M,K].
"""
from __future__ import annotations

import numpy as np
import torch

# ---------------------------------------------------------------------------
# packed weight format  (writers: flute/utils.py:59-253; reader:
# flute/csrc/packbits_utils.hpp:92-140 (b=4/2), :322-362 (b=3))
# ---------------------------------------------------------------------------


def _check_shape(K: int, N: int, num_bits: int, tile_p: int) -> None:
    if num_bits not in (2, 3, 4):
        raise ValueError(f"num_bits={num_bits}")
    if tile_p not in (32, 64):
        raise ValueError(f"tile_p={tile_p}")
    if num_bits == 3 and tile_p != 32:
        # flute/utils.py:137-139
        raise NotImplementedError("3-bit packing exists only for tile_P=32")
    cols_per_block = tile_p * 16 if num_bits == 3 else tile_p * (16 // num_bits)
    if K % 2 or N % cols_per_block:
        raise ValueError(f"K={K} N={N} not packable for b={num_bits} tile_P={tile_p}")


def _fields(W: np.ndarray, num_bits: int) -> np.ndarray:
    """[K,N] codes -> [K/2,N] 2b-bit fields: high b bits = W[2k], low = W[2k+1].

    flute/utils.py:77-84: slot 0 takes chunk row 1 (odd k), slot 1 row 0 (even
    k); bits are packed LSB-first (packbits_utils.py:44-47, legacy=False), so the
    odd-k code lands in the low bits of the field.
    """
    W = np.ascontiguousarray(W).astype(np.uint32)
    if W.max(initial=0) >= (1 << num_bits):
        raise OverflowError  # packbits_utils.py:31-32
    return (W[0::2] << num_bits) | W[1::2]


def pack(W: np.ndarray, num_bits: int, tile_p: int) -> np.ndarray:
    """Codes W[K,N] -> Q[P,K] int16, P = num_bits*N/16 (flute/utils.py:269-299)."""
    W = np.asarray(W)
    K, N = W.shape
    _check_shape(K, N, num_bits, tile_p)
    f = _fields(W, num_bits)                      # [K/2, N]
    K2 = K // 2
    if num_bits in (2, 4):
        J = 16 // num_bits
        # n = nb*J*tile_p + j*tile_p + t  ->  row p = nb*tile_p + t, field j
        f = f.reshape(K2, N // (J * tile_p), J, tile_p)
        q32 = np.zeros((K2, N // (J * tile_p), tile_p), np.uint32)
        for j in range(J):
            q32 |= f[:, :, j, :] << np.uint32(2 * num_bits * j)
        q32 = q32.reshape(K2, N // J).T            # [P, K/2]
    else:
        # flute/utils.py:131-253.  n = nb*512 + j*32 + t; j<15: plane j%3,
        # 6-bit slot j//3; j==15: 2 bits in the top of each of the 3 planes.
        NB = N // 512
        f = f.reshape(K2, NB, 16, 32)
        planes = np.zeros((3, K2, NB, 32), np.uint32)
        for j in range(15):
            planes[j % 3] |= f[:, :, j, :] << np.uint32(6 * (j // 3))
        for s in range(3):
            planes[s] |= ((f[:, :, 15, :] >> np.uint32(2 * s)) & np.uint32(3)) << np.uint32(30)
        P1 = N // 16
        q32 = np.zeros((3 * P1, K2), np.uint32)
        q32[:P1] = planes[0].reshape(K2, P1).T
        # second region: per block nb, 32 rows of plane 1 then 32 rows of plane 2
        p12 = np.stack([planes[1], planes[2]], axis=2)        # [K2, NB, 2, 32]
        q32[P1:] = p12.reshape(K2, 2 * P1).T
    q16 = np.ascontiguousarray(q32).view(np.uint16)           # little endian pairs
    return q16.view(np.int16).reshape(q32.shape[0], K)


def unpack(Q: np.ndarray, num_bits: int, tile_p: int) -> np.ndarray:
    """Q[P,K] int16 -> codes W[K,N] uint8.

    The reference has no CPU unpacker (its `unpack` runs the GPU kernel on an
    identity matrix, flute/utils.py:347-407); this inverts `pack` by the layout
    the kernel's `dequantize` reads (packbits_utils.hpp:92-140, 322-362).
    """
    Q = np.ascontiguousarray(Q)
    if Q.dtype != np.int16:
        raise TypeError(Q.dtype)
    P, K = Q.shape
    N = P * 16 // num_bits
    _check_shape(K, N, num_bits, tile_p)
    K2 = K // 2
    q32 = Q.view(np.uint16).astype(np.uint32)
    q32 = q32[:, 0::2] | (q32[:, 1::2] << np.uint32(16))       # [P, K/2]
    mask = np.uint32((1 << (2 * num_bits)) - 1)
    if num_bits in (2, 4):
        J = 16 // num_bits
        q = q32.T.reshape(K2, N // (J * tile_p), tile_p)
        f = np.zeros((K2, N // (J * tile_p), J, tile_p), np.uint32)
        for j in range(J):
            f[:, :, j, :] = (q >> np.uint32(2 * num_bits * j)) & mask
    else:
        NB = N // 512
        P1 = N // 16
        planes = np.zeros((3, K2, NB, 32), np.uint32)
        planes[0] = q32[:P1].T.reshape(K2, NB, 32)
        p12 = q32[P1:].T.reshape(K2, NB, 2, 32)
        planes[1] = p12[:, :, 0]
        planes[2] = p12[:, :, 1]
        f = np.zeros((K2, NB, 16, 32), np.uint32)
        for j in range(15):
            f[:, :, j, :] = (planes[j % 3] >> np.uint32(6 * (j // 3))) & mask
        f[:, :, 15, :] = (((planes[0] >> np.uint32(30)) & 3)
                          | (((planes[1] >> np.uint32(30)) & 3) << np.uint32(2))
                          | (((planes[2] >> np.uint32(30)) & 3) << np.uint32(4)))
    f = f.reshape(K2, N)
    W = np.zeros((K, N), np.uint8)
    W[0::2] = f >> np.uint32(num_bits)
    W[1::2] = f & np.uint32((1 << num_bits) - 1)
    return W


def pair_index(Q: np.ndarray, num_bits: int, tile_p: int) -> np.ndarray:
    """Q -> [K/2, N] pair-table indices (W[2k] << b | W[2k+1]); this is the
    index `dequantize` feeds to table2 (packbits_utils.hpp:105, :343-361)."""
    W = unpack(Q, num_bits, tile_p).astype(np.int64)
    return (W[0::2] << num_bits) | W[1::2]


# ---------------------------------------------------------------------------
# tables
# ---------------------------------------------------------------------------


def make_qmap2_from_qmap(qmap: torch.Tensor) -> torch.Tensor:
    """table[2^b] (fp16/bf16) -> table2[2^b,2^b,1] viewed as float32:
    element (i,j) = (table[i], table[j]) with table[i] in the low half
    (flute/utils.py:15-33)."""
    if qmap.ndim != 1:
        raise ValueError
    if qmap.dtype not in (torch.float16, torch.bfloat16):
        raise TypeError
    n = qmap.shape[0]
    q2 = torch.stack([qmap[:, None].expand(n, n), qmap[None, :].expand(n, n)], dim=-1)
    return q2.contiguous().view(torch.float32)


def table2_as_pairs(table2: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """fp32-viewed pair table -> [4^b, 2] tensor of `dtype`."""
    return table2.contiguous().view(dtype).reshape(-1, 2)


# NormalFloat-4 constants (flute/nf_utils.py:29)
NF4_VALUES = (
    -1.0, -0.6961928009986877, -0.5250730514526367, -0.39491748809814453,
    -0.28444138169288635, -0.18477343022823334, -0.09105003625154495, 0.0,
    0.07958029955625534, 0.16093020141124725, 0.24611230194568634,
    0.33791524171829224, 0.44070982933044434, 0.5626170039176941,
    0.7229568362236023, 1.0)


# ---------------------------------------------------------------------------
# the operator
# ---------------------------------------------------------------------------


def dequantize(Q: np.ndarray, S: torch.Tensor, table2: torch.Tensor,
               num_bits: int, group_size: int, tile_p: int) -> torch.Tensor:
    """W^[K,N] = round_T(table2-lookup * S[n, k//g]) - one rounding, in T
    (packbits_utils.hpp:139 `__hmul2`, :344-361)."""
    dtype = S.dtype
    idx = torch.from_numpy(pair_index(Q, num_bits, tile_p))     # [K/2, N]
    pairs = table2_as_pairs(table2, dtype)                      # [4^b, 2]
    K2, N = idx.shape
    w = pairs[idx]                                              # [K/2, N, 2]
    w = w.permute(0, 2, 1).reshape(2 * K2, N)                   # k = 2*kappa + e
    S_ = torch.repeat_interleave(S, group_size, dim=1).T        # [K, N]
    return w * S_                                               # rounds to T


def qgemm(X: torch.Tensor, Q, S: torch.Tensor, table: torch.Tensor,
          table2: torch.Tensor, num_bits: int, group_size: int, tile_p: int,
          accumulate: str = "fp32") -> torch.Tensor:
    """D[M,N] = X[M,K] @ W^[K,N]   (tests/kernel.py:68-71, tune.py:332-335).

    `table` is unused on purpose: the kernel reads only table2
    (qgemm_kernel.hpp:551-557, packbits_utils.hpp:105), which for HIGGS
    vector_size=2 is an arbitrary codebook (integrations/higgs.py:67-71).

    accumulate="fp32" (default): products of T values summed in fp32, rounded
    to T once - the arithmetic of the MMA (config.hpp:323-325, kMixed).
    accumulate="native": torch.mm in T on CPU, what the reference's test
    evaluates on its GPU (tests/kernel.py:71).
    """
    Qn = Q.numpy() if isinstance(Q, torch.Tensor) else np.asarray(Q)
    Wh = dequantize(Qn, S, table2, num_bits, group_size, tile_p)
    X2 = X.reshape(-1, X.shape[-1])
    if accumulate == "fp32":
        D = (X2.float() @ Wh.float()).to(X.dtype)
    elif accumulate == "native":
        D = torch.mm(X2, Wh)
    else:
        raise ValueError(accumulate)
    return D.reshape(*X.shape[:-1], Wh.shape[1])


def vector_dequantize_higgs(weight_higgs: torch.Tensor, scales_higgs: torch.Tensor,
                            grid: torch.Tensor) -> torch.Tensor:
    """HIGGS ground truth (tests/higgs.py:7-17): [out, in] dequantized weight."""
    group_size = weight_higgs.shape[1] * grid.shape[1] // scales_higgs.shape[1]
    w = grid[weight_higgs.long()]
    w = w.reshape(w.shape[0], -1, group_size) * scales_higgs[..., None]
    return w.reshape(w.shape[0], -1)


# ---------------------------------------------------------------------------
# Hadamard pre-rotation (parity unpinned - see module docstring)
# ---------------------------------------------------------------------------


def hadamard_matrix(n: int) -> torch.Tensor:
    """Sylvester-ordered orthonormal Hadamard matrix H_n / sqrt(n), float64."""
    if n < 1 or n & (n - 1):
        raise ValueError(n)
    H = torch.ones(1, 1, dtype=torch.float64)
    while H.shape[0] < n:
        H = torch.cat([torch.cat([H, H], 1), torch.cat([H, -H], 1)], 0)
    return H / (n ** 0.5)


def hadamard_transform(X: torch.Tensor, had_size: int) -> torch.Tensor:
    """X.reshape(-1, h) @ (H_h/sqrt(h)), back to X's shape and dtype
    (qgemm.cpp:201-211; hadamard_transform.cpp:17-56)."""
    flat = X.reshape(-1, had_size).double()
    return (flat @ hadamard_matrix(had_size)).to(X.dtype).reshape(X.shape)


def hadamard_transform_staged(X: torch.Tensor, had_size: int) -> torch.Tensor:
    """Emulation of the reference kernel's arithmetic (hadamard_transform_cuda.cu).

    HadaCore factors H_h = H_16 (x) ... (x) H_16 (x) H_{2^r} and applies one factor per
    tensor-core pass: `mma.m16n8k16` with the factor matrix held as +-c constants, c =
    2^(-bits/2) ROUNDED TO T (fp16 0x39A8 / bf16 0x3F35 = 0.70703125 for odd `bits`, :141-154),
    fp16: f16 accumulate (:56), bf16: f32 accumulate then cvt.rn.bf16x2 (:62-69) - i.e. every
    pass ends with a rounding to T.  Emulated as: per pass, exact products summed in fp32,
    rounded to T once (an upper bound on the precision of the f16-accumulating mma).  The
    passes act on successive 4-bit digit groups of the index, low digits first; the order does
    not change the mathematical result, only which partial sums are rounded.
    """
    dtype = X.dtype
    log_h = had_size.bit_length() - 1
    if had_size < 1 or (1 << log_h) != had_size:
        raise ValueError(had_size)
    flat = X.reshape(-1, had_size)
    rows = flat.shape[0]
    bits_done = 0
    cur = flat
    factors = [4] * (log_h // 4) + ([log_h % 4] if log_h % 4 else [])
    for fb in factors:
        n = 1 << fb
        c = torch.tensor(2.0 ** (-fb / 2)).to(dtype).float()            # scale constant in T
        Hf = (hadamard_matrix(n) * (n ** 0.5)).float() * c              # +-c entries
        lo = 1 << bits_done
        hi = had_size // (lo * n)
        t = cur.float().reshape(rows, hi, n, lo)
        t = torch.einsum("rhnl,nm->rhml", t, Hf)
        cur = t.reshape(rows, had_size).to(dtype)                       # rounding of this pass
        bits_done += fb
    return cur.reshape(X.shape)


def qgemm_hadamard(X, Q, S, table, table2, num_bits, group_size, hadamard_size, tile_p):
    """qgemm.cpp:214-244: rotate (rounded to T), then qgemm."""
    return qgemm(hadamard_transform(X, hadamard_size), Q, S, table, table2,
                 num_bits, group_size, tile_p)
