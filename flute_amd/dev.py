"""Planned launches for the offline tuner, the sweeps and the tests.

`torch.ops.flute.qgemm_raw_simple` has the reference's fixed schema; launch-plan overrides
(`flute_overrides` in include/flute_amd.h) therefore travel through this thin wrapper around the C ABI
entry `flute_qgemm_ex`, per call - the library keeps no tuning state.  Same validation, allocation and
stream handling as the operator (flute_amd/ops.py)."""
from typing import Optional

import torch

from . import _lib
from .ops import _DTYPE_ID, _stream_ptr, _validate

Overrides = _lib.Overrides


def qgemm_planned(input: torch.Tensor, weight: torch.Tensor, scales: torch.Tensor, table: torch.Tensor,
                  table2: torch.Tensor, workspace: torch.Tensor, num_bits: int, group_size: int,
                  template_id: int, num_sms: int, overrides: Optional[Overrides] = None,
                  hadamard_size: int = 0) -> torch.Tensor:
    _validate(input, weight, scales, table, table2, workspace, num_bits, group_size)
    K = input.shape[-1]
    N = scales.shape[0]
    x2d = input.reshape(-1, K).contiguous()
    M = x2d.shape[0]
    out = torch.empty((M, N), dtype=input.dtype, device=input.device)
    if M == 0:
        return out.reshape(input.shape[:-1] + (N,))
    scratch = torch.empty_like(x2d) if hadamard_size else None
    with torch.cuda.device(input.device):
        rc = _lib.get().flute_qgemm_ex(
            _DTYPE_ID[input.dtype], num_bits, group_size, hadamard_size, M, N, K, weight.shape[0],
            x2d.data_ptr(), weight.data_ptr(), out.data_ptr(), scales.data_ptr(), table.data_ptr(),
            table2.data_ptr(), scratch.data_ptr() if scratch is not None else None,
            workspace.data_ptr(), workspace.numel(), template_id, num_sms,
            overrides, _stream_ptr(input.device))
    _lib.check(rc)
    return out.reshape(input.shape[:-1] + (N,))


def get_plan(M: int, N: int, K: int, num_bits: int, group_size: int, template_id: int, num_sms: int,
             dtype: torch.dtype = torch.float16, overrides: Optional[Overrides] = None,
             workspace_bytes: int = 64 * 1024 * 1024) -> dict:
    plan = _lib.Plan()
    _lib.check(_lib.get().flute_qgemm_plan_ex(
        0 if dtype == torch.float16 else 1, num_bits, group_size, M, N, K, template_id, num_sms,
        workspace_bytes, overrides, plan))
    return plan.as_dict()


def overrides_from_tuple(t) -> Optional[Overrides]:
    """(family, m_block, waves, kw, splitk, m_tiles, slabs_per_wave[, ring_depth[, one_shot]]) -> Overrides; None
    when every field is -1 (automatic plan)."""
    t = tuple(int(v) for v in t) + (-1,) * (9 - len(t))
    if all(v == -1 for v in t):
        return None
    return Overrides(*t)
