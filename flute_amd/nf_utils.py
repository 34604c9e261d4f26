"""NormalFloat tables and the kernel-faithful fake quantizer (flute/nf_utils.py).

Only what the qgemm path needs to build its lookup tables and test fixtures:
`get_values_pivots` (nf_utils.py:14-32, without the hard `.cuda()` of :32),
`nf_quantize` (:46-71) and `nf_quantize_2` (:74-89).
"""
from typing import Optional, Tuple

import torch

# flute/nf_utils.py:29
NF4_VALUES = [
    -1.0, -0.6961928009986877, -0.5250730514526367, -0.39491748809814453,
    -0.28444138169288635, -0.18477343022823334, -0.09105003625154495, 0.0,
    0.07958029955625534, 0.16093020141124725, 0.24611230194568634,
    0.33791524171829224, 0.44070982933044434, 0.5626170039176941,
    0.7229568362236023, 1.0]


def _linspace(start: float, stop: float, num: int) -> torch.Tensor:
    steps = torch.arange(num, dtype=torch.float32) / (num - 1)
    return start + steps * (stop - start)


def get_values_pivots(bits: int = 4, symmetric: bool = False, dtype=torch.float32,
                      device: Optional[torch.device] = None):
    dist = torch.distributions.normal.Normal(torch.tensor(0.0), torch.tensor(1.0))
    offset = 0.5 * (1 / 32 + 1 / 30)
    if symmetric:
        v = dist.icdf(_linspace(offset, 1 - offset, 2 ** bits))
    else:
        v1 = -1 * dist.icdf(_linspace(1 - offset, 0.5, 2 ** (bits - 1)))
        v2 = dist.icdf(_linspace(0.5, 1 - offset, 2 ** (bits - 1) + 1)[1:])
        v = torch.cat((v1, v2))
    v = v / torch.max(torch.abs(v))
    if bits == 4 and not symmetric:
        v = torch.tensor(NF4_VALUES)
    p = (v[1:] + v[:-1]) / 2
    return v.to(dtype=dtype, device=device).clone(), p.to(dtype=dtype, device=device).clone()


def nf_quantize(W: torch.Tensor, num_bits: int, group_size: int,
                custom_scales: Optional[torch.Tensor] = None
                ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """(fake-quantized W, integer codes, absmax scales, table); nf_utils.py:46-71."""
    values, pivots = get_values_pivots(num_bits, False, device=W.device)
    qx = W.reshape(-1, group_size).float()
    absmax = custom_scales.reshape(-1, 1).float() if custom_scales is not None else \
        qx.abs().max(dim=1, keepdim=True).values
    index = torch.searchsorted(pivots, (qx / absmax).contiguous())
    dqx = values[index] * absmax
    return dqx.view(W.size()), index.view(W.size()), absmax.squeeze(), values


def nf_quantize_2(W: torch.Tensor, num_bits: int, group_size: int, dtype: torch.dtype) -> torch.Tensor:
    """Fake quantization with the kernel's rounding (table and scale in `dtype`,
    one product rounding); nf_utils.py:74-89."""
    values, pivots = get_values_pivots(num_bits, False, device=W.device)
    qx = W.reshape(-1, group_size).float()
    absmax = qx.abs().max(dim=1, keepdim=True).values
    index = torch.searchsorted(pivots, (qx / absmax).contiguous())
    dqx = values.to(dtype=dtype)[index] * absmax.to(dtype=dtype)
    return dqx.view(W.size())
