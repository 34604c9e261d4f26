"""HIGGS -> FLUTE data conversion (flute/integrations/higgs.py:8-126).

vector_size == 2: every uint8 code is a pair of b-bit sub-codes along K and the
pair table IS the codebook (`qmap2 = grid.view(2^b, 2^b, 2)`, higgs.py:67-71) -
the kernel never assumes table2 is an outer product.  vector_size == 1 is the
plain scalar-table case.
"""
from typing import Optional, Tuple

import torch

import flute_amd.tune
import flute_amd.utils


def prepare_data(weight_original: torch.Tensor, scales_original: torch.Tensor, grid: torch.Tensor,
                 num_bits: int, group_size: int, vector_size: int, dtype: torch.dtype,
                 device: torch.device, example_batch_size: Optional[int] = None,
                 check_correctness: bool = True
                 ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor,
                            flute_amd.tune.TuneMetaData]:
    dim0 = int(weight_original.shape[0] * vector_size)
    dim1 = int(weight_original.shape[1])
    if weight_original.ndim != 2 or scales_original.ndim != 2 or grid.ndim != 2:
        raise ValueError
    if scales_original.shape[0] != int(dim0 / group_size) or scales_original.shape[1] != dim1:
        raise ValueError
    if grid.shape[0] != int(2 ** (num_bits * vector_size)) or grid.shape[1] != vector_size:
        raise ValueError
    if weight_original.dtype != torch.uint8 or scales_original.dtype != dtype or grid.dtype != dtype:
        raise TypeError
    if not (weight_original.is_contiguous() and scales_original.is_contiguous()
            and grid.is_contiguous()):
        raise ValueError

    if vector_size == 2:
        if num_bits not in (2, 3, 4):
            raise NotImplementedError
        mask = (1 << num_bits) - 1
        W = torch.stack([(weight_original >> num_bits) & mask, weight_original & mask], dim=1)
        W = W.view(dim0, dim1)
        qmap_size = 2 ** num_bits
        qmap = torch.arange(qmap_size, dtype=dtype, device=device)      # unused by the kernel
        qmap2 = grid.view(qmap_size, qmap_size, vector_size).view(dtype=torch.float32).contiguous()
    elif vector_size == 1:
        W = weight_original
        qmap = grid.squeeze(dim=-1)
        qmap2 = flute_amd.utils.make_qmap2_from_qmap(qmap)
    else:
        raise NotImplementedError

    if example_batch_size is None:
        example_batch_size = 1
    example_inputs = torch.randn(example_batch_size, dim0, dtype=dtype, device=device)
    Q, tune_metadata = flute_amd.tune.tune_and_pack(
        inputs=example_inputs, weight=W.contiguous(), num_bits=num_bits, group_size=group_size,
        check_correctness=check_correctness)
    S = scales_original.T.contiguous()
    return Q, S, qmap, qmap2, tune_metadata


def prepare_data_transposed(weight_original, scales_original, grid, num_bits, group_size,
                            vector_size, dtype, device, example_batch_size=None,
                            check_correctness=True):
    """higgs.py:100-126: inputs given as [dim0, dim1/vector_size] / [dim0, dim1/group_size]."""
    if weight_original.ndim != 2 or scales_original.ndim != 2:
        raise ValueError
    return prepare_data(weight_original.T.contiguous(), scales_original.T.contiguous(), grid,
                        num_bits, group_size, vector_size, dtype, device, example_batch_size,
                        check_correctness)
