"""Generate tests/golden/*.npz from the REFERENCE's own Python code.

Run in the build container only (needs /root/reference); the GPU box and the
test-suite only ever read the committed .npz files.

    python tests/golden/make_golden.py

What is taken from the reference (imported by path, unmodified):
  * flute/utils.py::_pack_4bit/_pack_2bit/_pack_3bit/pack   -> Q
  * flute/utils.py::make_qmap2_from_qmap                    -> table2
  * flute/integrations/higgs.py::prepare_data               -> HIGGS W/qmap2
    (its call to tune_and_pack is redirected to utils.pack with a fixed
    template id because tuning needs the CUDA kernel)
  * tests/kernel.py:68-71 ground-truth formula, evaluated with torch on CPU
  * tests/higgs.py:7-17 vector-dequant ground truth
Every array is stored as raw integer bit patterns so that loading needs numpy
only.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from _load_reference import REF, load_reference  # noqa: E402

pkg, U, PB = load_reference()

NF4 = [-1.0, -0.6961928009986877, -0.5250730514526367, -0.39491748809814453,
       -0.28444138169288635, -0.18477343022823334, -0.09105003625154495, 0.0,
       0.07958029955625534, 0.16093020141124725, 0.24611230194568634,
       0.33791524171829224, 0.44070982933044434, 0.5626170039176941,
       0.7229568362236023, 1.0]   # flute/nf_utils.py:29


def bits16(t: torch.Tensor) -> np.ndarray:
    return t.contiguous().view(torch.int16).numpy().view(np.uint16)


def bits32(t: torch.Tensor) -> np.ndarray:
    return t.contiguous().view(torch.int32).numpy().view(np.uint32)


def first_template_with_tile_p(num_bits: int, tile_p: int) -> int:
    for (b, tid), cfg in sorted(pkg.TEMPLATE_CONFIGS.items()):
        if b == num_bits and cfg["TileP"] == tile_p:
            return tid
    raise KeyError


def kernel_case(name, num_bits, tile_p, group_size, dtype, K, N, M, table_kind, seed):
    torch.manual_seed(seed)
    G = K // group_size
    # tests/kernel.py:43-47 samples [0, 2^b - 1); we include the top code.
    W = torch.randint(0, 2 ** num_bits, (K, N), dtype=torch.int64)
    S = torch.randn((N, G), dtype=dtype)
    A = torch.randn((M, K), dtype=dtype) / 100.0
    if table_kind == "arange":
        qmap = torch.arange(2 ** num_bits, dtype=dtype)
    elif table_kind == "randn":
        qmap = torch.randn(2 ** num_bits, dtype=dtype)
    elif table_kind == "nf4":
        qmap = torch.tensor(NF4, dtype=dtype)
    else:
        raise ValueError
    qmap2 = U.make_qmap2_from_qmap(qmap)
    tid = first_template_with_tile_p(num_bits, tile_p)
    Q = U.pack(W.to(torch.uint8), num_bits, [tid], 108)
    # ground truth, tests/kernel.py:68-71
    W_ = qmap[W]
    S_ = torch.repeat_interleave(S, group_size, dim=1).T
    What = W_ * S_
    D_ = torch.mm(A, What)
    D_identity = torch.mm(torch.eye(K, dtype=dtype), What)
    np.savez_compressed(
        os.path.join(HERE, f"{name}.npz"),
        kind="kernel", num_bits=num_bits, tile_p=tile_p, group_size=group_size,
        dtype=str(dtype).replace("torch.", ""), template_id_ref=tid,
        W=W.to(torch.uint8).numpy(), Q=Q.numpy(), S=bits16(S), A=bits16(A),
        table=bits16(qmap), table2=bits32(qmap2), What=bits16(What),
        D=bits16(D_), D_identity=bits16(D_identity))
    print(name, tuple(Q.shape))


def higgs_case(name, num_bits, vector_size, dtype, K, N, group_size, seed):
    # redirect tune_and_pack (needs the CUDA kernel) to plain packing
    tune = types.ModuleType("flute.tune")

    class _Meta:
        pass

    def _tune_and_pack(inputs, weight, num_bits, group_size, check_correctness=True, **kw):
        tid = first_template_with_tile_p(num_bits, 32)
        Q = U.pack(weight, num_bits, [tid], 108)
        m = _Meta()
        m.template_id, m.num_sms = tid, 108
        return Q, m

    tune.tune_and_pack = _tune_and_pack
    tune.TuneMetaData = _Meta
    sys.modules["flute.tune"] = tune
    pkg.tune = tune
    spec = importlib.util.spec_from_file_location(
        "flute_ref_higgs", os.path.join(REF, "flute/integrations/higgs.py"))
    H = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(H)

    torch.manual_seed(seed)
    num_codes = 2 ** (num_bits * vector_size)
    # tests/higgs.py:73-85
    weight_higgs = torch.randint(0, num_codes, (N, K // vector_size), dtype=torch.uint8)
    scales_higgs = torch.randn((N, K // group_size), dtype=dtype)
    grid = torch.randn((num_codes, vector_size), dtype=dtype)
    Q, S, tables, tables2, meta = H.prepare_data_transposed(
        weight_original=weight_higgs, scales_original=scales_higgs, grid=grid,
        num_bits=num_bits, group_size=group_size, vector_size=vector_size,
        dtype=dtype, device=torch.device("cpu"), example_batch_size=1,
        check_correctness=False)
    # tests/higgs.py:7-17
    gs = weight_higgs.shape[1] * grid.shape[1] // scales_higgs.shape[1]
    w = grid[weight_higgs.int()]
    w = w.reshape(w.shape[0], -1, gs) * scales_higgs[..., None]
    w = w.reshape(w.shape[0], -1)           # [N, K]; qgemm(I) must equal w.T
    np.savez_compressed(
        os.path.join(HERE, f"{name}.npz"),
        kind="higgs", num_bits=num_bits, tile_p=32, group_size=group_size,
        vector_size=vector_size, dtype=str(dtype).replace("torch.", ""),
        weight_higgs=weight_higgs.numpy(), scales_higgs=bits16(scales_higgs),
        grid=bits16(grid), Q=Q.numpy(), S=bits16(S), table=bits16(tables),
        table2=bits32(tables2), D_identity=bits16(w.T.contiguous()))
    print(name, tuple(Q.shape))


if __name__ == "__main__":
    f16, bf16 = torch.float16, torch.bfloat16
    kernel_case("w4_tp32_g64_f16_nf4", 4, 32, 64, f16, 128, 512, 3, "nf4", 0)
    kernel_case("w4_tp64_g64_f16_randn", 4, 64, 64, f16, 128, 512, 5, "randn", 1)
    kernel_case("w4_tp32_g128_bf16_arange", 4, 32, 128, bf16, 256, 256, 1, "arange", 2)
    kernel_case("w2_tp32_g64_f16_randn", 2, 32, 64, f16, 128, 512, 2, "randn", 3)
    kernel_case("w2_tp64_g64_bf16_arange", 2, 64, 64, bf16, 128, 512, 1, "arange", 4)
    kernel_case("w3_tp32_g64_bf16_randn", 3, 32, 64, bf16, 128, 1024, 4, "randn", 5)
    kernel_case("w3_tp32_g64_f16_arange", 3, 32, 64, f16, 64, 512, 1, "arange", 6)
    higgs_case("higgs_w4_v2_f16", 4, 2, f16, 128, 256, 64, 7)
    higgs_case("higgs_w3_v2_bf16", 3, 2, bf16, 128, 512, 64, 8)
    higgs_case("higgs_w2_v1_f16", 2, 1, f16, 128, 256, 64, 9)


def template_map():
    """Reference template id -> TileP (data/qgemm_kernel_raw_generated_configs.pth):
    the gfx950 table must keep this map so reference-packed weights keep their id."""
    import json
    m = {f"{b}:{t}": v["TileP"] for (b, t), v in sorted(pkg.TEMPLATE_CONFIGS.items())}
    json.dump(m, open(os.path.join(HERE, "ref_template_tilep.json"), "w"))


if __name__ == "__main__":
    template_map()
