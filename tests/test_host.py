"""Host-side mirror of the reference interface (flute_amd.utils / tune / ops /
integrations) on CPU: packers against the reference-generated fixtures, fake
(meta) operator implementations, metadata round trips.  No GPU."""
import numpy as np
import pytest
import torch

import flute_amd
from flute_amd import tune, utils
from flute_amd.integrations import higgs
from flute_amd.integrations.base import FluteLinear


def tid_for(bits, tile_p):
    return [t for (b, t), c in sorted(flute_amd.TEMPLATE_CONFIGS.items())
            if b == bits and c["TileP"] == tile_p][0]


def test_pack_matches_reference_fixture(golden):
    if golden.kind != "kernel":
        pytest.skip("HIGGS fixture")
    Q = utils.pack(torch.from_numpy(golden.W), golden.num_bits, [tid_for(golden.num_bits, golden.tile_p)], 256)
    assert Q.dtype == torch.int16 and np.array_equal(Q.numpy(), golden.Q)


def test_qmap2_matches_reference_fixture(golden):
    if golden.kind == "higgs" and golden.vector_size == 2:
        pytest.skip("codebook")
    t2 = utils.make_qmap2_from_qmap(golden.table)
    assert torch.equal(t2.view(torch.int32), golden.table2.view(torch.int32))


def test_pack_errors_and_template_helpers():
    with pytest.raises(NotImplementedError):
        utils._pack_3bit(torch.zeros((64, 1024), dtype=torch.uint8), 64)
    with pytest.raises(ValueError):
        utils.pack(torch.zeros((64, 128), dtype=torch.uint8), 4, [0, 16], 256)    # mixed TileP
    with pytest.raises(OverflowError):
        utils._pack_4bit(torch.full((64, 128), 16, dtype=torch.uint8), 32)
    with pytest.raises(ValueError):
        utils.safe_cast(torch.tensor([1.5]), torch.int64)
    cfg = utils.get_template_config(4, 132, 256)
    assert cfg == {"tileM": 16, "tileK": 64, "tileP": 32, "blocks": 4 * 256}
    assert len(utils.get_template_ids(4)) == 144 and len(utils.get_template_ids(3)) == 36
    assert utils.is_template_supported(1, 4096, 4096, 4, 16, 256)
    assert not utils.is_template_supported(1, 4096, 4096, 3, 0, 256)
    assert len(flute_amd.TEMPLATE_CONFIGS) == 216
    assert set(flute_amd.TEMPLATE_CONFIGS[(4, 0)]) >= {"SMs_Multiple", "Threads", "TileM", "TileK", "TileP", "Stages"}


def _meta_args(M=3, N=256, K=128, bits=4, g=64, dtype=torch.float16, lead=()):
    mk = lambda *s, dt=dtype: torch.empty(*s, dtype=dt, device="meta")
    return [mk(*lead, M, K), mk(bits * N // 16, K, dt=torch.int16), mk(N, K // g), mk(2 ** bits),
            mk(2 ** bits, 2 ** bits, 1, dt=torch.float32), mk(64, dt=torch.uint8), bits, g, 16, 256]


def test_fake_impl_shapes_and_validation():
    # flute/ops.py:4-83
    out = flute_amd.qgemm(*_meta_args())
    assert out.shape == (3, 256) and out.dtype == torch.float16 and out.device.type == "meta"
    out = flute_amd.qgemm(*_meta_args(lead=(2, 5)))
    assert out.shape == (2, 5, 3, 256)
    a = _meta_args(dtype=torch.bfloat16)
    out = flute_amd.qgemm_hadamard(*a[:8], 64, *a[8:])
    assert out.shape == (3, 256) and out.dtype == torch.bfloat16
    bad = _meta_args()
    bad[1] = torch.empty(7, 128, dtype=torch.int16, device="meta")
    with pytest.raises(ValueError):
        flute_amd.qgemm(*bad)
    bad = _meta_args()
    bad[2] = torch.empty(256, 2, dtype=torch.float32, device="meta")
    with pytest.raises(TypeError):
        flute_amd.qgemm(*bad)
    with pytest.raises(TypeError):
        flute_amd.qgemm(*_meta_args(dtype=torch.float32))


def test_no_cpu_kernel_is_registered():
    a = _meta_args()
    cpu = [t.to("cpu") if False else (torch.zeros(t.shape, dtype=t.dtype) if isinstance(t, torch.Tensor) else t) for t in a]
    with pytest.raises((NotImplementedError, RuntimeError)):
        flute_amd.qgemm(*cpu)


def test_tune_metadata_roundtrip_and_key():
    m = tune.TuneMetaData(M=1, N=4096, K=4096, num_bits=4, group_size=64, num_sms=256,
                          dtype=torch.bfloat16, device=torch.device("cuda:0"), template_id=16)
    assert tune.TuneMetaData.from_dict(m.to_dict()) == m
    with pytest.raises(ValueError):
        tune.TuneMetaData.from_dict({**m.to_dict(), "dtype": "torch.int8"})
    assert tune.get_template_key(5, 8, 16, 4, 64, 256, torch.float16) == \
        tune.get_template_key(9, 8, 16, 4, 64, 256, torch.float16)       # 5 <= M <= 16 shares a template
    assert tune.get_template_key(1, 8, 16, 4, 64, 256, torch.float16) != \
        tune.get_template_key(9, 8, 16, 4, 64, 256, torch.float16)       # decode kernel: its own entry
    cands = tune.candidate_templates(1, 4096, 4096, 4, 64, 256, torch.float16)
    assert 1 <= len(cands) < 144
    assert {flute_amd.TEMPLATE_CONFIGS[(4, t)]["TileP"] for t in cands} == {32, 64}


def test_flute_linear_module_surface():
    lin = FluteLinear(128, 256, 4, 64, template_id=16, workspace_lazy_init=True, bias=True,
                      device=torch.device("cpu"), dtype=torch.float16)
    assert lin.weight.shape == (64, 128) and lin.weight.dtype == torch.int16
    assert lin.scales.shape == (256, 2) and lin.tables.shape == (16,)
    assert lin.tables2.shape == (16, 16, 1) and lin.tables2.dtype == torch.float32
    assert lin.get_extra_state() == {"num_bits": 4, "group_size": 64, "template_id": 16}
    lin.set_extra_state({"num_bits": 4, "group_size": 64, "template_id": 16})
    with pytest.raises(ValueError):
        lin.set_extra_state({"num_bits": 3, "group_size": 64, "template_id": 16})
    sd = lin.state_dict()
    assert {"weight", "scales", "tables", "tables2", "bias"} <= set(sd)
    with pytest.raises(NotImplementedError):
        FluteLinear(128, 256, 4, 64, 16, device=torch.device("cpu"), dtype=torch.float32)


def test_higgs_prepare_data_matches_reference_fixture(golden, monkeypatch):
    """integrations/higgs.py against the fixture produced by the reference's own
    prepare_data (tune_and_pack redirected to plain packing, as in make_golden.py)."""
    if golden.kind != "higgs":
        pytest.skip("kernel fixture")

    def fake_tune_and_pack(inputs, weight, num_bits, group_size, **kw):
        tid = tid_for(num_bits, 32)
        return utils.pack(weight, num_bits, [tid], 108), None

    monkeypatch.setattr(tune, "tune_and_pack", fake_tune_and_pack)
    Q, S, qmap, qmap2, _ = higgs.prepare_data_transposed(
        golden.weight_higgs, golden.scales_higgs, golden.grid, golden.num_bits, golden.group_size,
        golden.vector_size, golden.dtype, torch.device("cpu"), example_batch_size=1,
        check_correctness=False)
    assert np.array_equal(Q.numpy(), golden.Q)
    assert torch.equal(S.view(torch.int16), golden.S.view(torch.int16))
    assert torch.equal(qmap2.view(torch.int32), golden.table2.view(torch.int32))


def test_nf_tables():
    from flute_amd import nf_utils
    v, p = nf_utils.get_values_pivots(4)
    assert torch.allclose(v, torch.tensor(nf_utils.NF4_VALUES)) and p.shape == (15,)
    v3, _ = nf_utils.get_values_pivots(3)
    assert v3.shape == (8,) and v3.abs().max() == 1
    W = torch.randn(64, 128)
    dq, idx, absmax, vals = nf_utils.nf_quantize(W, 4, 64)
    assert idx.max() <= 15 and dq.shape == W.shape and absmax.shape == (128,)
    assert nf_utils.nf_quantize_2(W, 4, 64, torch.float16).dtype == torch.float16


def test_install_as_flute_alias():
    flute_amd.install_as_flute()
    import flute
    import flute.tune
    import flute.utils
    assert flute.qgemm is flute_amd.qgemm and flute.tune.TuneMetaData is tune.TuneMetaData


def test_prepare_model_flute_host_checks():
    """flute/integrations/base.py:44-200: the model walker refuses what it cannot quantize before touching a GPU."""
    from flute_amd.integrations.base import prepare_model_flute
    with pytest.raises(ValueError):                              # quantization and tuning run on the layer's GPU
        prepare_model_flute("m", torch.nn.Sequential(torch.nn.Linear(128, 128)).half(), 4, 64, 1)
    with pytest.raises(NotImplementedError):                     # fp32 layers are not quantized (base.py:80-81)
        prepare_model_flute("m", torch.nn.Sequential(torch.nn.Linear(128, 128)), 4, 64, 1)
    with pytest.raises(ValueError):                              # in_features must hold whole groups and 64-k lines
        prepare_model_flute("m", torch.nn.Sequential(torch.nn.Linear(96, 128)).half(), 4, 64, 1)
    # nothing to replace: a no-op
    m = torch.nn.Sequential(torch.nn.LayerNorm(8))
    prepare_model_flute("m", m, 4, 64, 1, fake=True)
    assert isinstance(m[0], torch.nn.LayerNorm)


def test_shipped_tuned_table_is_consistent():
    """flute_amd/data/gfx950_tuned.json (python -m flute_amd.tune on an MI355X): every entry names a template
    the planner accepts for that problem, the headline shape is present, `_tune` answers from it without a GPU."""
    from flute_amd import tune
    table = tune.load_tuned_table()
    assert len(table) > 500
    dt = {"float16": torch.float16, "bfloat16": torch.bfloat16}
    for i, (key, tid) in enumerate(sorted(table.items())):
        if i % 7:
            continue
        M, N, K, bits, g, num_sms, dtype, tile_p = key.split("|")
        M, N, K, bits, g, num_sms, tile_p = map(int, (M, N, K, bits, g, num_sms, tile_p))
        assert utils.is_template_supported(M, N, K, bits, tid, num_sms, g, dt[dtype]), key
        if tile_p:
            assert flute_amd.TEMPLATE_CONFIGS[(bits, tid)]["TileP"] == tile_p, key
    assert tune.lookup_tuned(1, 4096, 4096, 4, 64, 256, torch.float16) is not None
    assert tune.m_bucket(2) == 2 and tune.m_bucket(3) == 4 and tune.m_bucket(5) == 16 and tune.m_bucket(17) == 32 and tune.m_bucket(10 ** 6) == 4096
    # answered from the table: no CUDA device is touched
    tid = tune._tune(1, 4096, 4096, 4, 64, 256, torch.float16, torch.device("cpu"))
    assert tid == tune.lookup_tuned(1, 4096, 4096, 4, 64, 256, torch.float16)


def test_bench_gpus_flag_starts_the_ranks_itself():
    """`python bench.py --gpus 2` with NO launcher around it must come up as two ranks (the flag was parsed and
    never read, so the first multi-GPU record would have been one rank).  FLUTE_BENCH_DRYRUN=1 stops each rank after the
    process-group plumbing (gloo, no GPU here): the line carries the group's size and a sum of ones over the ranks.  And a
    launcher whose rank count disagrees with --gpus is refused."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["FLUTE_BENCH_DRYRUN"] = "1"
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks_in_group"] == 2 and out["ranks_counted_by_allreduce"] == 2
    assert out["self_launched"] is True and out["dry_run"] is True
    env2 = dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    p2 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4"], env=env2, capture_output=True,
                        text=True, timeout=120)
    assert p2.returncode != 0 and "WORLD_SIZE=2" in (p2.stderr + p2.stdout)


def test_every_key_of_the_shipped_table_resolves_to_a_legal_plan():
    """flute_amd/data/gfx950_tuned.json is what the product serves (FluteLinear, tune_and_pack, bench.py): every key's template id
    exists for its bit width, honours the key's TileP constraint, passes is_template_supported and gives flute_qgemm_plan a legal
    plan for the key's shape at the bucket's batch size (the reference's tuner checks every id it stores: tune.py:294-392)."""
    from flute_amd import _lib
    table = tune.load_tuned_table()
    assert len(table) > 5000
    lib = _lib.get()
    dt = {"float16": (0, torch.float16), "bfloat16": (1, torch.bfloat16)}
    fams = {}
    for key, tid in table.items():
        mb, N, K, bits, g, sms, dtype, tile_p = key.split("|")
        mb, N, K, bits, g, sms, tile_p = int(mb), int(N), int(K), int(bits), int(g), int(sms), int(tile_p)
        cfg = flute_amd.TEMPLATE_CONFIGS.get((bits, tid))
        assert cfg is not None, key
        assert tile_p == 0 or cfg["TileP"] == tile_p, key
        assert utils.is_template_supported(mb, N, K, bits, tid, sms, g, dt[dtype][1]), key
        p = _lib.Plan()
        assert lib.flute_qgemm_plan(dt[dtype][0], bits, g, mb, N, K, tid, sms, 64 << 20, p) == 0, key
        assert p.grid >= 1 and p.block in (64 * w for w in range(1, 17)) and p.lds_bytes <= 160 * 1024 and p.workspace_needed <= 64 << 20, key
        fams[p.family] = fams.get(p.family, 0) + 1
    assert set(fams) >= {0, 2, 3, 5, 6, 7}, fams                   # the table reaches every kernel family


def test_persistent_mfma_decode_kernel_activation_ring_layout():
    """qgemm_persistm.h: the LDS-DMA writes a request lane-linearly, so the swizzle sits in what a lane ASKS for - lane l of request r
    fetches chunk (l % 16) ^ 4 (l / 16) ^ g(r) of row 4 r + l / 16 - and the fragment read of MFMA row m = 4 r + mm, k-chunk 4 s + kg looks at
    position 16 mm + 4 (s ^ mm) + (kg ^ g(r)) of request r's KB.  The two agree, and the 16 lanes of every ds_read_b128 lane group
    (MI355X_MICROARCH.md's LDS table) hit 16 different 16-B slots - also when rows past the requested ones alias them (M <= 4 / 8: the
    lanes of a group then share addresses, which broadcast)."""
    g = lambda r: (4 - r) & 3  # noqa: E731
    groups = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
              list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)), list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]
    for r in range(4):
        where = {}
        for lane in range(64):
            mm = lane >> 4
            where[(mm, (lane & 15) ^ (4 * mm) ^ g(r))] = lane
        assert len(where) == 64
        for mm in range(4):
            for s in range(4):
                for kg in range(4):
                    assert where[(mm, 4 * s + kg)] == 16 * mm + 4 * (s ^ mm) + (kg ^ g(r))
    for xr in (1, 2, 4):
        for s in range(4):
            for grp in groups:
                addrs = set()
                for lane in grp:
                    i16, kg = lane & 15, lane >> 4
                    r, mm = (i16 % (4 * xr)) >> 2, i16 & 3
                    addrs.add(r * 1024 + (16 * mm + 4 * (s ^ mm) + (kg ^ g(r))) * 16)
                slots = {(a // 16) % 16 for a in addrs}
                assert len(slots) == len(addrs) and (xr < 4 or len(addrs) == 16), (xr, s, grp)       # distinct addresses never share a slot
