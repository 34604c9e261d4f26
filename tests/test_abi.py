"""C-ABI library: loads, exports every symbol include/flute_amd.h declares, and its
host-side logic (template table, launch planning, error codes) behaves.  No GPU."""
import ctypes
import json
import os
import re

import pytest

from flute_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_all_exported():
    hdr = open(os.path.join(ROOT, "include", "flute_amd.h")).read()
    declared = set(re.findall(r"\b(flute_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"flute_status", "flute_dtype"}
    assert declared, "no declarations found"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in flute_amd.h but not exported"
    assert declared == set(_lib.SYMBOLS), (declared ^ set(_lib.SYMBOLS))


def test_template_table_keeps_reference_tilep_map():
    ref = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_template_tilep.json")))
    lib = _lib.get()
    assert lib.flute_num_templates(4) == 144 and lib.flute_num_templates(3) == 36
    assert lib.flute_num_templates(2) == 36 and lib.flute_num_templates(5) == 0
    n = 0
    for key, tile_p in ref.items():
        b, t = map(int, key.split(":"))
        info = _lib.TemplateInfo()
        assert lib.flute_get_template_info(b, t, info) == 0
        assert info.tile_p == tile_p, (b, t)
        assert info.tile_k == 64 and info.sms_multiple in (1, 2, 4)
        n += 1
    assert n == 216
    info = _lib.TemplateInfo()
    assert lib.flute_get_template_info(4, 144, info) == -3
    assert lib.flute_get_template_info(7, 0, info) == -1


def test_error_messages_match_reference_prefixes():
    # flute/tune.py:160-167 string-matches these
    assert _lib.strerror(-3).startswith("Unsupported template_id value")
    assert _lib.strerror(-1).startswith("Unsupported num_bits value")
    assert _lib.strerror(-2).startswith("Unsupported group_size value")
    assert _lib.strerror(-6).startswith("CUDA error: invalid argument")   # flute/tune.py:160
    with pytest.raises(RuntimeError, match="Unsupported template_id value"):
        _lib.check(-3)


def plan(M, N, K, bits=4, g=64, tid=16, sms=256, ws=64 << 20, dtype=0, ovr=None):
    p = _lib.Plan()
    if ovr is not None:
        return _lib.get().flute_qgemm_plan_ex(dtype, bits, g, M, N, K, tid, sms, ws, ovr, p), p
    rc = _lib.get().flute_qgemm_plan(dtype, bits, g, M, N, K, tid, sms, ws, p)
    return rc, p


def test_plan_families_and_invariants():
    for bits, tid in ((4, 16), (2, 4), (3, 4)):
        for M in (1, 2, 3, 4, 5, 8, 16, 17, 64, 256, 1000):
            rc, p = plan(M, 4096, 4096, bits=bits, tid=tid)
            assert rc == 0
            want = 0 if M <= 2 else 2                # decode kernel for M <= 2 (its 4-row variant on request), MFMA kernel beyond
            if M in (3, 4):
                want = 0                             # small layers (4096^2 = 16 M weights): the four-row decode kernels are faster than their MFMA plans (3 bits: round 2; 2 / 4 bits: round 4)
            if bits == 3 and M == 1000:
                want = 3                             # 3 bits: the per-wave kernel is slow enough that 128 blocks already win
            if bits != 3 and M == 1000:
                want = 6                             # 2 / 4 bits: 8 x 32 tiles of 128 x 128, one per CU (qgemm_splitk.h, round 4)
            if bits != 3 and M == 256:
                want = 6                             # ... and 4 x 64 tiles of 64 x 64 over all of K (four K parts per workgroup, round 6)
            if bits != 3 and M == 64:
                want = 6                             # ... x 4 K slices at M = 64 (from M = 33; 2 bits: from 65 until round 6)
            if bits == 4 and 5 <= M <= 16:
                want = 7                             # 4 bits, K = 4096, one round of 4-unit workgroups: the lean MFMA decode kernel (round 5)
            if bits == 4 and 5 <= M <= 8:
                want = 8                             # ... up to eight rows the persistent MFMA decode kernel with the activations resident in LDS (round 6: 5.5 against 5.9 us)
            if bits == 2 and 3 <= M <= 16:
                want = 8                             # 2 bits: the persistent MFMA decode kernel's 2-bit member from three rows (round 6: M = 4 8.4 -> 5.0 us, M = 16 8.6 -> 6.1)
            assert p.family == want, (bits, M, p.family)    # (N = 4096: too few output blocks for the 2- / 4-bit block kernels)
            if p.family == 7:
                assert (p.grid, p.waves, p.block, p.lds_bytes, p.splitk, p.workspace_needed) == (256, 8, 512, 32768 + 32 * 4096, 1, 0)
                continue
            if p.family == 6:
                assert (p.grid, p.block) == (256, 768) and (p.splitk, p.workspace_needed) in ((1, 0), (4, 4 * 64 * 16384 + 65536))      # 8 compute + 4 loader waves
                continue
            if p.family == 2:
                assert p.m_block in (1, 2, 4) and p.m_tiles in (1, 2, 4) and p.slabs_per_wave in (1, 2)
            assert p.grid >= 1 and p.block % 64 == 0 and 64 <= p.block <= 1024
            assert p.lds_bytes <= 160 * 1024
            assert p.waves * 64 == p.block and p.waves % p.kw == 0
            if p.splitk > 1:
                rows = M
                if p.family == 3 and bits == 3 and p.splitk_mode == 1:      # slabs of whole blocks, fragment order (xwg.h; round 5)
                    assert p.m_block == 5 and p.splitk in (2, 4)
                    rows = -(-M // 128) * 128
                assert p.workspace_needed == p.splitk * rows * 4096 * 4 + (64 << 10) <= 64 << 20
                assert p.k_per_split * p.splitk >= 4096 and p.k_per_split % 64 == 0
            else:
                assert p.workspace_needed == 0 and p.k_per_split == 4096
    # M = 3, 4: the MFMA kernel by default; the four-row decode variant on request (override family 0) and for
    # small layers called with a Hadamard size (the rotation stays fused)
    lib = _lib.get()
    p = _lib.Plan()
    assert lib.flute_qgemm_plan_ex(0, 4, 64, 4, 8192, 4096, 16, 256, 64 << 20, None, p) == 0 and p.family != 0    # 32 M weights: an MFMA kernel
    assert lib.flute_qgemm_plan_ex(0, 4, 64, 4, 8192, 4096, 16, 256, 64 << 20, _lib.Overrides(family=0), p) == 0
    assert p.family == 0 and p.m_block == 4
    # the rotation is fused into a decode plan only while it is cheaper than a separate launch: M x K <= 8192 elements
    # (every workgroup rotates all rows for itself; measured in round 4, profiles/r04/hadamard_fused_vs_separate.json)
    assert lib.flute_qgemm_hadamard_fused(0, 4, 64, 512, 1, 4096, 3584, 16, 256, 64 << 20) == 1
    assert lib.flute_qgemm_hadamard_fused(0, 4, 64, 512, 2, 4096, 3584, 16, 256, 64 << 20) == 1
    assert lib.flute_qgemm_hadamard_fused(0, 4, 64, 512, 4, 4096, 3584, 16, 256, 64 << 20) == 0
    assert lib.flute_qgemm_hadamard_fused(0, 4, 64, 512, 4, 28672, 8192, 16, 256, 64 << 20) == 0
    assert lib.flute_qgemm_hadamard_fused(0, 4, 64, 512, 1, 28672, 8192, 16, 256, 64 << 20) == 1
    assert lib.flute_qgemm_hadamard_fused(0, 4, 64, 512, 2, 28672, 8192, 16, 256, 64 << 20) == 0
    assert lib.flute_qgemm_hadamard_fused(1, 3, 64, 512, 2, 4096, 4096, 4, 256, 64 << 20) == 1
    # no workspace -> never a grid-level K split
    for M in (1, 16, 64):
        rc, p = plan(M, 512, 16384, ws=0)
        assert rc == 0 and p.splitk == 1
    # block-tiled prefill kernel: 4-bit, enough 256 x 256 blocks for the chip, no K split
    rc, p = plan(4096, 4096, 4096)
    assert rc == 0 and p.family == 3 and p.grid == 256 and p.block == 512 and p.splitk == 1 and p.lds_bytes <= 160 * 1024
    assert p.m_block == 4                               # 256-row blocks, 1 x 8 wave split (qgemm_block2.h)
    rc, p = plan(2048, 4096, 4096)
    assert rc == 0 and p.family == 3 and p.m_block == 5 and p.grid == 256      # 128-row blocks: one per CU
    rc, p = plan(1024, 4096, 4096)
    assert rc == 0 and p.family == 6 and p.grid == 256 and p.splitk == 1      # too few 128 / 256 x 256 blocks: 128 x 128 tiles (round 4; the per-wave kernel before)
    rc, p = plan(256, 4096, 4096)
    assert rc == 0 and p.family == 6 and (p.grid, p.splitk, p.kw, p.m_tiles, p.workspace_needed) == (256, 1, 4, 4, 0)   # round 6: 64 x 64 tiles over all of K, four K parts per workgroup, no seam (rounds 4 / 5: the per-wave kernel - four K slices per 128 x 128 tile cost more than they saved)
    rc, p = plan(256, 11008, 4096)
    assert rc == 0 and p.family == 6 and (p.grid, p.splitk) == (172, 1)
    rc, p = plan(256, 8192, 8192)
    assert rc == 0 and p.family == 6 and (p.grid, p.splitk, p.splitk_mode) == (256, 2, 1) and p.workspace_needed == 2 * 128 * 65536 + 65536
    rc, p = plan(256, 8192, 8192, ws=1 << 20)
    assert rc == 0 and p.family != 6 or p.splitk == 1   # no room for the slabs: no in-launch split
    rc, p = plan(4096, 4096, 4096, bits=2, tid=0)
    assert rc == 0 and p.family == 3 and p.m_block == 4 and p.lds_bytes <= 160 * 1024      # 2-bit layers too
    rc, p = plan(4096, 4096, 4096, bits=3, tid=4)
    assert rc == 0 and p.family == 3 and p.m_block == 4 and p.grid == 256 and p.lds_bytes == 146 * 1024    # 3 bits: 256-row blocks too (planes 2, 3 of two waves in LDS)
    rc, p = plan(1024, 4096, 4096, bits=3, tid=4)
    assert rc == 0 and p.family == 3 and p.m_block == 5 and (p.grid, p.splitk) == (256, 2) and p.lds_bytes == 80 * 1024     # fewer blocks: 128 rows x two K slices (round 4)
    assert p.workspace_needed == 2 * 1024 * 4096 * 4 + 65536
    rc, p = plan(1024, 4096, 4096, bits=3, tid=4, ws=1 << 20)
    assert rc == 0 and p.splitk == 1                    # no room for the slabs: whole-K blocks
    rc, p = plan(128, 28672, 8192, bits=3, tid=4, dtype=1)
    assert rc == 0 and (p.family, p.m_block, p.splitk, p.grid) == (3, 5, 2, 224)      # 3 bits from M = 65: blocks x K slices against the per-wave kernel
    rc, p = plan(256, 4096, 4096, bits=3, tid=4, dtype=1)
    assert rc == 0 and p.family == 2                    # ... which keeps the small layers (30.5 us against 32.5)
    # 4-bit layers below M = 128 (round 4): 64-row tiles x K slices where the model beats the per-wave kernel, one round of workgroups
    rc, p = plan(64, 8192, 8192)
    assert rc == 0 and (p.family, p.m_tiles, p.kw, p.splitk, p.grid, p.splitk_mode) == (6, 4, 4, 2, 256, 1)     # round 6: 64 x 64 tiles x 2 slices (19.7 us; 64 x 128 x 4 slices 20.5)
    rc, p = plan(64, 28672, 8192)
    assert rc == 0 and p.family == 2                    # two rounds of tiles: per-wave kernel (two slabs per wave)
    rc, p = plan(33, 8192, 8192)
    assert rc == 0 and (p.family, p.m_tiles, p.kw, p.splitk) == (6, 4, 4, 2)
    rc, p = plan(32, 8192, 8192)
    assert rc == 0 and p.family == 2
    rc, p = plan(96, 8192, 8192, bits=2, tid=0)
    assert rc == 0 and (p.family, p.m_tiles, p.splitk, p.grid) == (6, 4, 2, 256)     # 2-bit layers from M = 65
    rc, p = plan(64, 8192, 8192, bits=2, tid=0)                   # round 6: 2-bit layers from M = 33 too (64 x 64 tiles: 20.8 -> 17.9 us)
    assert rc == 0 and (p.family, p.m_tiles, p.kw, p.splitk, p.grid) == (6, 4, 4, 2, 256)
    rc, p = plan(32, 8192, 8192, bits=2, tid=0)
    assert rc == 0 and p.family == 2
    rc, p = plan(64, 8192, 8192, tid=17)
    assert rc == 0 and p.family == 2                    # a tuned id (QuantMapMode digit 1) keeps the kernel it was timed on
    rc, p = plan(300, 1024, 4096, bits=3, tid=4)
    assert rc == 0 and p.family == 2                    # too few blocks: the per-wave MFMA kernel
    # decode kernel: planner shapes (any wave count), one-shot variant for single-visit launches
    rc, p = plan(2, 28672, 8192, tid=19)                     # QuantMapMode digit 3: the ring kernel
    assert rc == 0 and p.family == 0 and p.waves == 14 and p.kw == 1 and p.visits == 2 and p.one_shot == 0
    rc, p = plan(2, 28672, 8192)                             # two rows of a big layer: the persistent one-shot kernel as well
    assert rc == 0 and p.family == 0 and p.one_shot == 3 and p.m_block == 2 and (p.waves, p.grid, p.visits) == (7, 256, 4)
    rc, p = plan(2, 8192, 28672)                             # ... unless the two rows' activations do not fit LDS beside the table
    assert rc == 0 and p.family == 0 and p.one_shot == 0
    # one row on layers of >= 64 M weights: the persistent one-shot kernel (qgemm_persist.h), ~8 waves per CU, every
    # unit slot used: 7168 unit rows = 256 workgroups x 7 waves x 4 visits, K = 4 segments of 4 pieces
    rc, p = plan(1, 28672, 8192)
    assert rc == 0 and p.family == 0 and p.one_shot == 3 and (p.waves, p.kw, p.grid, p.visits, p.k_chunks, p.ring_depth) == (7, 1, 256, 4, 4, 4)
    assert p.lds_bytes <= 80 * 1024
    rc, p = plan(1, 8192, 28672)
    assert rc == 0 and p.one_shot == 3 and (p.waves, p.grid, p.visits, p.k_chunks) == (8, 256, 1, 14)
    rc, p = plan(1, 28672, 8192, bits=3, tid=4)
    assert rc == 0 and p.one_shot == 3 and p.ring_depth == 2 and p.grid * p.waves * p.visits >= 28672 // 16
    rc, p = plan(1, 8192, 8192, bits=3, tid=4)               # 512 unit rows: too few for a wave per row
    assert rc == 0 and p.one_shot != 3
    # layers up to 64 M weights: the one-shot kernels.  4096^2 has 8 pieces per unit row: a wave takes all 8 (no cross-wave
    # reduction).  Round 5: one row of a 4-bit layer with K = 2048 / 4096 / 8192 takes the lean kernel (qgemm_fast.h,
    # one_shot 4) under the automatic ids (QuantMapMode digit 0, Stages 2 / 3: first / second shape); the other digits keep the
    # round-4 one-shot kernel they were tuned on (digit 1: 4 pieces per wave, software-pipelined loop, one_shot 2)
    rc, p = plan(1, 4096, 4096)
    assert rc == 0 and p.family == 0 and p.one_shot == 4 and p.visits == 1 and p.waves % p.kw == 0
    assert (p.waves, p.kw, p.ring_depth, p.grid, p.block) == (4, 1, 8, 256, 256) and p.lds_bytes <= 80 * 1024
    rc, p = plan(1, 4096, 4096, tid=20)
    assert rc == 0 and p.one_shot == 4 and (p.waves, p.kw, p.ring_depth, p.grid, p.block) == (8, 2, 4, 256, 512)
    rc, p = plan(1, 4096, 4096, tid=18)
    assert rc == 0 and p.family == 0 and p.one_shot == 2 and (p.waves, p.kw, p.ring_depth, p.grid, p.block) == (4, 1, 8, 256, 256)
    rc, p = plan(1, 4096, 4096, bits=2, tid=0)
    assert rc == 0 and p.one_shot in (1, 2)
    lib = _lib.get()
    q = _lib.Plan()
    for (M, bits, g, N, K, tid) in ((1, 4, 64, 4096, 4480, 16), (2, 4, 128, 11008, 4096, 0), (4, 2, 64, 4096, 4096, 4),
                                    (1, 3, 64, 8192, 8192, 4), (2, 3, 128, 4096, 4096, 4), (1, 4, 256, 3584, 8192, 16)):
        assert lib.flute_qgemm_plan_ex(0, bits, g, M, N, K, tid, 256, 64 << 20, _lib.Overrides(family=0, one_shot=1), q) == 0
        J = 16 if bits == 3 else 16 // bits
        pieces = -(-K // 512)
        assert q.one_shot >= 1 and q.splitk == 1 and q.waves % q.kw == 0 and q.block == q.waves * 64
        assert -(-pieces // q.kw) <= q.ring_depth and q.grid == -(-(N // J) // (q.waves // q.kw))
        assert q.lds_bytes <= 160 * 1024
        if q.one_shot == 2:                                   # pipelined loop: every wave holds ring_depth pieces
            assert M == 1 and pieces == q.kw * q.ring_depth and (N // J) % (q.waves // q.kw) == 0
    # one_shot = 0 / an explicit ring depth / group size 32 / an odd number of groups (the one-shot kernel reads scale
    # rows as aligned dwords): the persistent ring kernel
    assert lib.flute_qgemm_plan_ex(0, 4, 64, 1, 4096, 4416, 16, 256, 64 << 20, _lib.Overrides(one_shot=1), q) == 0 and q.one_shot == 0
    assert lib.flute_qgemm_plan_ex(0, 4, 64, 1, 4096, 4096, 16, 256, 64 << 20, _lib.Overrides(one_shot=0), q) == 0 and q.one_shot == 0
    assert lib.flute_qgemm_plan_ex(0, 4, 64, 1, 4096, 4096, 16, 256, 64 << 20, _lib.Overrides(ring_depth=4), q) == 0 and q.one_shot == 0
    assert lib.flute_qgemm_plan_ex(0, 4, 32, 1, 4096, 4096, 16, 256, 64 << 20, None, q) == 0 and q.one_shot == 0
    # persistent one-shot kernel by override; not for three rows / ragged K / odd group counts (ring kernel instead)
    assert lib.flute_qgemm_plan_ex(0, 4, 64, 1, 4096, 4096, 16, 256, 64 << 20, _lib.Overrides(one_shot=2), q) == 0 and q.one_shot == 3
    assert q.grid * q.waves * q.visits >= 1024 and q.k_chunks * q.ring_depth == 8
    assert lib.flute_qgemm_plan_ex(0, 4, 64, 2, 4096, 4096, 16, 256, 64 << 20, _lib.Overrides(one_shot=2), q) == 0 and q.one_shot == 3 and q.m_block == 2
    assert lib.flute_qgemm_plan_ex(0, 4, 64, 3, 4096, 4096, 16, 256, 64 << 20, _lib.Overrides(family=0, one_shot=2), q) == 0 and q.one_shot == 0
    assert lib.flute_qgemm_plan_ex(0, 4, 64, 1, 4096, 4352, 16, 256, 64 << 20, _lib.Overrides(one_shot=2), q) == 0 and q.one_shot == 0
    assert lib.flute_qgemm_plan_ex(0, 4, 64, 1, 4096, 4416, 16, 256, 64 << 20, _lib.Overrides(one_shot=2), q) == 0 and q.one_shot == 0
    # template knobs: QuantMapMode digit 3 -> ring kernel, 1 / 2 -> one-shot with 4 / 8 pieces per wave
    assert lib.flute_qgemm_plan_ex(0, 4, 64, 1, 4096, 4096, 19, 256, 64 << 20, None, q) == 0 and q.one_shot == 0
    assert lib.flute_qgemm_plan_ex(0, 4, 64, 1, 4096, 4096, 17, 256, 64 << 20, None, q) == 0 and q.one_shot >= 1 and q.ring_depth == 4
    assert lib.flute_qgemm_plan_ex(0, 4, 64, 1, 4096, 4096, 18, 256, 64 << 20, None, q) == 0 and q.one_shot >= 1 and q.ring_depth == 8
    # skinny MFMA kernel (family 5): 4-bit, 3 <= M <= 16, slabs of 64 columns filling 55..100 % of the CUs (or >= 1.7 rounds), K = 4096
    rc, p = plan(16, 11008, 4096, ovr=_lib.Overrides(family=5))
    assert rc == 0 and p.family == 5 and (p.grid, p.waves, p.kw, p.ring_depth, p.splitk, p.workspace_needed) == (172, 8, 8, 16, 1, 0)
    assert p.lds_bytes <= 160 * 1024
    rc, p = plan(16, 11008, 4096)                            # round 6: the lean MFMA decode kernel, three column groups per workgroup on one staged activation set (11.4 against 12.6 us)
    assert rc == 0 and p.family == 7 and (p.grid, p.slabs_per_wave, p.waves, p.block) == (230, 3, 8, 512)
    rc, p = plan(16, 8192, 4096)
    assert rc == 0 and p.family == 7 and (p.grid, p.slabs_per_wave) == (256, 2)
    rc, p = plan(16, 14336, 4096)                            # 896 groups: more than one round even at three groups per workgroup - the skinny kernel
    assert rc == 0 and p.family == 5 and p.grid == 224
    rc, p = plan(4, 14336, 4096, ovr=_lib.Overrides(family=5))
    assert rc == 0 and p.family == 5 and p.grid == 224
    rc, p = plan(4, 14336, 4096)                             # (round 6: up to four rows the persistent MFMA decode kernel - 11.5 against 12.1 us)
    assert rc == 0 and (p.family, p.slabs_per_wave, p.grid, p.visits) == (8, 1, 224, 4)
    rc, p = plan(16, 28672, 4096)                            # 448 slabs: the per-wave kernel, two slabs per wave, no lane sharing
    assert rc == 0 and p.family == 2 and (p.m_block, p.m_tiles, p.slabs_per_wave, p.grid) == (1, 1, 2, 224)
    rc, p = plan(16, 10240, 8192, ovr=_lib.Overrides(family=2))     # 160 slabs fill 62 % of the CUs: no lane sharing either
    assert rc == 0 and p.family == 2 and (p.m_block, p.slabs_per_wave, p.grid) == (1, 1, 160)
    rc, p = plan(16, 10240, 8192)                            # (round 6: the persistent MFMA decode kernel - 214 workgroups x one set of three column groups)
    assert rc == 0 and (p.family, p.slabs_per_wave, p.grid, p.visits) == (8, 3, 214, 1)
    rc, p = plan(16, 4096, 4096)                             # one round of 4-unit workgroups: the lean MFMA decode kernel (round 5)
    assert rc == 0 and p.family == 7 and p.grid == 256
    assert _lib.get().flute_qgemm_plan_ex(0, 4, 64, 16, 4096, 4096, 16, 256, 64 << 20, _lib.Overrides(family=2), p) == 0
    assert p.family == 2 and (p.m_block, p.grid) == (4, 256)   # the per-wave kernel there: 64 slabs, four lanes share a unit
    rc, p = plan(16, 4096, 4096, tid=19)                     # QuantMapMode digit 3 at M <= 16: skinny wherever it exists
    assert rc == 0 and p.family == 5 and p.grid == 64
    for (M, N, K, bits, tid) in ((16, 4096, 4096, 4, 16), (16, 20480, 4096, 4, 16), (16, 28672, 4096, 4, 16), (17, 14336, 4096, 4, 16), (2, 14336, 4096, 4, 16),
                                 (16, 14336, 4096, 2, 0), (16, 14336, 8192, 4, 16), (16, 14336, 4096, 4, 17), (16, 14336, 4096, 4, 18),
                                 (32, 14336, 4096, 4, 19)):
        rc, p = plan(M, N, K, bits=bits, tid=tid)
        assert rc == 0 and p.family != 5, (M, N, K, bits, tid)
    assert lib.flute_qgemm_plan_ex(0, 4, 64, 8, 1024, 2048, 16, 256, 64 << 20, _lib.Overrides(family=5), q) == 0
    assert q.family == 5 and (q.grid, q.waves, q.ring_depth) == (16, 8, 8)
    assert lib.flute_qgemm_plan_ex(0, 4, 64, 8, 1024, 2048, 16, 256, 64 << 20, _lib.Overrides(family=5, waves=4), q) == 0
    assert q.family == 5 and (q.waves, q.ring_depth) == (4, 16)
    assert lib.flute_qgemm_plan_ex(0, 4, 64, 8, 1024, 4096 + 64, 16, 256, 64 << 20, _lib.Overrides(family=5), q) == 0 and q.family == 2
    # decode kernel: persistent grid never exceeds the unit groups
    rc, p = plan(2, 28672, 8192, tid=19)
    assert rc == 0 and p.family == 0 and p.grid <= 28672 // 4


def test_plan_invariants_over_random_shapes():
    """Every plan the planner can hand out, over random shapes x template ids x CU counts: the launch fits the hardware and
    the numbers a kernel derives its geometry from are consistent with the shape (what each kernel's host contract
    says in its header).  No GPU."""
    import random
    from flute_amd import TEMPLATE_CONFIGS
    lib = _lib.get()
    p = _lib.Plan()
    rng = random.Random(7)
    fams = {}
    for _ in range(6000):
        bits = rng.choice([4, 4, 2, 3])
        tids = [t for (b, t), c in sorted(TEMPLATE_CONFIGS.items()) if b == bits and (bits != 3 or c["TileP"] == 32)]
        tid = rng.choice(tids)
        if bits == 4 and rng.random() < 0.2:                  # the automatic ids (QuantMapMode digit 0, Stages 2 / 3, SMs_Multiple 1) more often
            tid = rng.choice([0, 4, 16, 20])
        tile_p = TEMPLATE_CONFIGS[(bits, tid)]["TileP"]
        g = rng.choice([32, 64, 128, 256])
        J = 16 if bits == 3 else 16 // bits
        N = J * tile_p * rng.choice([1, 2, 4, 8, 16, 28, 32, 43, 56, 64, 112, 224])
        K = rng.choice([256, 512, 1024, 2048, 3584, 4096, 4096 + 64, 5120, 8192, 11008, 14336, 28672])
        if K % g:
            continue
        M = rng.choice([1, 1, 2, 3, 4, 5, 8, 16, 17, 32, 33, 48, 64, 65, 96, 127, 128, 256, 1000, 4096])
        num_sms = rng.choice([256, 256, 256, 304, 8, 120])
        dtype = rng.choice([0, 1])
        rc = lib.flute_qgemm_plan_ex(dtype, bits, g, M, N, K, tid, num_sms, 64 << 20, None, p)
        assert rc == 0, (bits, tid, g, M, N, K, num_sms, rc)
        key = (p.family, p.one_shot)
        fams[key] = fams.get(key, 0) + 1
        units = N // J
        what = (bits, tid, g, M, N, K, num_sms, p.family, p.one_shot)
        assert p.grid >= 1 and 64 <= p.block <= 1024 and p.block == p.waves * 64, what
        assert 0 < p.lds_bytes <= 160 * 1024, what
        assert p.splitk >= 1 and (p.splitk == 1 or p.workspace_needed <= 64 << 20), what
        if p.family == 0:
            assert M <= 4 and p.m_block >= M and p.waves % p.kw == 0, what
            if p.one_shot == 3:                                   # persistent one-shot kernel
                assert M <= 2 and p.kw == 1 and p.waves <= 8 and K % (512 * p.ring_depth) == 0, what
                assert p.k_chunks * p.ring_depth * 512 == K and p.grid * p.waves * p.visits >= units, what
                assert (K // g) % 2 == 0 and g >= 64, what
            elif p.one_shot in (1, 2):                            # one-shot kernel
                assert p.visits == 1 and p.splitk == 1 and (K // g) % 2 == 0 and g >= 64, what
                assert p.grid == -(-units // (p.waves // p.kw)) and -(-(-(-K // 512)) // p.kw) <= p.ring_depth, what
            elif p.one_shot == 4:                                 # lean one-row kernel (qgemm_fast.h's host contract)
                assert bits == 4 and M <= p.m_block <= 4 and g >= 64 and K in (2048, 3584, 4096, 8192) and 512 * p.ring_depth * p.kw == K, what
                assert p.waves in (4, 8) and p.grid * (p.waves // p.kw) == N // 4 and p.m_block * K * 2 <= 32768, what
                assert p.lds_bytes == 65536 + p.m_block * 2 * K + p.waves * 4 * p.ring_depth * (512 // g) * 2 + 128 + 64 * p.waves, what
                assert tid % 4 == 0, what                         # automatic only for the ids whose last digit leaves the choice to the planner
            else:
                assert p.ring_depth in (2, 4), what
        elif p.family == 5:                                       # skinny MFMA kernel
            assert bits == 4 and 3 <= M <= 16 and p.splitk >= 1, what
            assert p.ring_depth in (4, 8, 16) and p.ring_depth * p.waves * 32 * p.splitk == K and p.grid == N // 64 * p.splitk, what
            assert (K // g) % 2 == 0 and (p.ring_depth * 32) // g <= 8, what
            if p.splitk > 1:                                      # grid-level K split inside the launch (round 4): 4 KB of slab per slice and slab
                assert p.splitk_mode == 1 and p.workspace_needed == p.splitk * (N // 64) * 4096 + 65536 and p.grid <= num_sms and K // p.splitk >= 2048, what
            else:
                assert p.workspace_needed == 0, what
        elif p.family == 7:                                       # lean MFMA decode kernel (qgemm_fastm.h's host contract)
            assert bits == 4 and 5 <= M <= 16 and K in (2048, 4096) and g >= 64 and (K // 8) // g >= 2 and tid % 4 == 0, what
            ng = p.slabs_per_wave                                  # column groups per workgroup (round 6): the fewest that make one round
            assert ng in (1, 2, 3) and p.grid == -(-(N // 16) // ng) and p.grid <= num_sms and (ng == 1 or -(-(N // 16) // (ng - 1)) > num_sms), what
            assert p.waves == 8 and p.lds_bytes == 32768 + 32 * K and p.ring_depth * 128 * 8 == K, what
        elif p.family == 8:                                       # persistent MFMA decode kernel (qgemm_persistm.h's host contract)
            assert bits in (2, 4) and 3 <= M <= 16 and K % 128 == 0 and g in (64, 128) and (g == 64 or K % 256 == 0) and (bits == 2 or tid % 4 == 0), what
            assert K >= 6144 or (K >= 3584 and (bits == 2 or M <= 8 or K not in (2048, 4096))), what
            ng, xr = p.slabs_per_wave, p.k_chunks                  # column groups per set, activation requests per macro-step
            nsets = -(-(N // 16) // ng)
            assert ng in (1, 2, 3) and xr == (1 if M <= 4 else 2 if M <= 8 else 4) and p.grid <= min(num_sms, nsets) and p.grid * p.visits >= nsets, what
            assert (p.visits - 1) * p.grid < nsets and N * K + (1 if M >= 5 or bits == 2 else 0) > 16 << 20 and (N // 16) * 2 >= num_sms, what
            assert bits == 4 or M <= 8 or p.visits * K <= 28672, what
            dx = 3 if (xr == 4 or (xr == 2 and ng == 3)) else 6
            xres = p.one_shot                                      # activations resident in LDS: 4 xr rows x K within 64 KB
            assert xres == (1 if (K * xr <= 8192 and xr <= 2 and not (xr == 1 and ng == 3)) else 0), what
            rings = 65536 + 8 * 6 * 256 if xres else 8 * dx * (xr * 1024 + 256)
            assert p.waves == 8 and p.lds_bytes == (2048 if bits == 2 else 65536 if xr == 1 else 32768) + rings + 8 * ng * 1024 and N * K // 2 < 2 ** 32, what
        elif p.family == 2:                                       # per-wave MFMA kernel
            assert p.m_block in (1, 2, 4) and p.m_tiles in (1, 2, 4) and p.slabs_per_wave in (1, 2), what
            assert p.slabs_per_wave == 1 or (bits == 4 and p.m_block == 1), what
            assert bits != 3 or p.m_block == 1, what
        elif p.family == 6:                                       # split-K block kernel (qgemm_splitk.h's host contract)
            assert p.m_tiles in (8, 4) and p.kw in (2, 4) and (p.kw == 2 or p.m_tiles == 4), what      # 128- / 64-row tiles; K parts per workgroup (round 6: 4 = 64-column tiles)
            tiles = -(-M // (p.m_tiles * 16)) * (N // (256 // p.kw))
            assert bits in (2, 4) and M >= 33 and p.block == 768 and p.waves == 12 and p.grid == tiles * p.splitk, what
            assert p.k_per_split * p.splitk == K and p.k_per_split % (p.kw * max(64, g)) == 0 and (K // g) % 8 == 0 and N % 128 == 0, what
            gw = p.k_per_split // g                               # one scale image of eight 8-group blocks per column group
            assert gw + (7 if gw % 8 else 0) <= 64, what
            slab = p.m_tiles * 16384 // p.kw                       # 8 waves x m_tiles / kw row tiles x 2 KB
            assert p.splitk_mode == (1 if p.splitk > 1 else 0) and p.workspace_needed == (p.splitk * tiles * slab + 65536 if p.splitk > 1 else 0), what
            assert p.lds_bytes == (128 << (2 * bits)) + p.kw * 3 * p.m_tiles * 2048 + (8 // p.kw) * 4096, what
        else:                                                     # block kernels
            assert p.family == 3 and p.m_block in (4, 5, 9, 10, 12) and p.block == 512, what
            assert (K // g) % 8 == 0 and K % 64 == 0 and N % 256 == 0, what
            assert p.m_block != 4 or bits != 3 or p.lds_bytes == 146 * 1024, what
    # the sweep reaches every kernel of the library
    for key in ((0, 0), (0, 1), (0, 2), (0, 3), (0, 4), (2, 0), (3, 0), (5, 0), (6, 0), (7, 0), (8, 0), (8, 1)):
        assert fams.get(key, 0) > 0, (key, fams)


def test_tuned_digit3_ids_keep_the_skinny_kernel_at_m3_m4():
    """a 4-bit id with QuantMapMode digit 3 was tuned on the skinny MFMA kernel; the round-4 rule that sends
    small 2- / 4-bit layers to the four-row decode kernel at M = 3, 4 applies to automatic-digit ids only."""
    for (N, K) in ((4608, 2048), (8192, 2048), (4096, 4096)):
        for M in (3, 4, 8):
            rc, p = plan(M, N, K, tid=19)
            assert rc == 0 and p.family == 5, (M, N, K, p.family, p.one_shot)
        rc, p = plan(4, N, K, tid=16)                      # automatic digit: the planner's own choice (decode kernel on <= 16 M weights)
        assert rc == 0 and p.family == 0, (N, K, p.family)


def test_per_wave_overrides_are_not_reinterpreted_by_the_splitk_planner():
    """m_tiles / waves / splitk given WITHOUT family = 6 belong to the per-wave kernel."""
    lib = _lib.get()
    q = _lib.Plan()
    assert lib.flute_qgemm_plan_ex(0, 4, 64, 256, 11008, 4096, 16, 256, 64 << 20, _lib.Overrides(m_tiles=2), q) == 0
    assert q.family == 2 and q.m_tiles == 2, (q.family, q.m_tiles)
    assert lib.flute_qgemm_plan_ex(0, 4, 64, 256, 11008, 4096, 16, 256, 64 << 20, _lib.Overrides(family=6, m_tiles=4), q) == 0
    assert q.family == 6 and q.m_tiles == 4


def test_plan_rejects_bad_arguments():
    assert plan(1, 4096, 4096, bits=5)[0] == -1
    assert plan(1, 4096, 4096, g=48)[0] == -2
    assert plan(1, 4096, 4096, tid=999)[0] == -3
    assert plan(1, 4096, 4096, bits=3, tid=0)[0] == -3      # 3-bit has no TileP=64 layout (utils.py:137-139)
    assert plan(1, 4096 + 16, 4096)[0] == -4                # N not a multiple of the packing block
    assert plan(1, 4096, 4096 + 32)[0] == -4                # K % 64
    assert plan(1, 4096, 4096, g=256, tid=16)[0] == 0
    assert plan(1, 4096, 4096, dtype=3)[0] == -7
    assert _lib.get().flute_qgemm_plan(0, 4, 64, 1, 4096, 4096, 16, 256, 0, None) == -9


def test_qgemm_null_and_empty_calls_return_before_launch():
    lib = _lib.get()
    args = [0, 4, 64, 0, 4096, 4096, 1024] + [None] * 7 + [0, 16, 256, None]
    assert lib.flute_qgemm(*args) == 0                      # M == 0: nothing to do
    args[3] = 1
    assert lib.flute_qgemm(*args) == -9                     # null pointers are refused, no launch
    args[6] = 7
    assert lib.flute_qgemm(*args) == -4                     # P inconsistent with N
    assert lib.flute_hadamard(0, None, None, 0, 64, None) == 0
    assert lib.flute_hadamard(0, None, None, 64, 64, None) == -9


def test_qgemm_hadamard_entry_host_logic():
    """flute_qgemm_hadamard: which calls fuse the rotation (decode plans, power-of-two blocks <= 512 that
    divide K) and the argument checks that run before any launch."""
    lib = _lib.get()
    ws = 64 << 20
    fused = lambda M, K, h, bits=4, tid=16: lib.flute_qgemm_hadamard_fused(0, bits, 64, h, M, 4096, K, tid, 256, ws)  # noqa: E731
    assert fused(1, 4096, 512) == 1 and fused(4, 2048, 16) == 1 and fused(1, 3584, 512) == 1 and fused(4, 4096, 16) == 0
    assert fused(5, 4096, 512) == 0          # MFMA plan: rotated into the caller's scratch first
    assert fused(1, 4096, 1024) == 0         # block wider than one wave's 512-element span
    assert fused(1, 4096 + 64, 512) == 0     # blocks would straddle rows
    assert fused(1, 4096, 48) == 0           # not a power of two
    assert fused(3, 2048, 512, bits=3, tid=4) == 1 and fused(2, 4096, 512, bits=3, tid=4) == 1     # 3 bits: four rows too
    head = [0, 4, 64, 512, 0, 4096, 4096, 1024]
    tail = [None] * 8 + [0, 16, 256, None]
    assert lib.flute_qgemm_hadamard(*(head + tail)) == 0            # M == 0
    head[4] = 1
    head[3] = 48
    assert lib.flute_qgemm_hadamard(*(head + tail)) == -8           # hadamard size must be a power of two
    head[3] = 512
    assert lib.flute_qgemm_hadamard(*(head + tail)) == -9           # null tensors, fused plan: refused before launch
    head[4] = 9
    assert lib.flute_qgemm_hadamard(*(head + tail)) == -9           # unfused plan without scratch


def test_round5_planner_rules():
    """The plans the round-5 regret sweeps asked for (profiles/r05_planner_regret_*.json), as host logic: no GPU."""
    # two slabs per wave x the grid K split on a deep layer: QuantMapMode digit 3 above M = 16, and the automatic digit for K >= 16384
    for tid in (0, 3, 16):
        rc, p = plan(64, 8192, 28672, tid=tid)
        assert rc == 0 and (p.family, p.m_block, p.slabs_per_wave, p.splitk, p.grid) == (2, 1, 2, 4, 256), (tid, p.as_dict())
    rc, p = plan(64, 8192, 28672, tid=1)                          # digit 1 keeps its meaning: no lane sharing, one slab per wave
    assert rc == 0 and (p.family, p.m_block, p.slabs_per_wave) == (2, 1, 1)
    rc, p = plan(16, 8192, 28672, tid=3)                          # digit 3 at M <= 16: unchanged (the skinny kernel / two slabs per wave)
    assert rc == 0 and p.family in (2, 5)
    # a grid K split instead of more lane sharing on deep layers (K >= 10240: two slices, K >= 12288: four)
    rc, p = plan(4, 4096, 11008, ovr=_lib.Overrides(family=2))
    assert rc == 0 and (p.family, p.m_block, p.splitk, p.grid) == (2, 2, 2, 256), p.as_dict()
    rc, p = plan(4, 4096, 11008)                                  # (round 6: the persistent MFMA decode kernel serves it - 9.2 against 14.7 us)
    assert rc == 0 and (p.family, p.slabs_per_wave, p.grid) == (8, 1, 256), p.as_dict()
    rc, p = plan(48, 3584, 14336)                                 # (round 6: 64 x 64 tiles x 4 slices are faster still - 16.7 against 21.6 us; by override the rule stands)
    assert rc == 0 and (p.family, p.m_tiles, p.kw, p.splitk, p.grid) == (6, 4, 4, 4, 224), p.as_dict()
    rc, p = plan(48, 3584, 14336, ovr=_lib.Overrides(family=2))
    assert rc == 0 and (p.family, p.m_block, p.splitk) == (2, 1, 4), p.as_dict()
    rc, p = plan(32, 8192, 8192)                                  # K = 8192: lane sharing stays (measured in round 3)
    assert rc == 0 and p.family == 2 and p.m_block == 2 and p.splitk == 1, p.as_dict()
    # 3-bit blocks x UNEVEN K slices (K = 3584 = 3 x 1024 + 512), the cheapest candidate, combined in the launch
    rc, p = plan(128, 14336, 3584, bits=3, tid=5, dtype=1)
    assert rc == 0 and (p.family, p.m_block, p.splitk, p.k_per_split, p.grid, p.splitk_mode) == (3, 5, 4, 1024, 224, 1), p.as_dict()
    rc, p = plan(128, 8192, 8192, bits=3, tid=4, dtype=1)         # 64-row blocks x 4 slices beat 128-row blocks x 4 here: reduce launch
    assert rc == 0 and (p.family, p.m_block, p.splitk, p.grid, p.splitk_mode) == (3, 12, 4, 256, 0), p.as_dict()
    # skinny 3-bit blocks up to M = 48 only where their grid fills 80 % of the CUs
    rc, p = plan(48, 8192, 8192, bits=3, tid=4, dtype=1)
    assert rc == 0 and (p.family, p.m_block, p.grid) == (3, 12, 256), p.as_dict()
    rc, p = plan(48, 10240, 8192, bits=3, tid=4, dtype=1)
    assert rc == 0 and p.family == 2, p.as_dict()
    # the lean decode kernel: K = 3584 (7 pieces), K = 8192 for two rows on >= 80 % of a round, K = 2048 two rows up to two rounds
    rc, p = plan(1, 4096, 3584)
    assert rc == 0 and (p.family, p.one_shot, p.waves, p.kw, p.ring_depth, p.grid) == (0, 4, 4, 1, 7, 256), p.as_dict()
    rc, p = plan(2, 3584, 8192)
    assert rc == 0 and (p.one_shot, p.waves, p.kw, p.ring_depth, p.m_block, p.grid) == (4, 8, 2, 8, 2, 224), p.as_dict()
    assert plan(1, 3584, 8192)[1].one_shot != 4 and plan(2, 2048, 8192)[1].one_shot != 4
    rc, p = plan(2, 8192, 2048)
    assert rc == 0 and (p.one_shot, p.grid) == (4, 512), p.as_dict()
    assert plan(4, 8192, 2048)[1].one_shot != 4 and plan(2, 16384, 2048)[1].one_shot != 4



def test_round6_planner_rules():
    """Round 6, as host logic: the split-K block kernel's 64 x 64 tiles (four K parts per workgroup) where they fill at least half
    the chip in one round (measured: profiles/r06/call17_automatic_plan_vs_forced.log), XCD groups of up to eight row tiles, and the
    lean MFMA decode kernel's column groups per workgroup (profiles/r06/call18_fastm_column_groups.log)."""
    def sk(M, N, K, **kw):
        rc, p = plan(M, N, K, **kw)
        assert rc == 0, (M, N, K)
        return (p.family, p.m_tiles, p.kw, p.splitk, p.grid, p.m_block)
    assert sk(256, 4096, 4096) == (6, 4, 4, 1, 256, 4)                 # all of K per workgroup: no seam; the four row tiles of a column tile on one XCD
    assert sk(192, 4096, 4096) == (6, 4, 4, 1, 192, 1)                 # three row tiles: natural order
    assert sk(128, 4096, 4096) == (6, 4, 4, 2, 256, 1)                 # two K slices, combined in the launch (L form: one row tile per wave)
    assert sk(512, 2048, 4096) == (6, 4, 4, 1, 256, 8)
    assert sk(256, 2048, 8192) == (6, 4, 4, 2, 256, 1)
    assert sk(256, 11008, 4096) == (6, 8, 2, 1, 172, 2)                # more than one round of 64 x 64 tiles: 128-row tiles as before
    assert sk(1024, 4096, 4096) == (6, 8, 2, 1, 256, 8)
    rc, p = plan(256, 4096, 4096)
    assert p.lds_bytes == 32768 + 4 * 3 * 8192 + 2 * 4096 and p.workspace_needed == 0 and p.splitk_mode == 0 and p.block == 768
    rc, p = plan(256, 4096, 4096, ovr=_lib.Overrides(family=6, kw=2))  # rounds 4 / 5: 64 x 128 tiles x two slices
    assert rc == 0 and (p.kw, p.m_tiles, p.splitk, p.grid, p.splitk_mode) == (2, 4, 2, 256, 1)
    assert plan(256, 4096, 4096, ovr=_lib.Overrides(family=6, waves=8))[0] != 0          # the variant without loader waves is gone
    assert plan(256, 4096, 4096, ovr=_lib.Overrides(family=6, kw=4, m_tiles=8))[0] != 0  # four K parts: 64-row tiles only (LDS)
    assert plan(256, 4096, 8192, ovr=_lib.Overrides(family=6, kw=4, splitk=1))[0] != 0   # 128 groups per workgroup K range: more than one scale image holds
    assert plan(256, 4096, 8192, ovr=_lib.Overrides(family=6, kw=4, splitk=2))[0] == 0
    # lean MFMA decode kernel: the fewest column groups per workgroup that make one round
    for (N, ng, grid) in ((4096, 1, 256), (5120, 2, 160), (6144, 2, 192), (8192, 2, 256), (11008, 3, 230)):
        rc, p = plan(16, N, 4096)
        assert rc == 0 and (p.family, p.slabs_per_wave, p.grid) == (7, ng, grid), (N, p.as_dict())
    rc, p = plan(16, 14336, 4096)
    assert rc == 0 and p.family == 5                                    # 299 workgroups at three groups: more than one round
    rc, p = plan(16, 11008, 4096, ovr=_lib.Overrides(family=7, slabs_per_wave=1))
    assert rc == 0 and (p.family, p.slabs_per_wave, p.grid) == (7, 1, 688)
