"""Layout contracts of the split-K block kernel (flute_amd/csrc/qgemm_splitk.h), modelled lane by lane on the CPU: what a
wave's LDS-DMA requests write, what its ds_read_b128 / ds_read_u16 then read, which bank slots the reads of a lane group
touch, and the plan the host makes for it.  The formulas below are the kernel's, transcribed; the GPU parity tests
(tests/test_qgemm_gpu.py::test_splitk_block_kernel) check the kernel itself against the oracle."""
import numpy as np
import pytest

from flute_amd import _lib

# ds_read_b128 lane groups of MI355X_MICROARCH.md (LDS table): one LDS cycle per group when the 16 lanes hit 16 different
# 16-B slots of the 256-B bank row
GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))]
GROUPS += [[l + 32 for l in g] for g in GROUPS]


def sk_swz(row8, rh):
    return (row8 >> 1) | (rh << 2)


def test_activation_pieces_land_where_the_fragments_read_them():
    K, m0, k0 = 256, 128, 64
    X = np.arange(384 * K, dtype=np.int64).reshape(384, K)          # value = flat element index
    stage = np.full(16 * 1024 // 2, -1, dtype=np.int64)               # one 64-k stage of a K half, in 2-B elements
    for wg in range(4):                                               # the four waves of a K half, PPW = 4 pieces each
        for i in range(4):
            rt, rh = wg * 2 + (i >> 1), i & 1
            for lane in range(64):
                row8 = lane >> 3
                src_row = m0 + rt * 16 + rh * 8 + row8
                src_k = k0 + ((lane & 7) ^ sk_swz(row8, rh)) * 8
                dst = ((wg * 4 + i) * 1024 + lane * 16) // 2          # lane-linear 16 B
                stage[dst:dst + 8] = X[src_row, src_k:src_k + 8]
    assert (stage >= 0).all()
    for R in range(8):
        for h in range(2):
            addrs = {}
            for lane in range(64):
                r16, q4 = lane & 15, lane >> 4
                a = R * 2048 + (r16 >> 3) * 1024 + (r16 & 7) * 128 + (((h * 4 + q4) ^ sk_swz(r16 & 7, r16 >> 3)) * 16)
                addrs[lane] = a
                got = stage[a // 2:a // 2 + 8]
                want = X[m0 + R * 16 + r16, k0 + h * 32 + q4 * 8:k0 + h * 32 + q4 * 8 + 8]   # MFMA B operand: row r16, k = 8 q4 ..
                assert (got == want).all(), (R, h, lane)
            for grp in GROUPS:
                assert len({(addrs[l] // 16) % 16 for l in grp}) == 16, (R, h)


def weight_row(r16, U):
    """qgemm_splitk.h (round 6): MFMA weight row r16 = (b3 b2 b1 b0) of a column tile -> (unit, field): lane bit 1 is a field bit."""
    return (((r16 >> 2) % (U // 2)) << 1) | (r16 & 1), ((r16 >> 2) // (U // 2)) * 2 + ((r16 >> 1) & 1)


@pytest.mark.parametrize("bits,tile_p", [(4, 32), (4, 64), (2, 32), (2, 64)])
@pytest.mark.parametrize("kp", [2, 4])
def test_scale_image_and_output_columns(bits, tile_p, kp):
    J = 16 // bits
    U = 32 // J
    FPT = 16 // U
    N, G, lg = 512, 72, 6
    S = np.arange(N * G, dtype=np.int64).reshape(N, G)

    def unit_col0(u):
        return (u // tile_p) * (J * tile_p) + (u % tile_p)

    # every weight row of a column tile is one (unit, field), every (unit, field) exactly once
    assert sorted(weight_row(r, U) for r in range(16)) == sorted((u, f) for u in range(U) for f in range(FPT))
    bpw = 8 // kp
    for unit0 in (0, U * 5):
        for kwg in (0, 512, 1600):                                    # first group of the workgroup's K range: 0, 8, 25
            g0e = (kwg >> lg) & ~7
            img = np.full(4096 // 2, -1, dtype=np.int64)              # ONE image per column group: eight 8-group blocks x 32 columns x 16 B
            for kh in range(kp):                                      # the KP waves of the group fetch 8 / KP blocks each
                for r in range(bpw // 2):
                    for lane in range(64):
                        cl = lane & 31
                        col = unit_col0(unit0 + cl % U) + (cl // U) * tile_p
                        g = g0e + (kh * bpw + (lane >> 5)) * 8 + r * 16
                        dst = ((kh * bpw + 2 * r) * 512 + lane * 16) // 2
                        for e in range(8):
                            img[dst + e] = S[col, g + e] if g + e < G else 0
            assert (img >= 0).all()
            for lane in range(64):
                r16 = lane & 15
                u8, fsel = weight_row(r16, U)
                for t in range(2):
                    col = unit_col0(unit0 + u8) + (fsel + FPT * t) * tile_p    # weight row r16 of column tile t
                    for k in range(kwg, min(kwg + 2048, G << lg), 32):          # every 32-k half step of any K part
                        rel = (k >> lg) - g0e
                        if rel >= 64:
                            break
                        a = (fsel * U + u8) * 16 + (rel >> 3) * 512 + (rel & 7) * 2 + t * 256
                        assert img[a // 2] == S[col, k >> lg], (lane, t, k)
    # the epilogue: accumulator register j of lane (r16, q4) = weight row 4 q4 + j; after v_permlane16_swap of (j0, j2) and (j1, j3)
    # between the lane rows q4, q4 ^ 1 every lane holds four consecutive columns: a = rows (a0, b0, a2, b2), b = rows (a1, b1, a3, b3)
    regs = {q4: [weight_row(4 * q4 + j, U) for j in range(4)] for q4 in range(4)}
    for q4 in range(4):
        p = q4 ^ 1
        if q4 % 2 == 0:
            held = [regs[q4][0], regs[q4][1], regs[p][0], regs[p][1]]       # a keeps its even rows, b's even rows take a's odd rows
        else:
            held = [regs[p][2], regs[p][3], regs[q4][2], regs[q4][3]]
        c_unit = ((q4 & ~1) % (U // 2)) * 2
        c_field = (q4 // (U // 2)) * 2 + (q4 & 1)
        for t in range(2):
            c0 = unit_col0(c_unit) + (c_field + FPT * t) * tile_p
            for j, (u, f) in enumerate(held):
                assert unit_col0(u) + (f + FPT * t) * tile_p == c0 + j, (q4, t, j)


def test_weight_request_touches_two_lines_per_lane_quad():
    """Round 6: lane (r16, q4) asks for chunk 4 ((r16 >> 1) & 1) + q4 of its unit's 128-B step piece - the four lanes of a quad are two
    units x two half steps (two cache lines; rounds 4 / 5: four units = four lines) - and quad_perm [2h, 2h + 1, 2h, 2h + 1] hands
    every lane the half step h of ITS unit."""
    for U in (8, 4):
        held = {}
        for lane in range(64):
            r16, q4 = lane & 15, lane >> 4
            u8, _ = weight_row(r16, U)
            held[lane] = (u8, (r16 >> 1) & 1, q4)                      # (unit row, half step, 16-B chunk of the half step)
        for quad in range(16):
            lines = {held[l][0] for l in range(4 * quad, 4 * quad + 4)}
            assert len(lines) == 2
        for h in (0, 1):
            for lane in range(64):
                src = (lane & ~3) | (2 * h + (lane & 1))
                u8, _ = weight_row(lane & 15, U)
                assert held[src] == (u8, h, lane >> 4)


def _plan(M, N, K, bits=4, g=64, tid=16, ws=64 << 20, dtype=0, **ovr):
    p = _lib.Plan()
    rc = _lib.get().flute_qgemm_plan_ex(dtype, bits, g, M, N, K, tid, 256, ws, _lib.Overrides(**ovr), p)
    return rc, p


def test_plan_family6():
    rc, p = _plan(256, 4096, 4096, family=6)
    assert rc == 0 and p.family == 6 and p.block == 768 and p.waves == 12 and p.kw == 4      # 8 compute + 4 loader waves; round 6: four K parts
    assert _plan(256, 4096, 4096, family=6, waves=8)[0] != 0                                 # the variant without loader waves was dropped in round 6
    assert p.m_tiles == 4 and p.splitk == 1 and p.k_per_split == 4096 and p.grid == 256 and p.splitk_mode == 0     # 64 x 64 tiles over all of K: no seam (15.5 us; 64 x 128 x two slices 18.3)
    assert p.workspace_needed == 0 and p.lds_bytes == 32768 + 4 * 3 * 8192 + 2 * 4096 and p.m_block == 4      # all four row tiles of a column tile on one XCD
    rc, p = _plan(256, 4096, 4096, family=6, kw=2)
    assert rc == 0 and p.kw == 2 and p.m_tiles == 4 and p.splitk == 2 and p.k_per_split == 2048 and p.grid == 256 and p.splitk_mode == 1     # rounds 4 / 5: 64-row tiles x two slices
    assert p.workspace_needed == 2 * 128 * 32768 + (64 << 10) and p.lds_bytes == 32768 + 49152 + 16384   # 32 KB per 64-row tile and slice
    rc, p = _plan(256, 4096, 4096, family=6, m_tiles=8)
    assert rc == 0 and p.m_tiles == 8 and p.kw == 2 and p.splitk == 2 and p.grid == 128 and p.workspace_needed == 2 * 64 * 65536 + (64 << 10) and p.lds_bytes == 32768 + 98304 + 16384
    assert _plan(256, 4096, 4096, family=6, splitk=16, m_tiles=8)[0] != 0     # 64 MiB of slabs + the state words: one page too many
    for sk in (1, 2, 4, 8, 16):
        rc, p = _plan(256, 4096, 4096, family=6, splitk=sk, m_tiles=8, ws=128 << 20)
        assert rc == 0 and p.splitk == sk and p.grid == 64 * sk and p.k_per_split * sk == 4096
        assert p.splitk_mode == (1 if sk > 1 else 0)
    assert _plan(256, 4096, 4096, family=6, splitk=3)[0] != 0          # 4096 / 3
    assert _plan(256, 4096, 4096, family=6, splitk=32)[0] != 0
    assert _plan(256, 4096, 4096, family=6, splitk=4, ws=1 << 20)[0] != 0   # slabs do not fit
    assert _plan(256, 4096, 4096, family=6, splitk=1, ws=0)[0] == 0
    assert _plan(256, 4096, 4096, bits=3, tid=0, family=6)[0] != 0     # 3 bits: other kernels
    assert _plan(256, 4096, 4096, g=32, family=6, splitk=1)[0] != 0    # 128 groups per workgroup K range: more than eight scale blocks
    assert _plan(256, 4096, 4096, g=32, family=6, splitk=2)[0] == 0
    rc, p = _plan(256, 4096, 4096, g=256, family=6, splitk=8)           # K half = one 256-wide group
    assert rc == 0 and p.k_per_split == 512
    assert _plan(256, 4096, 4096, g=256, family=6, splitk=16)[0] != 0
    rc, p = _plan(200, 11008, 4096, family=6, m_tiles=8)
    assert rc == 0 and p.grid == 2 * 86 * p.splitk
    assert _plan(256, 4096, 4096, family=4)[0] != 0 and _plan(256, 4096, 4096, family=8)[0] != 0   # unknown families are refused (7 = the lean MFMA decode kernel since round 5: falls back above M = 16)


def test_xcd_group_order_is_a_bijection():
    """qgemm_splitk.h's XCD-aware tile order (splitk == 1): groups of E = 1, 2, 4, 8 row tiles x column tile are dealt to the eight XCDs
    (block b -> XCD b % 8) so that the E row tiles that share a column tile's weights run on ONE XCD (round 4: pairs; round 6: up to
    eight - the four 64-row tiles of M = 256); every tile is produced exactly once for any tile grid, the incomplete last group of
    eight keeps the natural order."""
    def group_of(tiles_m, ovr=0):                                   # api.hip, plan_splitk: the largest of 8, 4, 2 that divides into a power of two
        E = 8
        while E > 1 and (tiles_m % E or ((tiles_m // E) & (tiles_m // E - 1))):
            E >>= 1
        if ovr in (1, 2, 4, 8):
            E = ovr
        if tiles_m % E or ((tiles_m // E) & (tiles_m // E - 1)):
            E = 1
        return E

    def remap(tile, tiles_m, tiles_n, E):
        if E < 2:
            return tile % tiles_m, tile // tiles_m                  # api.hip: pair_lg = -1
        P = tiles_m // E
        e_lg, lg, c8 = E.bit_length() - 1, P.bit_length() - 1, (P * tiles_n) & ~7
        em = E - 1
        if tile < (c8 << e_lg):
            i = tile >> 3
            c, e = (i >> e_lg) * 8 + (tile & 7), i & em
        else:
            c, e = tile >> e_lg, tile & em
        return ((c & ((1 << lg) - 1)) << e_lg) + e, c >> lg
    for tiles_m in (1, 2, 3, 4, 6, 8, 12, 16):
        for tiles_n in (1, 2, 7, 8, 9, 32, 64, 86, 112, 224):
            for ovr in (0, 1, 2, 4, 8):
                E = group_of(tiles_m, ovr)
                seen = {remap(t, tiles_m, tiles_n, E) for t in range(tiles_m * tiles_n)}
                assert len(seen) == tiles_m * tiles_n and all(0 <= a < tiles_m and 0 <= b < tiles_n for a, b in seen), (tiles_m, tiles_n, E)
    assert group_of(4) == 4 and group_of(2) == 2 and group_of(8) == 8 and group_of(6) == 1 and group_of(3) == 1 and group_of(12) == 1 and group_of(16) == 8
    # M = 256 on 4096 x 11008 (128-row tiles): the 22 (21) tiles of an XCD are 11-12 whole pairs
    per = {}
    for t in range(2 * 86):
        per.setdefault(t % 8, []).append(remap(t, 2, 86, 2))
    assert all(len({tn for _, tn in v}) <= len(v) // 2 + 1 for v in per.values())
    # M = 256 on 4096^2 (64 x 64 tiles, round 6): the 32 tiles of an XCD are the four row tiles of eight column tiles
    per = {}
    for t in range(4 * 64):
        per.setdefault(t % 8, []).append(remap(t, 4, 64, 4))
    assert all(len(v) == 32 and len({tn for _, tn in v}) == 8 for v in per.values())
