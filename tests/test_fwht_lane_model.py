"""CPU model of the Hadamard transform's lane stages as csrc/fwht.h runs them since round 5 (no GPU needed).

The kernel's butterflies over the six lane bits of a 512-element block (element = lane * 8 + register) used to fetch the partner
lane's value through the LDS crossbar (`__shfl_xor`).  Now: lane bits 0-3 are ONE `v_fmac_f32_dpp v, v(partner), c` per value and
stage with c = -1 in the lanes whose bit is set - those lanes then hold the NEGATIVE of the butterfly's result, a pending sign that is
equal in both lanes of every later pair and is applied once (parity of the lane's low bits); the partner of bit 2 is reached by a quad
reverse followed by `row_half_mirror`; lane bits 4 and 5 are register stages between two lane swaps (`v_permlane16_swap` /
`v_permlane32_swap`).  The model below restates exactly that data movement with numpy float32 arithmetic and checks it, bit for bit,
against the textbook butterflies - the property `tests/test_qgemm_gpu.py::test_hadamard_*` then checks on the hardware against the
definition (reference: flute/csrc/hadamard/hadamard_transform_cuda.cu:141-154, the staged butterflies of the CUDA kernel)."""
import numpy as np

L = np.arange(64)
F = np.float32


def textbook(v, nbits):
    v = v.copy()
    for s in range(nbits):
        p = v[L ^ (1 << s)]
        hi = ((L >> s) & 1).astype(bool)[:, None]
        v = np.where(hi, p - v, v + p)
    return v


def permlane_swap(a, b, width):
    """v_permlane{16,32}_swap a, b: the odd `width`-lane groups of a trade places with the even groups of b."""
    odd = ((L // width) & 1).astype(bool)
    na = np.where(odd, b[L ^ width], a)
    nb = np.where(odd, b, a[L ^ width])
    return na, nb


def kernel_form(v, nbits):
    v = v.copy()
    for s in range(min(nbits, 4)):
        c = np.where((L >> s) & 1, -1.0, 1.0).astype(F)[:, None]
        if s == 2:
            partner = v[L ^ 3][L ^ 7]          # quad_perm [3,2,1,0], then row_half_mirror: (l ^ 3) ^ 7 = l ^ 4
        else:
            partner = v[L ^ (1 << s)]          # quad_perm [1,0,3,2] / [2,3,0,1], row_ror:8
        v = (v + partner * c).astype(F)        # v_fmac_f32: one rounding, the product by +-1 is exact
    low = (1 << min(nbits, 4)) - 1
    neg = np.array([bin(l & low).count("1") & 1 for l in L], dtype=bool)[:, None]
    v = np.where(neg, -v, v)                   # the pending signs: one v_xor per value
    for s in (4, 5):
        if s < nbits:
            for i in range(0, 8, 2):
                a, b = permlane_swap(v[:, i], v[:, i + 1], 1 << s)
                a, b = (a + b).astype(F), (a - b).astype(F)
                v[:, i], v[:, i + 1] = permlane_swap(a, b, 1 << s)
    return v


def test_lane_stages_equal_the_textbook_butterflies_bit_for_bit():
    rng = np.random.default_rng(5)
    for nbits in range(7):
        for scale in (1.0, 1e-3, 3e4):
            v = (rng.standard_normal((64, 8)) * scale).astype(F)
            want, got = textbook(v, nbits), kernel_form(v, nbits)
            assert want.dtype == got.dtype == F
            assert np.array_equal(want.view(np.uint32), got.view(np.uint32)), nbits


def test_partner_permutations():
    # the DPP controls of fwht.h: every one of them is the involution l -> l ^ (1 << s) on the 64 lanes
    assert np.array_equal((L ^ 3) ^ 7, L ^ 4)
    for s in range(6):
        p = L ^ (1 << s)
        assert np.array_equal(p[p], L) and np.all((p >> 4) == (L >> 4) if s < 4 else True)       # bits 0-3 stay inside a DPP row of 16
    a, b = np.arange(64), np.arange(64) + 100
    for width in (16, 32):
        na, nb = permlane_swap(a, b, width)
        ra, rb = permlane_swap(na, nb, width)
        assert np.array_equal(ra, a) and np.array_equal(rb, b)                                   # the swap is its own inverse
        assert sorted(np.concatenate([na, nb]).tolist()) == sorted(np.concatenate([a, b]).tolist())
