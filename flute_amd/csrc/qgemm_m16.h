// Column-per-lane MFMA kernel: every M above the decode kernel's range
// (MT = 1: 5 <= M <= 16, the HBM-bound small batch; MT = 2/4: 32/64-row tiles for prefill).
//
// Second gfx950 replacement for qgemm_device (flute/csrc/qgemm_kernel.hpp:617-712)
// in the regime the reference serves with its TileM=16 templates
// (flute/codegen_utils.py:90-94).  One v_mfma_f32_16x16x32 consumes 16 output
// columns x 32 k; the packed format puts J columns in every 32-bit word, so a wave
// that wants 16 different columns has two choices:
//   R = 1   lane -> unit, all J fields of its words: J MFMAs per k-step, a slab of
//           16 units per wave (no redundant loads, coarse: N/J/16 slabs);
//   R > 1   R lanes share one unit's words and each takes J/R of its fields: J/R
//           MFMAs per k-step, slab = 16/R units - 4x more slabs at R = 4, which is
//           what lets a 4096-wide layer fill 256 CUs WITHOUT a grid-level K split
//           (and its second reduce launch); the shared 16-B loads are merged by the
//           texture unit, the cost is R x VGPR-write bandwidth.
// Design points (r01 profile: VALU ops cost a quad-cycle, LDS lookups ~1.3 per pair):
//   * main loop has no barrier: weights AND the A fragment (lane (m, q) wants
//     x[m, k0+8q .. +7], rows >= M stay zero) come straight from global/L2 through
//     4-deep register rings; scales sit in a wave-private LDS table;
//   * pair-table lookups use the 256-B-stride table + one v_perm_b32 per address
//     (per-lane selector when R > 1);
//   * fp16: w^ = round_T(lut * s) with v_pk_mul_f16 - the reference's exact
//     arithmetic (packbits_utils.hpp:139); bf16 has no packed multiply on gfx950,
//     so the group scale is applied to the fp32 MFMA result of each group run;
//   * K is split over the workgroup's waves and reduced through LDS;
//   * both rings are inline-asm loads with ONE counted s_waitcnt per k-step (see common.h:
//     hipcc drains compiler-visible rings with vmcnt(0) at every loop header); the loads are
//     unconditional - rows >= M and k-steps past the end are clamped, their results unused.
#pragma once
#include <type_traits>
#include <utility>

#include "common.h"
#include "qgemm_mfma.h"

namespace flute_amd {

template <int BITS> struct M16Cfg {
    static constexpr int LUT_STRIDE = 256;
    static constexpr int LUT_BYTES = (1 << (2 * BITS)) * 256;     // 64 KB / 16 KB / 4 KB
    static constexpr int GB = 32;                                  // scale groups per staged block
};

// ring depth D (k-steps in flight): per step a lane issues NP weight loads + MT A-fragment loads
__host__ __device__ constexpr int m16_depth(int bits, int R, int mt) {
    return (bits == 3) ? (mt == 1 ? 4 : 2) : (mt == 4 ? 4 : (R == 4 ? 16 : 8));
}

__host__ __device__ inline size_t m16_lds_bytes(int bits, int R, int waves) {
    const int J = (bits == 3) ? 16 : 16 / bits;
    const int nmf = J / R;
    size_t b = (size_t)(1 << (2 * bits)) * 256;
    b += (size_t)waves * nmf * 32 * 16 * 2;           // wave-private scale tables
    b += (size_t)waves * nmf * 1024;                  // cross-wave reduction (one 16-row tile at a time)
    return b;
}

template <typename T, int BITS, int TILEP, int R, int MT>
__global__ __launch_bounds__(512) void qgemm_m16_kernel(const QGemmArgs a) {
    using L = Layout<BITS>;
    using NT = Num<T>;
    using C = M16Cfg<BITS>;
    constexpr int J = L::J;
    constexpr int NP = L::NPLANES;
    constexpr int NMF = J / R;                 // MFMAs (column tiles) per k-step and row tile
    constexpr int SU = 16 / R;                 // units per slab
    constexpr int D = m16_depth(BITS, R, MT);  // ring depth
    constexpr int LPS = NP + MT;               // loads per k-step
    constexpr int GB = C::GB;
    constexpr bool PRE = __is_same(T, F16);
    static_assert(BITS != 3 || R == 1, "3-bit fields are not byte aligned: R = 1 only");
    static_assert((D - 1) * LPS < 64, "vmcnt is a 6-bit counter");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (lds_base_of(smem) != 0) __builtin_trap();
#ifdef FLUTE_STAMPS   // development build: 100 MHz wall-clock stamps per wave into the (unused) workspace
    uint64_t stamp[8];
    stamp[0] = wall_clock64();
#define FLUTE_STAMP(i) stamp[i] = wall_clock64()
#else
#define FLUTE_STAMP(i)
#endif

    const int tid = threadIdx.x;
    const int nthr = blockDim.x;
    const int lane = tid & 63;
    const int r16 = lane & 15;
    const int q4 = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nw = nthr >> 6;
    const int kw = a.kw;
    const int ns = nw / kw;
    const int sl = wave / kw;
    const int kpart = wave - sl * kw;
    const int lg = a.lg;

    int bid = blockIdx.x;
    const int split = bid % a.splitk;  bid /= a.splitk;
    const int mtiles = (a.M + MT * 16 - 1) / (MT * 16);
    const int mtile = bid % mtiles;
    const int sg = bid / mtiles;
    const int m0 = mtile * (MT * 16);
    const int slab = sg * ns + sl;
    const int u = slab * SU + r16 / R;                    // this lane's unit
    const int f = r16 % R;                                // which share of the fields
    const int kbeg = split * a.k_per_split;
    const int kend = min(a.K, kbeg + a.k_per_split);
    const int kpw = (((kend - kbeg + kw - 1) / kw) + 31) & ~31;
    const int kb = min(kend, kbeg + kpart * kpw);
    const int ke = min(kend, kb + kpw);
    const int nsteps = (ke - kb) >> 5;
    const int klast = (nsteps > 0) ? kb + (nsteps - 1) * 32 : min(kb, a.K - 32);   // clamp target

    const uint32_t sc_base = C::LUT_BYTES + (uint32_t)wave * (NMF * GB * 16 * 2);
    const uint32_t red_base = C::LUT_BYTES + (uint32_t)nw * (NMF * GB * 16 * 2);

    const uint16_t* A = reinterpret_cast<const uint16_t*>(a.A);
    const uint16_t* S = reinterpret_cast<const uint16_t*>(a.S);
    const size_t row_words = (size_t)(a.K >> 1);
    const uint32_t* qrow[NP];
#pragma unroll
    for (int pl = 0; pl < NP; ++pl)
        qrow[pl] = a.Q + (size_t)unit_row<BITS, TILEP>(u, pl, a.N) * row_words + q4 * 4;
    // A-fragment rows of this lane (row >= M: clamped; that accumulator row is never stored)
    const uint16_t* xrow[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
        xrow[mt] = A + (size_t)min(m0 + mt * 16 + r16, a.M - 1) * a.K + q4 * 8;

    // column of this lane in column tile i: field j = i*R + f of unit u
    int ncol[NMF];
    uint32_t selv[NMF];
#pragma unroll
    for (int i = 0; i < NMF; ++i) {
        const int j = i * R + f;
        ncol[i] = unit_col0<BITS, TILEP>(u) + j * TILEP;
        selv[i] = 0x0c0c0400u | ((4u + (uint32_t)j) << 8);        // b=4: byte j of the word
    }
    const uint32_t lane_off = (uint32_t)lane * 4;

    // ---- rings: per k-step NP weight loads then MT A-fragment loads, D steps deep ----
    ring16_t qr[D][NP];
    ring16_t xr[D][MT];
    auto ring_issue = [&](int slot, int t) {
        const int k = (t < nsteps) ? kb + t * 32 : klast;
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) qr[slot][pl] = ring_load16(qrow[pl] + (k >> 1));
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) xr[slot][mt] = ring_load16(xrow[mt] + k);
    };
#pragma unroll
    for (int t = 0; t < D; ++t) ring_issue(t, t);
    FLUTE_STAMP(4);

    // ---- pair table (stride 256 B; 64 copies of 4 B) ----
    {
        constexpr int ENT = 1 << (2 * BITS);
        for (int p = tid; p < ENT * 4; p += nthr) {
            const uint32_t v = a.QM2[p >> 2];
            uint4* d = reinterpret_cast<uint4*>(smem + (size_t)(p >> 2) * 256 + (p & 3) * 64);
            const uint4 vv = make_uint4(v, v, v, v);
            d[0] = vv; d[1] = vv; d[2] = vv; d[3] = vv;
        }
    }

    FLUTE_STAMP(5);
    // ---- wave-private scale table: block of GB groups, layout [i][group][column] (halves) ----
    uint16_t* scw = reinterpret_cast<uint16_t*>(smem + sc_base);
    auto stage_scales = [&](int gblk0) {
        // lane (c = lane&15, o = lane>>4) fetches groups [gblk0 + 8o, +8) of its column
        const int c = r16, o = q4;
#pragma unroll
        for (int i = 0; i < NMF; ++i) {
            const int g0 = gblk0 + o * 8;
            const uint16_t* sp = S + (size_t)ncol[i] * a.G + g0;
            uint16_t hv[8];
            if (g0 + 8 <= a.G && ((reinterpret_cast<uintptr_t>(sp) & 15) == 0)) {
                const uint4 t = *reinterpret_cast<const uint4*>(sp);
                hv[0] = t.x & 0xffff; hv[1] = t.x >> 16; hv[2] = t.y & 0xffff; hv[3] = t.y >> 16;
                hv[4] = t.z & 0xffff; hv[5] = t.z >> 16; hv[6] = t.w & 0xffff; hv[7] = t.w >> 16;
            } else {
#pragma unroll
                for (int r = 0; r < 8; ++r) hv[r] = (g0 + r < a.G) ? sp[r] : (uint16_t)0;
            }
#pragma unroll
            for (int r = 0; r < 8; ++r) scw[(i * GB + o * 8 + r) * 16 + c] = hv[r];
        }
    };
    int gblk0 = (kb >> lg) & ~7;                 // first staged group (8-aligned for vector loads)
    if (nsteps > 0) stage_scales(gblk0);
    FLUTE_STAMP(6);
    __syncthreads();                             // table + scales visible (only barrier before the epilogue)
    __builtin_amdgcn_s_waitcnt(0x0F70);          // hipcc: nothing of yours is outstanding (see decode kernel)
    FLUTE_STAMP(1);

    f32x4_t acc[MT][NMF], run[MT][NMF];
    uint32_t sreg[NMF];                          // current group's scale (raw T in the low half)
#pragma unroll
    for (int i = 0; i < NMF; ++i) {
        sreg[i] = 0;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            acc[mt][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            run[mt][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        }
    }
    int cur_group = -1;
    auto fold_run = [&]() {
#pragma unroll
        for (int i = 0; i < NMF; ++i) {
            const float sf = NT::to_float((uint16_t)sreg[i]);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[mt][i][e] = __builtin_fmaf(run[mt][i][e], sf, acc[mt][i][e]);
                run[mt][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            }
        }
    };

    // One k-step: wait for slot s (at most NWAIT younger loads outstanding), compute, optionally
    // refill.  The LAST ring turn issues no refills, so its wait counts shrink statically.
    auto kstep = [&](auto s_tag, auto last_tag, int t) {
        constexpr int s = decltype(s_tag)::value;
        constexpr bool LAST = decltype(last_tag)::value;
        constexpr int NWAIT = LAST ? (D - 1 - s) * LPS : (D - 1) * LPS;
        if constexpr (NP == 1 && MT == 1) ring_wait<NWAIT>(qr[s][0], xr[s][0]);
        else if constexpr (NP == 1 && MT == 2) ring_wait<NWAIT>(qr[s][0], xr[s][0], xr[s][1]);
        else if constexpr (NP == 1 && MT == 4) { ring_wait<NWAIT>(qr[s][0], xr[s][0], xr[s][1]); ring_wait<NWAIT>(xr[s][2], xr[s][3]); }
        else if constexpr (NP == 3 && MT == 1) { ring_wait<NWAIT>(qr[s][0], qr[s][1], qr[s][2]); ring_wait<NWAIT>(xr[s][0]); }
        else { ring_wait<NWAIT>(qr[s][0], qr[s][1], qr[s][2]); ring_wait<NWAIT>(xr[s][0], xr[s][1]); }
        if (t < nsteps) {
            const int k0 = kb + t * 32;
            const int grp = k0 >> lg;
            if (grp != cur_group) {                                   // wave-uniform
                if constexpr (!PRE) { if (cur_group >= 0) fold_run(); }
                if (grp >= gblk0 + GB) {                              // next block of scales
                    gblk0 = grp & ~7;
                    stage_scales(gblk0);
                }
#pragma unroll
                for (int i = 0; i < NMF; ++i)
                    sreg[i] = *reinterpret_cast<const uint16_t*>(
                        smem + sc_base + (uint32_t)((i * GB + (grp - gblk0)) * 16 + r16) * 2);
                cur_group = grp;
            }
            u32x4_t af[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) af[mt] = u32x4_t{xr[s][mt].x, xr[s][mt].y, xr[s][mt].z, xr[s][mt].w};
#pragma unroll
            for (int i = 0; i < NMF; ++i) {
                u32x4_t bf;
#pragma unroll
                for (int ww = 0; ww < 4; ++ww) {
                    uint32_t addr;
                    if constexpr (BITS == 4) {
                        addr = __builtin_amdgcn_perm(qr[s][0][ww], lane_off, selv[i]);
                    } else if constexpr (BITS == 2) {
                        const uint32_t idx = (R == 1) ? ((qr[s][0][ww] >> (4 * i)) & 0xfu)
                                                      : __builtin_amdgcn_ubfe(qr[s][0][ww], 4u * (uint32_t)(i * R + f), 4u);
                        addr = (idx << 8) | lane_off;
                    } else {
                        uint32_t wv[NP];
#pragma unroll
                        for (int pl = 0; pl < NP; ++pl) wv[pl] = qr[s][pl][ww];
                        addr = (field<BITS>(wv, i) << 8) | lane_off;
                    }
                    const uint32_t v = lds_ld32(addr);
                    bf[ww] = PRE ? NT::mul_scale(v, sreg[i]) : v;
                }
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    if constexpr (PRE) acc[mt][i] = Mfma<T>::run(af[mt], bf, acc[mt][i]);
                    else run[mt][i] = Mfma<T>::run(af[mt], bf, run[mt][i]);
                }
            }
        }
        if constexpr (!LAST) ring_issue(s, t + D);     // unconditional; clamped if past the end
    };
    auto ring_turn = [&](auto last_tag, int t0) {
        [&]<int... S>(std::integer_sequence<int, S...>) {
            (kstep(std::integral_constant<int, S>{}, last_tag, t0 + S), ...);
        }(std::make_integer_sequence<int, D>{});
    };
    const int nround = (nsteps + D - 1) / D * D;         // whole ring turns: slot index stays static
    int t0 = 0;
    for (; t0 + D < nround; t0 += D) ring_turn(std::false_type{}, t0);
    ring_turn(std::true_type{}, t0);                     // also the only turn when nsteps <= D
    if constexpr (!PRE) { if (cur_group >= 0) fold_run(); }
    FLUTE_STAMP(2);

    // ---- reduce the kw partial tiles through LDS (one 16-row tile at a time), write rows < M ----
    float* red = reinterpret_cast<float*>(smem + red_base);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        if (kw > 1) {
            if (mt > 0) __syncthreads();          // previous tile's readers are done
#pragma unroll
            for (int i = 0; i < NMF; ++i)
                *reinterpret_cast<f32x4_t*>(red + ((size_t)(wave * NMF + i) * 64 + lane) * 4) = acc[mt][i];
            __syncthreads();
            if (kpart == 0) {
                for (int kp = 1; kp < kw; ++kp)
#pragma unroll
                    for (int i = 0; i < NMF; ++i) {
                        const f32x4_t o = *reinterpret_cast<const f32x4_t*>(
                            red + ((size_t)((wave + kp) * NMF + i) * 64 + lane) * 4);
                        acc[mt][i] += o;
                    }
            }
        }
        if (kpart == 0) {
#pragma unroll
            for (int i = 0; i < NMF; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int row = m0 + mt * 16 + q4 * 4 + e;
                    if (row < a.M) {
                        if (a.splitk == 1)
                            reinterpret_cast<uint16_t*>(a.D)[(size_t)row * a.N + ncol[i]] = NT::from_float(acc[mt][i][e]);
                        else
                            a.partial[((size_t)split * a.M + row) * a.N + ncol[i]] = acc[mt][i][e];
                    }
                }
        }
    }
#ifdef FLUTE_STAMPS
    __builtin_amdgcn_s_waitcnt(0);               // stores retired
    FLUTE_STAMP(3);
    if (lane == 0 && a.splitk == 1 && a.partial != nullptr) {
        uint64_t* o = reinterpret_cast<uint64_t*>(a.partial) + ((size_t)blockIdx.x * nw + wave) * 8;
        for (int i = 0; i < 8; ++i) o[i] = stamp[i];
    }
#endif
#undef FLUTE_STAMP
}

}  // namespace flute_amd
