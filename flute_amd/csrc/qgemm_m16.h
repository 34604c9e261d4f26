// Small-batch MFMA kernel (5 <= M <= 16, also usable down to M = 1).
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
//   * K is split over the workgroup's waves and reduced through LDS.
#pragma once
#include "common.h"
#include "qgemm_mfma.h"

namespace flute_amd {

template <int BITS> struct M16Cfg {
    static constexpr int LUT_STRIDE = 256;
    static constexpr int LUT_BYTES = (1 << (2 * BITS)) * 256;     // 64 KB / 16 KB / 4 KB
    static constexpr int GB = 32;                                  // scale groups per staged block
    // ring depths in k-steps: weights must cover an HBM latency (unique bytes in flight per
    // wave = PFQ * 1 KiB / R), the A fragment comes from L2
    static constexpr int PFX = (BITS == 3) ? 2 : 8;
};

__host__ __device__ inline size_t m16_lds_bytes(int bits, int R, int waves) {
    const int J = (bits == 3) ? 16 : 16 / bits;
    const int nmf = J / R;
    size_t b = (size_t)(1 << (2 * bits)) * 256;
    b += (size_t)waves * nmf * 32 * 16 * 2;           // wave-private scale tables
    b += (size_t)waves * nmf * 1024;                  // cross-wave reduction (f32x4 per lane)
    return b;
}

template <typename T, int BITS, int TILEP, int R>
__global__ __launch_bounds__(512) void qgemm_m16_kernel(const QGemmArgs a) {
    using L = Layout<BITS>;
    using NT = Num<T>;
    using C = M16Cfg<BITS>;
    constexpr int J = L::J;
    constexpr int NP = L::NPLANES;
    constexpr int NMF = J / R;                 // MFMAs (column tiles) per k-step
    constexpr int SU = 16 / R;                 // units per slab
    constexpr int PFX = C::PFX;
    constexpr int PF = (BITS == 3) ? 2 : (R == 4 ? 16 : 8);      // weight ring
    static_assert(PF % PFX == 0, "A-fragment ring slot must be static");
    constexpr int GB = C::GB;
    constexpr bool PRE = __is_same(T, F16);
    static_assert(BITS != 3 || R == 1, "3-bit fields are not byte aligned: R = 1 only");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (lds_base_of(smem) != 0) __builtin_trap();

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
    const int g = 1 << lg;

    const int split = blockIdx.x % a.splitk;
    const int sg = blockIdx.x / a.splitk;
    const int slab = sg * ns + sl;
    const int u = slab * SU + r16 / R;                    // this lane's unit
    const int f = r16 % R;                                // which share of the fields
    const int kbeg = split * a.k_per_split;
    const int kend = min(a.K, kbeg + a.k_per_split);
    const int kpw = (((kend - kbeg + kw - 1) / kw) + 31) & ~31;
    const int kb = min(kend, kbeg + kpart * kpw);
    const int ke = min(kend, kb + kpw);
    const int nsteps = (ke - kb) >> 5;

    const uint32_t sc_base = C::LUT_BYTES + (uint32_t)wave * (NMF * GB * 16 * 2);
    const uint32_t red_base = C::LUT_BYTES + (uint32_t)nw * (NMF * GB * 16 * 2);

    const uint16_t* A = reinterpret_cast<const uint16_t*>(a.A);
    const uint16_t* S = reinterpret_cast<const uint16_t*>(a.S);
    const size_t row_words = (size_t)(a.K >> 1);
    const uint32_t* qrow[NP];
#pragma unroll
    for (int pl = 0; pl < NP; ++pl)
        qrow[pl] = a.Q + (size_t)unit_row<BITS, TILEP>(u, pl, a.N) * row_words + q4 * 4;
    const bool xrow_ok = r16 < a.M;
    const uint16_t* xrow = A + (size_t)min(r16, a.M - 1) * a.K + q4 * 8;

    // column of this lane in MFMA i: field j = i*R + f of unit u
    int ncol[NMF];
    uint32_t selv[NMF];
#pragma unroll
    for (int i = 0; i < NMF; ++i) {
        const int j = i * R + f;
        ncol[i] = unit_col0<BITS, TILEP>(u) + j * TILEP;
        selv[i] = 0x0c0c0400u | ((4u + (uint32_t)j) << 8);        // b=4: byte j of the word
    }
    const uint32_t lane_off = (uint32_t)lane * 4;

    // ---- rings: weights + A fragment, PF k-steps deep ----
    uint4 qr[PF][NP];
    uint4 xr[PFX];
#pragma unroll
    for (int t = 0; t < PF; ++t) {
        if (t < nsteps) {
            const int k = kb + t * 32;
#pragma unroll
            for (int pl = 0; pl < NP; ++pl)
                qr[t][pl] = *reinterpret_cast<const uint4*>(qrow[pl] + (k >> 1));
        }
    }
#pragma unroll
    for (int t = 0; t < PFX; ++t) {
        xr[t] = make_uint4(0, 0, 0, 0);
        if (t < nsteps && xrow_ok) xr[t] = *reinterpret_cast<const uint4*>(xrow + kb + t * 32);
    }

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

    // ---- wave-private scale table: block of GB groups, layout [i][group][column] (halves) ----
    uint16_t* scw = reinterpret_cast<uint16_t*>(smem + sc_base);
    auto stage_scales = [&](int gblk0) {
        // lane (c = lane&15, o = lane>>4) fetches groups [gblk0 + 8o, +8) of its column
        const int c = r16, o = q4;
#pragma unroll
        for (int i = 0; i < NMF; ++i) {
#pragma unroll
            for (int h = 0; h < GB / 32; ++h) {
                const int g0 = gblk0 + (h * 4 + o) * 8;
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
                for (int r = 0; r < 8; ++r)
                    scw[(i * GB + (h * 4 + o) * 8 + r) * 16 + c] = hv[r];
            }
        }
    };
    int gblk0 = (kb >> lg) & ~7;                 // first staged group (8-aligned for vector loads)
    if (nsteps > 0) stage_scales(gblk0);
    __syncthreads();                             // table + scales visible (only barrier before the epilogue)

    f32x4_t acc[NMF], run[NMF];
    uint32_t sreg[NMF];                          // current group's scale (raw T in the low half)
#pragma unroll
    for (int i = 0; i < NMF; ++i) {
        acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        run[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        sreg[i] = 0;
    }
    int cur_group = -1;

    for (int t0 = 0; t0 < nsteps; t0 += PF) {
#pragma unroll
        for (int s = 0; s < PF; ++s) {
            const int t = t0 + s;
            if (t < nsteps) {
                const int k0 = kb + t * 32;
                const int grp = k0 >> lg;
                if (grp != cur_group) {                                   // wave-uniform
                    if constexpr (!PRE) {
                        if (cur_group >= 0) {
#pragma unroll
                            for (int i = 0; i < NMF; ++i) {
                                const float sf = NT::to_float((uint16_t)sreg[i]);
#pragma unroll
                                for (int e = 0; e < 4; ++e) acc[i][e] = __builtin_fmaf(run[i][e], sf, acc[i][e]);
                                run[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                            }
                        }
                    }
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
                uint32_t w[NP][4];
#pragma unroll
                for (int pl = 0; pl < NP; ++pl) {
                    w[pl][0] = qr[s][pl].x; w[pl][1] = qr[s][pl].y; w[pl][2] = qr[s][pl].z; w[pl][3] = qr[s][pl].w;
                }
                const int sx = s % PFX;                         // static after unrolling
                const u32x4_t af = u32x4_t{xr[sx].x, xr[sx].y, xr[sx].z, xr[sx].w};
                // refill both rings
                if (t + PF < nsteps) {
                    const int kn = k0 + PF * 32;
#pragma unroll
                    for (int pl = 0; pl < NP; ++pl)
                        qr[s][pl] = *reinterpret_cast<const uint4*>(qrow[pl] + (kn >> 1));
                }
                if (t + PFX < nsteps && xrow_ok) xr[sx] = *reinterpret_cast<const uint4*>(xrow + k0 + PFX * 32);
#pragma unroll
                for (int i = 0; i < NMF; ++i) {
                    u32x4_t bf;
#pragma unroll
                    for (int ww = 0; ww < 4; ++ww) {
                        uint32_t addr;
                        if constexpr (BITS == 4) {
                            addr = __builtin_amdgcn_perm(w[0][ww], lane_off, selv[i]);
                        } else if constexpr (BITS == 2) {
                            const uint32_t idx = (R == 1) ? ((w[0][ww] >> (4 * i)) & 0xfu)
                                                          : __builtin_amdgcn_ubfe(w[0][ww], 4u * (uint32_t)(i * R + f), 4u);
                            addr = (idx << 8) | lane_off;
                        } else {
                            uint32_t wv[NP];
#pragma unroll
                            for (int pl = 0; pl < NP; ++pl) wv[pl] = w[pl][ww];
                            addr = (field<BITS>(wv, i) << 8) | lane_off;
                        }
                        const uint32_t v = lds_ld32(addr);
                        bf[ww] = PRE ? NT::mul_scale(v, sreg[i]) : v;
                    }
                    if constexpr (PRE) acc[i] = Mfma<T>::run(af, bf, acc[i]);
                    else run[i] = Mfma<T>::run(af, bf, run[i]);
                }
            }
        }
    }
    if constexpr (!PRE) {
        if (cur_group >= 0) {
#pragma unroll
            for (int i = 0; i < NMF; ++i) {
                const float sf = NT::to_float((uint16_t)sreg[i]);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][e] = __builtin_fmaf(run[i][e], sf, acc[i][e]);
            }
        }
    }

    // ---- reduce the kw partial tiles through LDS, write rows < M ----
    float* red = reinterpret_cast<float*>(smem + red_base);
    if (kw > 1) {
#pragma unroll
        for (int i = 0; i < NMF; ++i)
            *reinterpret_cast<f32x4_t*>(red + ((size_t)(wave * NMF + i) * 64 + lane) * 4) = acc[i];
        __syncthreads();
        if (kpart == 0) {
            for (int kp = 1; kp < kw; ++kp)
#pragma unroll
                for (int i = 0; i < NMF; ++i) {
                    const f32x4_t o = *reinterpret_cast<const f32x4_t*>(
                        red + ((size_t)((wave + kp) * NMF + i) * 64 + lane) * 4);
                    acc[i] += o;
                }
        }
    }
    if (kpart == 0) {
#pragma unroll
        for (int i = 0; i < NMF; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = q4 * 4 + e;
                if (row < a.M) {
                    if (a.splitk == 1)
                        reinterpret_cast<uint16_t*>(a.D)[(size_t)row * a.N + ncol[i]] = NT::from_float(acc[i][e]);
                    else
                        a.partial[((size_t)split * a.M + row) * a.N + ncol[i]] = acc[i][e];
                }
            }
    }
}

}  // namespace flute_amd
