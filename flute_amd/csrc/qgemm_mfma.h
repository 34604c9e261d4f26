// Batched / prefill kernel (M > 8): LUT-dequant straight into MFMA B fragments.
//
// Replaces qgemm_device's ldmatrix -> dequantize -> mma.sync m16n8k16 loop
// (flute/csrc/qgemm_kernel.hpp:617-712, config.hpp:323-325) with a wave64 /
// v_mfma_f32_16x16x32_{f16,bf16} design:
//   * a wave owns a slab of 16 units (16 Q32 rows, or 16 triples for b=3);
//     lane (r = lane&15, q = lane>>4) loads the 16 B of unit r that hold
//     k = k0 + 8q .. 8q+7.  That one dwordx4 is exactly the K-run an MFMA B
//     operand lane needs, for ALL J columns of the unit: 4 LUT lookups per
//     column turn it into J B-fragments with no cross-lane traffic
//     (the 3-bit planes are three such loads - no ds_bpermute needed);
//   * activations go through LDS in 256-B swizzled rows (slot ^= row&15, so a
//     ds_read_b128 lane group touches all 64 banks once) and are shared by
//     the workgroup's waves; per-group scales ride along, transposed so one
//     ds_read gives the J column scales of the lane's unit;
//   * weights are prefetched PF k-steps ahead in a register ring and their
//     loads stay in flight across the per-chunk barrier;
//   * fp32 accumulate; grid-level split-K writes fp32 slabs (splitk_reduce).
#pragma once
#include "common.h"

namespace flute_amd {

typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_mfma_t __attribute__((ext_vector_type(8)));

template <typename T> struct Mfma;
template <> struct Mfma<F16> {
    static __device__ __forceinline__ f32x4_t run(u32x4_t a, u32x4_t b, f32x4_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a),
                                                      __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    }
};
template <> struct Mfma<BF16> {
    static __device__ __forceinline__ f32x4_t run(u32x4_t a, u32x4_t b, f32x4_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_mfma_t, a),
                                                       __builtin_bit_cast(bf16x8_mfma_t, b), c, 0,
                                                       0, 0);
    }
};

template <int MT> struct MfmaCfg {
    static constexpr int KC = (MT >= 4) ? 128 : 256;   // k per staged chunk
    static constexpr int TM = MT * 16;
    static constexpr int PF = 4;                       // weight prefetch depth (k-steps)
};

__host__ __device__ inline size_t mfma_lds_bytes(int bits, int mt, int lg, int waves,
                                                 int lut_shift) {
    const int J = (bits == 3) ? 16 : 16 / bits;
    const int lut_n = 1 << (2 * bits);
    const int kc = (mt >= 4) ? 128 : 256;
    const int gcap = (kc >> lg) + 2;
    size_t b = (((size_t)lut_n << (lut_shift + 2)) + 15) & ~(size_t)15;
    b += (size_t)2 * mt * 16 * kc * 2;
    b += (size_t)2 * gcap * waves * 16 * J * 2;
    return b;
}

template <typename T, int BITS, int TILEP, int MT>
__global__ __launch_bounds__(256) void qgemm_mfma_kernel(const QGemmArgs a) {
    using L = Layout<BITS>;
    using NT = Num<T>;
    using C = MfmaCfg<MT>;
    constexpr int J = L::J;
    constexpr int NP = L::NPLANES;
    constexpr int LUT_N = L::LUT_N;
    constexpr int KC = C::KC;
    constexpr int TM = C::TM;
    constexpr int PF = C::PF;
    constexpr int NSTEP = KC / 32;
    constexpr int SLOTS = KC / 8;            // 16-B slots per activation row
    static_assert(NSTEP % PF == 0, "ring slot must be static");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int nthr = blockDim.x;
    const int lane = tid & 63;
    const int r16 = lane & 15;
    const int q4 = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nw = nthr >> 6;
    const int lg = a.lg;
    const int gcap = (KC >> lg) + 2;

    int bid = blockIdx.x;
    const int split = bid % a.splitk;  bid /= a.splitk;
    const int mtiles = (a.M + TM - 1) / TM;
    const int mtile = bid % mtiles;
    const int sg = bid / mtiles;
    const int m0 = mtile * TM;
    const int ubase = sg * nw * 16;                 // first unit of the workgroup
    const int u = ubase + wave * 16 + r16;          // this lane's unit
    const int kbeg = split * a.k_per_split;
    const int kend = min(a.K, kbeg + a.k_per_split);

    const int lsh = a.lut_shift;
    const size_t lut_bytes = (((size_t)LUT_N << (lsh + 2)) + 15) & ~(size_t)15;
    uint32_t* lut = reinterpret_cast<uint32_t*>(smem);
    uint16_t* xs = reinterpret_cast<uint16_t*>(smem + lut_bytes);
    uint16_t* ss = xs + (size_t)2 * TM * KC;

    const uint16_t* A = reinterpret_cast<const uint16_t*>(a.A);
    const uint16_t* S = reinterpret_cast<const uint16_t*>(a.S);

    if (lsh >= 2) {
        for (int e = tid; e < (LUT_N << (lsh - 2)); e += nthr) {
            const uint32_t v = a.QM2[e >> (lsh - 2)];
            reinterpret_cast<uint4*>(lut)[e] = make_uint4(v, v, v, v);
        }
    } else {
        for (int e = tid; e < (LUT_N << lsh); e += nthr) lut[e] = a.QM2[e >> lsh];
    }
    const uint32_t* lut_lane = lut + (lane & ((1 << lsh) - 1));

    const size_t row_words = (size_t)(a.K >> 1);
    const uint32_t* qrow[NP];
#pragma unroll
    for (int pl = 0; pl < NP; ++pl)
        qrow[pl] = a.Q + (size_t)unit_row<BITS, TILEP>(u, pl, a.N) * row_words + q4 * 4;

    // ---- weight prefetch ring ----
    uint4 qr[PF][NP];
#pragma unroll
    for (int t = 0; t < PF; ++t) {
        const int k = kbeg + t * 32;
        if (k < kend) {
#pragma unroll
            for (int pl = 0; pl < NP; ++pl)
                qr[t][pl] = *reinterpret_cast<const uint4*>(qrow[pl] + (k >> 1));
        }
    }

    f32x4_t acc[MT][J];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < J; ++j) acc[mt][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    int c = 0;
    for (int kc0 = kbeg; kc0 < kend; kc0 += KC, ++c) {
        const int buf = c & 1;
        const int kc_len = min(KC, kend - kc0);
        const int g0c = kc0 >> lg;
        const int gcnt = ((kc0 + kc_len - 1) >> lg) - g0c + 1;
        uint16_t* xsb = xs + (size_t)buf * TM * KC;
        uint16_t* ssb = ss + (size_t)buf * gcap * nw * 16 * J;

        // ---- stage activations, 16-B slots XOR-swizzled inside each 256-B row piece ----
        for (int p = tid; p < TM * SLOTS; p += nthr) {
            const int row = p / SLOTS;
            const int s = p - row * SLOTS;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (m0 + row < a.M && s * 8 < kc_len)
                v = *reinterpret_cast<const uint4*>(A + (size_t)(m0 + row) * a.K + kc0 + s * 8);
            const int ps = (s & ~15) | ((s ^ row) & 15);
            *reinterpret_cast<uint4*>(xsb + (size_t)row * KC + ps * 8) = v;
        }
        // ---- stage scales as [group][unit][J] ----
        for (int e = tid; e < gcnt * nw * 16 * J; e += nthr) {
            const int gl = e % gcnt;
            const int rr = e / gcnt;
            const int j = rr % J;
            const int ulc = rr / J;
            const int n = unit_col0<BITS, TILEP>(ubase + ulc) + j * TILEP;
            ssb[(gl * nw * 16 + ulc) * J + j] = S[(size_t)n * a.G + g0c + gl];
        }
        __syncthreads();

#pragma unroll
        for (int ks = 0; ks < NSTEP; ++ks) {
            const int k0 = kc0 + ks * 32;
            if (k0 < kend) {
                const int slot = ks % PF;
                uint32_t w[NP][4];
#pragma unroll
                for (int pl = 0; pl < NP; ++pl) {
                    w[pl][0] = qr[slot][pl].x; w[pl][1] = qr[slot][pl].y;
                    w[pl][2] = qr[slot][pl].z; w[pl][3] = qr[slot][pl].w;
                }
                // refill the ring slot (PF steps ahead; may belong to the next chunk)
                {
                    const int kn = k0 + PF * 32;
                    if (kn < kend) {
#pragma unroll
                        for (int pl = 0; pl < NP; ++pl)
                            qr[slot][pl] = *reinterpret_cast<const uint4*>(qrow[pl] + (kn >> 1));
                    }
                }
                // scales of this lane's unit for the group of k0
                const int gl = (k0 >> lg) - g0c;
                uint32_t sw[J / 2];
                {
                    const uint32_t* sp = reinterpret_cast<const uint32_t*>(
                        ssb + (gl * nw * 16 + wave * 16 + r16) * J);
                    if constexpr (J == 4) {
                        const uint2 t = *reinterpret_cast<const uint2*>(sp);
                        sw[0] = t.x; sw[1] = t.y;
                    } else {
#pragma unroll
                        for (int h = 0; h < J / 8; ++h) {
                            const uint4 t = reinterpret_cast<const uint4*>(sp)[h];
                            sw[4 * h + 0] = t.x; sw[4 * h + 1] = t.y;
                            sw[4 * h + 2] = t.z; sw[4 * h + 3] = t.w;
                        }
                    }
                }
                // A fragments (rows mt*16 + r16, k-run q4 of this step)
                u32x4_t af[MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const int row = mt * 16 + r16;
                    const int s = ks * 4 + q4;
                    const int ps = (s & ~15) | ((s ^ row) & 15);
                    const uint4 t = *reinterpret_cast<const uint4*>(xsb + (size_t)row * KC + ps * 8);
                    af[mt] = u32x4_t{t.x, t.y, t.z, t.w};
                }
#pragma unroll
                for (int j = 0; j < J; ++j) {
                    const uint32_t s = (j & 1) ? (sw[j >> 1] >> 16) : sw[j >> 1];
                    u32x4_t bf;
#pragma unroll
                    for (int ww = 0; ww < 4; ++ww) {
                        uint32_t wv[NP];
#pragma unroll
                        for (int pl = 0; pl < NP; ++pl) wv[pl] = w[pl][ww];
                        const uint32_t idx = field<BITS>(wv, j);
                        bf[ww] = NT::mul_scale(lut_lane[idx << lsh], s);
                    }
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        acc[mt][j] = Mfma<T>::run(af[mt], bf, acc[mt][j]);
                }
            }
        }
    }

    // ---- epilogue: C layout col = lane&15 (unit r16), row = 4*q4 + i ----
    const int n0 = unit_col0<BITS, TILEP>(u);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < J; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = m0 + mt * 16 + q4 * 4 + i;
                if (row < a.M) {
                    const int n = n0 + j * TILEP;
                    if (a.splitk == 1)
                        reinterpret_cast<uint16_t*>(a.D)[(size_t)row * a.N + n] =
                            NT::from_float(acc[mt][j][i]);
                    else
                        a.partial[((size_t)split * a.M + row) * a.N + n] = acc[mt][j][i];
                }
            }
}

// ---- split-K second pass: D = T(sum_s partial[s]) ---------------------------
template <typename T>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ partial,
                                                            uint16_t* __restrict__ D, size_t mn,
                                                            int splitk) {
    const size_t i4 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i4 >= mn) return;                      // mn is a multiple of 16 (N % 16 == 0)
    float4 s = *reinterpret_cast<const float4*>(partial + i4);
    for (int k = 1; k < splitk; ++k) {
        const float4 t = *reinterpret_cast<const float4*>(partial + (size_t)k * mn + i4);
        s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
    }
    ushort4 o;
    o.x = Num<T>::from_float(s.x); o.y = Num<T>::from_float(s.y);
    o.z = Num<T>::from_float(s.z); o.w = Num<T>::from_float(s.w);
    *reinterpret_cast<ushort4*>(D + i4) = o;
}

}  // namespace flute_amd
