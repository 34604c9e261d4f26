// Decode-regime kernel (M <= 8): HBM-bound streaming LUT-dequant GEMV.
//
// Replaces, for small M, the reference's qgemm_device main loop
// (flute/csrc/qgemm_kernel.hpp:617-712) + Stream-K fixup
// (tile_scheduler_utils.hpp:58-211).  CDNA4 design, not a translation:
//   * one wave streams ONE unit (one Q32 row, or the 3-plane triple for b=3)
//     along K: lane l of wave-instruction i reads the 16 B at byte l*16 of the
//     i-th KiB -> every global_load_dwordx4 is a fully coalesced 1 KiB burst,
//     straight to VGPRs (weights are used once: no LDS round trip);
//   * the 4^b-entry pair table is replicated 32x in LDS so that lane l always
//     hits bank l%32: the data-dependent ds_read_b32 is conflict free;
//   * per-group scales and the activation rows are staged in LDS per 4096-k
//     chunk (one ds_read_b64/b128 gives the J column scales of a line);
//   * w^ = round_T(lut * scale) with v_pk_mul_f16 (bf16: fp32 mul + RNE cvt),
//     accumulated in fp32 with v_dot2c_f32_{f16,bf16};
//   * K is split over `kw` waves of the workgroup and reduced through LDS; a
//     grid-level split (splitk) writes fp32 slabs reduced by splitk_reduce.
#pragma once
#include "common.h"

namespace flute_amd {

// k per staged chunk: 4096 for M<=2, then halved per doubling of the row block so
// that the double-buffered activation stage stays <= 32 KB
__host__ __device__ constexpr int dec_kc(int mb) { return mb <= 2 ? 4096 : 8192 / mb; }

template <int BITS> struct DecBatch { static constexpr int U = (BITS == 3) ? 4 : 8; };

// bytes of dynamic LDS the kernel needs (host and device use the same formula)
__host__ __device__ inline size_t decode_lds_bytes(int bits, int mb, int lg, int waves, int kw,
                                                   int krange, int lut_shift) {
    const int J = (bits == 3) ? 16 : 16 / bits;
    const int lut_n = 1 << (2 * bits);
    const int kc = dec_kc(mb);
    const int nbuf = (krange > kc) ? 2 : 1;
    const int upw = waves / kw;
    const int gcap = (kc >> lg) + 1;
    size_t b = (((size_t)lut_n << (lut_shift + 2)) + 15) & ~(size_t)15;   // replicated pair table
    b += (size_t)nbuf * mb * kc * 2;                     // activations
    b += (((size_t)nbuf * gcap * upw * J * 2) + 15) & ~(size_t)15;   // scales
    b += (size_t)waves * J * mb * 4;                     // cross-wave reduction
    return b;
}

template <typename T, int BITS, int TILEP, int MB>
__global__ __launch_bounds__(512) void qgemv_kernel(const QGemmArgs a) {
    using L = Layout<BITS>;
    using NT = Num<T>;
    constexpr int J = L::J;
    constexpr int NP = L::NPLANES;
    constexpr int LUT_N = L::LUT_N;
    constexpr int U = DecBatch<BITS>::U;
    constexpr int KC = dec_kc(MB);

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int nthr = blockDim.x;
    const int lane = tid & 63;
    const int sub = lane & 7;            // 16-B piece inside the 128-B line
    const int oct = lane >> 3;           // which of the 8 lines of a wave-instruction
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nw = nthr >> 6;
    const int kw = a.kw;
    const int upw = nw / kw;
    const int ul = wave / kw;
    const int kpart = wave - ul * kw;

    const int split = blockIdx.x % a.splitk;
    const int ug = blockIdx.x / a.splitk;
    const int u = ug * upw + ul;
    const int kbeg = split * a.k_per_split;
    const int kend = min(a.K, kbeg + a.k_per_split);
    const int krange = kend - kbeg;
    const int nbuf = (a.k_per_split > KC) ? 2 : 1;
    const int lg = a.lg;
    const int gcap = (KC >> lg) + 1;

    // ---- LDS carve (all offsets multiples of 16) ----
    const int lsh = a.lut_shift;
    const size_t lut_bytes = (((size_t)LUT_N << (lsh + 2)) + 15) & ~(size_t)15;
    uint32_t* lut = reinterpret_cast<uint32_t*>(smem);
    uint16_t* xs = reinterpret_cast<uint16_t*>(smem + lut_bytes);
    uint16_t* ss = xs + (size_t)nbuf * MB * KC;
    const size_t ss_bytes = (((size_t)nbuf * gcap * upw * J * 2) + 15) & ~(size_t)15;
    float* red = reinterpret_cast<float*>(reinterpret_cast<char*>(ss) + ss_bytes);

    const uint16_t* A = reinterpret_cast<const uint16_t*>(a.A);
    const uint16_t* S = reinterpret_cast<const uint16_t*>(a.S);

    // ---- replicated pair table: entry e occupies the (4 << lsh) bytes at e << (lsh+2) ----
    if (lsh >= 2) {
        for (int e = tid; e < (LUT_N << (lsh - 2)); e += nthr) {
            const uint32_t v = a.QM2[e >> (lsh - 2)];
            reinterpret_cast<uint4*>(lut)[e] = make_uint4(v, v, v, v);
        }
    } else {
        for (int e = tid; e < (LUT_N << lsh); e += nthr) lut[e] = a.QM2[e >> lsh];
    }

    const size_t row_words = (size_t)(a.K >> 1);
    const uint32_t* qrow[NP];
#pragma unroll
    for (int pl = 0; pl < NP; ++pl)
        qrow[pl] = a.Q + (size_t)unit_row<BITS, TILEP>(u, pl, a.N) * row_words;

    float acc[J][MB];
#pragma unroll
    for (int j = 0; j < J; ++j)
#pragma unroll
        for (int m = 0; m < MB; ++m) acc[j][m] = 0.f;

    const uint32_t* lut_lane = lut + (lane & ((1 << lsh) - 1));

    int c = 0;
    for (int kc0 = kbeg; kc0 < kend; kc0 += KC, ++c) {
        const int buf = (nbuf == 2) ? (c & 1) : 0;
        const int kc_len = min(KC, kend - kc0);
        const int Lc = kc_len >> 6;                    // 64-k lines in this chunk
        const int Lw = (Lc + kw - 1) / kw;             // lines per wave
        const int l0 = kpart * Lw;
        const int myL = max(0, min(Lw, Lc - l0));
        const int g0c = kc0 >> lg;
        const int gcnt = ((kc0 + kc_len - 1) >> lg) - g0c + 1;

        uint16_t* xsb = xs + (size_t)buf * MB * KC;
        uint16_t* ssb = ss + (size_t)buf * gcap * upw * J;

        // first batch of weight loads goes out before anything waits
        uint4 q[U][NP];
        const int nI = (myL + 7) >> 3;
        auto load_batch = [&](int ib) {
#pragma unroll
            for (int i = 0; i < U; ++i) {
                const int ln = (ib + i) * 8 + oct;
                if (ln < myL) {
                    const size_t woff = (size_t)((kc0 + (l0 + ln) * 64) >> 1) + sub * 4;
#pragma unroll
                    for (int pl = 0; pl < NP; ++pl)
                        q[i][pl] = *reinterpret_cast<const uint4*>(qrow[pl] + woff);
                }
            }
        };
        load_batch(0);

        // ---- stage activations (rows clamped to M-1, zero past kc_len) ----
        for (int p = tid; p < MB * (KC / 8); p += nthr) {
            const int m = p / (KC / 8);
            const int kk = (p - m * (KC / 8)) * 8;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (kk < kc_len) {
                const int row = min(a.m0 + m, a.M - 1);
                v = *reinterpret_cast<const uint4*>(A + (size_t)row * a.K + kc0 + kk);
            }
            *reinterpret_cast<uint4*>(xsb + (size_t)m * KC + kk) = v;
        }
        // ---- stage scales transposed to [group][unit][J] ----
        for (int e = tid; e < gcnt * upw * J; e += nthr) {
            const int gl = e % gcnt;
            const int r = e / gcnt;
            const int j = r % J;
            const int ulc = r / J;
            const int n = unit_col0<BITS, TILEP>(ug * upw + ulc) + j * TILEP;
            ssb[(gl * upw + ulc) * J + j] = S[(size_t)n * a.G + g0c + gl];
        }
        __syncthreads();

        for (int ib = 0; ib < nI; ib += U) {
            if (ib > 0) load_batch(ib);
#pragma unroll
            for (int i = 0; i < U; ++i) {
                const int ln = (ib + i) * 8 + oct;
                if (ln < myL) {
                    const int kl = (l0 + ln) * 64 + sub * 8;          // k inside chunk
                    const int gl = ((kc0 + kl) >> lg) - g0c;
                    // scales of the J columns for this k
                    uint32_t sw[J / 2];
                    {
                        const uint32_t* sp =
                            reinterpret_cast<const uint32_t*>(ssb + (gl * upw + ul) * J);
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
                    uint4 x[MB];
#pragma unroll
                    for (int m = 0; m < MB; ++m)
                        x[m] = *reinterpret_cast<const uint4*>(xsb + (size_t)m * KC + kl);

#pragma unroll
                    for (int ww = 0; ww < 4; ++ww) {
                        uint32_t w[NP];
#pragma unroll
                        for (int pl = 0; pl < NP; ++pl)
                            w[pl] = reinterpret_cast<const uint32_t*>(&q[i][pl])[ww];
#pragma unroll
                        for (int j = 0; j < J; ++j) {
                            const uint32_t idx = field<BITS>(w, j);
                            const uint32_t v = lut_lane[idx << lsh];
                            const uint32_t s = (j & 1) ? (sw[j >> 1] >> 16) : sw[j >> 1];
                            const uint32_t ws = NT::mul_scale(v, s);
#pragma unroll
                            for (int m = 0; m < MB; ++m)
                                acc[j][m] = NT::dot2(
                                    ws, reinterpret_cast<const uint32_t*>(&x[m])[ww], acc[j][m]);
                        }
                    }
                }
            }
        }
    }

    // ---- reduce: lanes -> wave -> (kw waves) -> output ----
#pragma unroll
    for (int j = 0; j < J; ++j)
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            const float v = wave_sum(acc[j][m]);
            if (lane == 0) red[wave * (J * MB) + j * MB + m] = v;
        }
    __syncthreads();
    for (int t = tid; t < upw * J * MB; t += nthr) {
        const int ulc = t / (J * MB);
        const int r = t - ulc * (J * MB);
        const int j = r / MB;
        const int m = r - j * MB;
        float sum = 0.f;
        for (int kp = 0; kp < kw; ++kp) sum += red[(ulc * kw + kp) * (J * MB) + r];
        const int row = a.m0 + m;
        if (row < a.M) {
            const int n = unit_col0<BITS, TILEP>(ug * upw + ulc) + j * TILEP;
            if (a.splitk == 1)
                reinterpret_cast<uint16_t*>(a.D)[(size_t)row * a.N + n] = NT::from_float(sum);
            else
                a.partial[((size_t)split * a.M + row) * a.N + n] = sum;
        }
    }
}

}  // namespace flute_amd
