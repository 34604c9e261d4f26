// Shared device-side definitions for the gfx950 qgemm kernels.
//
// Packed weight format (the reference's wire format, kept bit-for-bit so that
// weights packed by flute/utils.py:59-253 run unchanged).  View Q[P,K] int16
// as Q32[P,K/2] uint32 (k = 2*kappa, 2*kappa+1 little endian):
//   b in {4,2}, J = 16/b, row p = nb*TileP + t:
//       field j (bits [2b*j, 2b*j+2b)) is column n = nb*J*TileP + j*TileP + t,
//       high b bits = code of W[2kappa, n], low b bits = W[2kappa+1, n]
//   b = 3 (TileP = 32), P1 = N/16, unit (nb, t):
//       w0 = Q32[nb*32+t], w1 = Q32[P1+nb*64+t], w2 = Q32[P1+nb*64+32+t]
//       j < 15: 6-bit field (w[j%3] >> 6*(j/3)) & 63;  j == 15: the three
//       2-bit tops of w0,w1,w2 (low to high); n = nb*512 + j*32 + t
// (reader in the reference: flute/csrc/packbits_utils.hpp:92-140, 322-362).
//
// A "unit" is the set of Q32 rows that decode together: one row for b=2/4,
// the (w0,w1,w2) triple for b=3.  One unit covers J columns (4 / 8 / 16).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace flute_amd {

typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
typedef __bf16 b2_t __attribute__((ext_vector_type(2)));
typedef short bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

struct F16 {};   // dtype id 0
struct BF16 {};  // dtype id 1

// ---- arithmetic contract (packbits_utils.hpp:139, :344-361) ---------------
//   w^ = round_T(table2[field] * scale)   one rounding, in T, both halves
//   acc (fp32) += x * w^                  (config.hpp:323-325, kMixed)

template <typename T> struct Num;

template <> struct Num<F16> {
    // pair (uint32 holding two fp16) times a scalar fp16 (low 16 bits of s)
    static __device__ __forceinline__ uint32_t mul_scale(uint32_t v, uint32_t s) {
        h2_t a = __builtin_bit_cast(h2_t, v);
        h2_t b = __builtin_bit_cast(h2_t, s);
        h2_t bb = {b.x, b.x};
        return __builtin_bit_cast(uint32_t, a * bb);           // v_pk_mul_f16 op_sel
    }
    static __device__ __forceinline__ void mul_scale4(const uint32_t (&v)[4], uint32_t s, uint32_t (&out)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) out[i] = mul_scale(v[i], s);
    }
    static __device__ __forceinline__ float dot2(uint32_t a, uint32_t b, float c) {
        return __builtin_amdgcn_fdot2(__builtin_bit_cast(h2_t, a),
                                      __builtin_bit_cast(h2_t, b), c, false);
    }
    // the same with a zero accumulator as an inline constant (hipcc otherwise zeroes the register first: v_mov + v_dot2c)
    static __device__ __forceinline__ float dot2z(uint32_t a, uint32_t b) {
        float r;
        asm("v_dot2_f32_f16 %0, %1, %2, 0" : "=v"(r) : "v"(a), "v"(b));
        return r;
    }
    static __device__ __forceinline__ uint16_t from_float(float f) {
        _Float16 h = (_Float16)f;
        return __builtin_bit_cast(uint16_t, h);
    }
    static __device__ __forceinline__ float to_float(uint16_t u) {
        return (float)__builtin_bit_cast(_Float16, u);
    }
};

template <> struct Num<BF16> {
    static __device__ __forceinline__ uint32_t mul_scale(uint32_t v, uint32_t s) {
        // bf16 x bf16 is exact in fp32; one RNE rounding back to bf16
        float lo = __builtin_bit_cast(float, v << 16);
        float hi = __builtin_bit_cast(float, v & 0xffff0000u);
        float sf = __builtin_bit_cast(float, s << 16);
        b2_t r = {(__bf16)(lo * sf), (__bf16)(hi * sf)};        // v_cvt_pk_bf16_f32
        return __builtin_bit_cast(uint32_t, r);
    }
    // The four words of a column tile times one scale (the block kernels' multiply stage), round 5: the two fp32 products of a word come
    // straight out of the packed word - v_dot2_f32_bf16 against (s, 0) and (0, s): lo * s + hi * 0 and lo * 0 + hi * s, exact (one
    // term is zero; the pair tables hold finite values) - then one v_cvt_pk_bf16_f32: 3 VALU per word where unpack / multiply / convert
    // takes 4 (gfx950 has no packed bf16 multiply).  ONE asm block: a DOT result read by another VALU instruction needs three wait
    // states on gfx90a+ and hipcc's hazard recognizer does not look into asm statements (a per-word version with the conversion right
    // behind its products computed wrong values in one kernel: caught by test_random_vs_oracle_all_M) - the eight products first, the
    // conversions behind them in the same order, so that every conversion is >= 3 instructions behind its second product; the closing
    // s_nop keeps a following MFMA off the last conversion's result.
    static __device__ __forceinline__ void mul_scale4(const uint32_t (&v)[4], uint32_t s, uint32_t (&out)[4]) {
        const uint32_t s_lo = s & 0xffffu, s_hi = s << 16;
        float p0, p1, p2, p3, p4, p5, p6, p7;
        asm("v_dot2_f32_bf16 %4, %12, %16, 0\n\t"
            "v_dot2_f32_bf16 %5, %12, %17, 0\n\t"
            "v_dot2_f32_bf16 %6, %13, %16, 0\n\t"
            "v_dot2_f32_bf16 %7, %13, %17, 0\n\t"
            "v_dot2_f32_bf16 %8, %14, %16, 0\n\t"
            "v_dot2_f32_bf16 %9, %14, %17, 0\n\t"
            "v_dot2_f32_bf16 %10, %15, %16, 0\n\t"
            "v_dot2_f32_bf16 %11, %15, %17, 0\n\t"
            "v_cvt_pk_bf16_f32 %0, %4, %5\n\t"
            "v_cvt_pk_bf16_f32 %1, %6, %7\n\t"
            "v_cvt_pk_bf16_f32 %2, %8, %9\n\t"
            "v_cvt_pk_bf16_f32 %3, %10, %11\n\t"
            "s_nop 1"
            : "=&v"(out[0]), "=&v"(out[1]), "=&v"(out[2]), "=&v"(out[3]),
              "=&v"(p0), "=&v"(p1), "=&v"(p2), "=&v"(p3), "=&v"(p4), "=&v"(p5), "=&v"(p6), "=&v"(p7)
            : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(s_lo), "v"(s_hi));
    }
    static __device__ __forceinline__ float dot2(uint32_t a, uint32_t b, float c) {
        return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(b2_t, a),
                                               __builtin_bit_cast(b2_t, b), c, false);
    }
    static __device__ __forceinline__ float dot2z(uint32_t a, uint32_t b) {
        float r;
        asm("v_dot2_f32_bf16 %0, %1, %2, 0" : "=v"(r) : "v"(a), "v"(b));
        return r;
    }
    static __device__ __forceinline__ uint16_t from_float(float f) {
        __bf16 h = (__bf16)f;
        return __builtin_bit_cast(uint16_t, h);
    }
    static __device__ __forceinline__ float to_float(uint16_t u) {
        return __builtin_bit_cast(float, (uint32_t)u << 16);
    }
};

// ---- layout traits ---------------------------------------------------------

template <int BITS> struct Layout {
    static constexpr int J = (BITS == 3) ? 16 : 16 / BITS;   // columns per unit
    static constexpr int NPLANES = (BITS == 3) ? 3 : 1;       // Q32 rows per unit
    static constexpr int FIELD_BITS = 2 * BITS;
    static constexpr int LUT_N = 1 << FIELD_BITS;              // pair-table entries
};

// first column of unit u (columns are n0 + j*TILEP)
template <int BITS, int TILEP>
__device__ __forceinline__ int unit_col0(int u) {
    constexpr int J = Layout<BITS>::J;
    return (u / TILEP) * (J * TILEP) + (u % TILEP);
}

// Q32 row of plane `pl` of unit u
template <int BITS, int TILEP>
__device__ __forceinline__ int unit_row(int u, int pl, int N) {
    if constexpr (BITS != 3) {
        return u;
    } else {
        if (pl == 0) return u;
        const int P1 = N >> 4;
        return P1 + (u >> 5) * 64 + (pl - 1) * 32 + (u & 31);
    }
}

// pair-table index of column j from the unit's words (same kappa in each plane)
template <int BITS>
__device__ __forceinline__ uint32_t field(const uint32_t (&w)[Layout<BITS>::NPLANES], int j) {
    if constexpr (BITS == 4) {
        return (w[0] >> (8 * j)) & 0xffu;
    } else if constexpr (BITS == 2) {
        return (w[0] >> (4 * j)) & 0xfu;
    } else {
        if (j < 15) return (w[j % 3] >> (6 * (j / 3))) & 63u;
        return (w[0] >> 30) | ((w[1] >> 28) & 0xcu) | ((w[2] >> 26) & 0x30u);
    }
}

// ---- kernel arguments --------------------------------------------------------

struct QGemmArgs {
    const void* A;          // [M,K]   T
    const uint32_t* Q;      // [P,K/2] packed
    void* D;                // [M,N]   T
    const void* S;          // [N,G]   T
    const uint32_t* QM2;    // [4^b]   pair table (two T per word)
    float* partial;         // [splitk][M][N] fp32 when splitk > 1
    int M, N, K, G;
    int lg;                 // log2(group_size)
    int units;              // N / J
    int splitk;             // grid split of K
    int k_per_split;        // multiple of 64
    int kw;                 // waves of one workgroup that share a unit (split K inside the WG)
    int m0;                 // first row of A/D this launch handles (M-blocking of the decode kernel)
    int lut_shift;          // log2(LDS replicas of the pair table): 5, 4, 3 or 0 (MFMA kernel)
    int lds_budget;         // dynamic LDS the launch was sized for (decode kernel carve)
    int lkw;                // log2(kw) (kw and the waves per workgroup are powers of two)
    int had_log;            // decode kernel: log2 of the fused Hadamard block (0 = none, <= 9)
    float had_scale;        // 2^(-had_log/2)
    // launch geometry computed once by the host planner (integer divisions cost the kernels'
    // prologue ~0.1 us each).  decode: kc, nbuf, gcap, log2(upw), x_off, s_off, red_off, log2(kc),
    // unit groups / workgroups (quotient, remainder); MFMA: depth, scale_bytes, slot_bytes,
    // wave_bytes, row-tile count, XCD-aware block order flag
    int geo[10];
    // grid-level K split combined INSIDE the launch (xwg.h, L form; qgemm_tile.h since round 4): two state words per
    // (slab group, row tile) of the grid; nullptr: fp32 slabs + the reduce launch
    uint32_t* state;
};

// ---- LDS access by absolute byte address -----------------------------------------
// `extern __shared__` is a link-time symbol: indexing it makes hipcc emit a
// `v_add_u32 addr, <symbol>, addr` per data-dependent access even though the symbol
// is 0.  Hot loops therefore address LDS through address_space(3) integer pointers.
typedef __attribute__((address_space(3))) const uint32_t lds_cu32_t;
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4v_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const u32x2_t lds_cu32x2_t;
typedef __attribute__((address_space(3))) const u32x4v_t lds_cu32x4_t;

__device__ __forceinline__ uint32_t lds_ld32(uint32_t a) { return *(lds_cu32_t*)(uintptr_t)a; }
__device__ __forceinline__ uint2 lds_ld64(uint32_t a) {
    const u32x2_t v = *(lds_cu32x2_t*)(uintptr_t)a;
    return make_uint2(v.x, v.y);
}
__device__ __forceinline__ uint4 lds_ld128(uint32_t a) {
    const u32x4v_t v = *(lds_cu32x4_t*)(uintptr_t)a;
    return make_uint4(v.x, v.y, v.z, v.w);
}
// byte address of the dynamic LDS segment (0 unless the kernel has static LDS)
__device__ __forceinline__ uint32_t lds_base_of(char* smem) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
}

// ---- weight ring: loads hipcc must not count --------------------------------------
// hipcc cannot carry vmcnt counts over a loop back-edge and drains the ring with
// s_waitcnt vmcnt(0) at every loop header (r01: compute and HBM time ADDED instead of
// overlapping).  The ring therefore uses inline-asm loads, invisible to the compiler's
// bookkeeping, and explicit counted waits that name the destination registers
// (cdna_hip_programming.md 5.7 form ii).  Loads retire in order, so compiler-issued
// loads in between can only make a counted wait conservative, never unsafe.
// The ring registers must never be copied between the load and its wait (the data is
// not there yet), so ring slots are kept in this vector type from load to use.
typedef uint32_t ring16_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ ring16_t ring_load16(const void* p) {
    ring16_t v;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    return v;
}
// streamed-once data (the packed weights): non-temporal, so the stream does not evict the
// activations / scales that every workgroup re-reads from L2
__device__ __forceinline__ ring16_t ring_load16_nt(const void* p) {
    ring16_t v;
#ifndef FLUTE_NT     // measured on MI355X (r01): nt on the weight stream is neutral to slightly negative
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
#else
    asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(v) : "v"(p) : "memory");
#endif
    return v;
}
__device__ __forceinline__ uint32_t ring_load4(const void* p) {
    uint32_t v;
    asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    return v;
}
// wait until at most N vector-memory ops are outstanding; the listed registers become
// valid here (the "+v" ties every later use of them behind the wait)
template <int N> __device__ __forceinline__ void ring_wait(ring16_t& a) {
    asm volatile("s_waitcnt vmcnt(%1)" : "+v"(a) : "n"(N) : "memory");
}
template <int N> __device__ __forceinline__ void ring_wait(ring16_t& a, ring16_t& b) {
    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N) : "memory");
}
template <int N> __device__ __forceinline__ void ring_wait(ring16_t& a, ring16_t& b, ring16_t& c) {
    asm volatile("s_waitcnt vmcnt(%3)" : "+v"(a), "+v"(b), "+v"(c) : "n"(N) : "memory");
}

template <int N> __device__ __forceinline__ void ring_wait4(uint32_t& a, uint32_t& b) {
    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N) : "memory");
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

}  // namespace flute_amd
