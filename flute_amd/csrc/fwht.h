// Butterfly stages of the fast Walsh-Hadamard transform on 8 elements per lane x 64 lanes
// (Sylvester order: element index = lane * 8 + register).  Shared by the stand-alone transform
// (hadamard.hip) and by the decode kernel's fused pre-rotation of its activations.
#pragma once
#include "common.h"

namespace flute_amd {

// butterflies over the 3 register bits (only the lowest `nbits` of them)
__device__ __forceinline__ void reg_stages(float (&v)[8], int nbits) {
    if (nbits >= 1) {
#pragma unroll
        for (int i = 0; i < 8; i += 2) { const float a = v[i], b = v[i + 1]; v[i] = a + b; v[i + 1] = a - b; }
    }
    if (nbits >= 2) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (!(i & 2)) { const float a = v[i], b = v[i + 2]; v[i] = a + b; v[i + 2] = a - b; }
    }
    if (nbits >= 3) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { const float a = v[i], b = v[i + 4]; v[i] = a + b; v[i + 4] = a - b; }
    }
}

// Butterflies over `nbits` lane bits (lane bit s pairs lanes l and l ^ (1 << s)), without a trip through the LDS crossbar.
// Rounds 1 - 4 fetched the partner's value with __shfl_xor (ds_bpermute_b32): six dependent LDS round trips per transform,
// 1.4 us of a 5.3-us launch for the fused pre-rotation of ONE 3584-element row.  Round 5 (same fp32 operations on the same
// operands in the same order: results are bit-identical, tests/test_hadamard_gpu.py):
//   lane bits 0 - 3: ONE instruction per value and stage, v_fmac_f32_dpp v, v(partner), c with c = -1 in the lanes whose bit s
//     is set: those lanes hold v - p = -(p - v), the NEGATIVE of the butterfly's result; the pending sign is the same in both
//     lanes of every later pair (they differ in a higher bit only), so it commutes with the later stages and is applied once
//     at the end (parity of the lane's low bits: one v_xor per value).  Partner by DPP: quad_perm for bits 0 and 1, row_ror:8
//     for bit 3, and for bit 2 a quad reverse followed by row_half_mirror ((l ^ 3) ^ 7 = l ^ 4: one extra v_mov_dpp).
//   lane bits 4, 5: gfx950's lane swaps turn the lane stage into a register stage.  v_permlane16_swap a, b leaves a = rows
//     (a0, b0, a2, b2) and b = rows (a1, b1, a3, b3): both halves of a butterfly of value a now sit in rows 0 / 2 of the two
//     registers (those of value b in rows 1 / 3), a + b and a - b are the results, and the same swap puts them back.  Four
//     instructions per PAIR of values; v_permlane32_swap does the same for the wave's halves.
#define FLUTE_FWHT_OPS8 "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7])
#define FLUTE_FWHT_DPP_STAGE(CTRL)                                                                       \
    asm volatile("s_nop 1\n\t"                                                                           \
                 "v_fmac_f32_dpp %0, %0, %8 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                     \
                 "v_fmac_f32_dpp %1, %1, %8 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                     \
                 "v_fmac_f32_dpp %2, %2, %8 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                     \
                 "v_fmac_f32_dpp %3, %3, %8 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                     \
                 "v_fmac_f32_dpp %4, %4, %8 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                     \
                 "v_fmac_f32_dpp %5, %5, %8 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                     \
                 "v_fmac_f32_dpp %6, %6, %8 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                     \
                 "v_fmac_f32_dpp %7, %7, %8 " CTRL " row_mask:0xf bank_mask:0xf"                         \
                 : FLUTE_FWHT_OPS8 : "v"(c))

template <int S>
__device__ __forceinline__ void lane_stage(float (&v)[8], int lane) {
    if constexpr (S < 4) {
        const float c = ((lane >> S) & 1) ? -1.0f : 1.0f;
        if constexpr (S == 0) FLUTE_FWHT_DPP_STAGE("quad_perm:[1,0,3,2]");
        else if constexpr (S == 1) FLUTE_FWHT_DPP_STAGE("quad_perm:[2,3,0,1]");
        else if constexpr (S == 3) FLUTE_FWHT_DPP_STAGE("row_ror:8");
        else {
            float t[8];
            asm volatile("s_nop 1\n\t"
                         "v_mov_b32_dpp %0, %8 quad_perm:[3,2,1,0] row_mask:0xf bank_mask:0xf\n\t"
                         "v_mov_b32_dpp %1, %9 quad_perm:[3,2,1,0] row_mask:0xf bank_mask:0xf\n\t"
                         "v_mov_b32_dpp %2, %10 quad_perm:[3,2,1,0] row_mask:0xf bank_mask:0xf\n\t"
                         "v_mov_b32_dpp %3, %11 quad_perm:[3,2,1,0] row_mask:0xf bank_mask:0xf\n\t"
                         "v_mov_b32_dpp %4, %12 quad_perm:[3,2,1,0] row_mask:0xf bank_mask:0xf\n\t"
                         "v_mov_b32_dpp %5, %13 quad_perm:[3,2,1,0] row_mask:0xf bank_mask:0xf\n\t"
                         "v_mov_b32_dpp %6, %14 quad_perm:[3,2,1,0] row_mask:0xf bank_mask:0xf\n\t"
                         "v_mov_b32_dpp %7, %15 quad_perm:[3,2,1,0] row_mask:0xf bank_mask:0xf"
                         : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3]), "=&v"(t[4]), "=&v"(t[5]), "=&v"(t[6]), "=&v"(t[7])
                         : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]));
            asm volatile("s_nop 1\n\t"
                         "v_fmac_f32_dpp %0, %9, %8 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f32_dpp %1, %10, %8 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f32_dpp %2, %11, %8 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f32_dpp %3, %12, %8 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f32_dpp %4, %13, %8 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f32_dpp %5, %14, %8 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f32_dpp %6, %15, %8 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f32_dpp %7, %16, %8 row_half_mirror row_mask:0xf bank_mask:0xf"
                         : FLUTE_FWHT_OPS8
                         : "v"(c), "v"(t[0]), "v"(t[1]), "v"(t[2]), "v"(t[3]), "v"(t[4]), "v"(t[5]), "v"(t[6]), "v"(t[7]));
        }
    } else {
#define FLUTE_FWHT_SWAPS(OP)                                                                                           \
        asm volatile("s_nop 1\n\t" OP " %0, %1\n\t" OP " %2, %3\n\t" OP " %4, %5\n\t" OP " %6, %7\n\ts_nop 1" : FLUTE_FWHT_OPS8)
        if constexpr (S == 4) FLUTE_FWHT_SWAPS("v_permlane16_swap_b32"); else FLUTE_FWHT_SWAPS("v_permlane32_swap_b32");
#pragma unroll
        for (int i = 0; i < 8; i += 2) { const float a = v[i], b = v[i + 1]; v[i] = a + b; v[i + 1] = a - b; }
        if constexpr (S == 4) FLUTE_FWHT_SWAPS("v_permlane16_swap_b32"); else FLUTE_FWHT_SWAPS("v_permlane32_swap_b32");
#undef FLUTE_FWHT_SWAPS
    }
}
__device__ __forceinline__ void lane_stages(float (&v)[8], int lane, int nbits) {
    if (0 < nbits) lane_stage<0>(v, lane);
    if (1 < nbits) lane_stage<1>(v, lane);
    if (2 < nbits) lane_stage<2>(v, lane);
    if (3 < nbits) lane_stage<3>(v, lane);
    if (0 < nbits) {                                    // the pending signs of the DPP stages
        const uint32_t sg = (uint32_t)(__builtin_popcount(lane & ((1 << (nbits < 4 ? nbits : 4)) - 1)) & 1) << 31;
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, v[i]) ^ sg);
    }
    if (4 < nbits) lane_stage<4>(v, lane);
    if (5 < nbits) lane_stage<5>(v, lane);
}
#undef FLUTE_FWHT_DPP_STAGE
#undef FLUTE_FWHT_OPS8

// v * scale rounded to fp32, THEN to T: the product is made opaque so that hipcc cannot contract the
// multiply and the conversion into one mixed-precision instruction (v_fma_mixlo_f16: a single
// rounding) in one kernel and not in the other - the stand-alone transform and the fused one must
// round identically (bit-identical flute.qgemm_hadamard either way)
template <typename T>
__device__ __forceinline__ uint16_t scale_round(float v, float scale) {
    float t = v * scale;
    asm volatile("" : "+v"(t));
    return Num<T>::from_float(t);
}

// 8 packed T (one 16-B piece, element index = lane * 8 + i inside a 512-element span) transformed in
// place over blocks of 2^log_h <= 512 elements: fp32 butterflies, orthonormal scale, one rounding
template <typename T>
__device__ __forceinline__ void fwht_piece(uint32_t (&w)[4], int lane, int log_h, float scale) {
    float v[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[2 * i] = Num<T>::to_float((uint16_t)(w[i] & 0xffff));
        v[2 * i + 1] = Num<T>::to_float((uint16_t)(w[i] >> 16));
    }
    reg_stages(v, log_h < 3 ? log_h : 3);
    lane_stages(v, lane, log_h > 3 ? log_h - 3 : 0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
        w[i] = (uint32_t)scale_round<T>(v[2 * i], scale) | ((uint32_t)scale_round<T>(v[2 * i + 1], scale) << 16);
}

}  // namespace flute_amd
