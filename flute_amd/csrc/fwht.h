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

// butterflies over `nbits` lane bits (lane bit s pairs lanes l and l^(1<<s))
__device__ __forceinline__ void lane_stages(float (&v)[8], int lane, int nbits) {
#pragma unroll
    for (int s = 0; s < 6; ++s) {
        if (s < nbits) {
            const bool hi = (lane >> s) & 1;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float p = __shfl_xor(v[i], 1 << s, 64);
                v[i] = hi ? (p - v[i]) : (v[i] + p);
            }
        }
    }
}

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
