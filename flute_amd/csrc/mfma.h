// MFMA wrappers (v_mfma_f32_16x16x32_{f16,bf16}; fp32 accumulate as config.hpp:323-325, kMixed) and the
// second pass of a grid-level K split.
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
