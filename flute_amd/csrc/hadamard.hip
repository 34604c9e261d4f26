// Fast Walsh-Hadamard transform for the HIGGS pre-rotation.
//
// Replaces the reference's tensor-core HadaCore kernel
// (flute/csrc/hadamard_transform_cuda.cu:92-662, dispatcher :701-748; host
// wrapper hadamard_transform.cpp:17-56): out = in.reshape(-1, h) @ (H_h/sqrt(h)),
// Sylvester order, h a power of two <= 2^15, fp16 / bf16.
//
// The op is HBM-bound (2*numel*2 B, log2(h) adds per element), so on CDNA4 it
// is a butterfly network, not a matmul: 8 elements per lane from one 16-B load
// (3 in-register stages), 6 cross-lane stages inside the wave64, and for
// h > 512 one LDS transpose that turns the remaining (<= 6) high bits into
// register/lane bits again.  All arithmetic is fp32, one rounding at the end
// (the reference's fp16 path accumulates in fp16 per 16x16 factor,
// hadamard_transform_cuda.cu:56 - parity for this op is tolerance-based).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.h"
#include "fwht.h"

namespace flute_amd {

template <typename T>
__device__ __forceinline__ void load8(const uint16_t* p, size_t base, size_t numel, float (&v)[8]) {
    if (base + 8 <= numel) {
        const uint4 t = *reinterpret_cast<const uint4*>(p + base);
        const uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[2 * i] = Num<T>::to_float((uint16_t)(w[i] & 0xffff));
            v[2 * i + 1] = Num<T>::to_float((uint16_t)(w[i] >> 16));
        }
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i)
            v[i] = (base + i < numel) ? Num<T>::to_float(p[base + i]) : 0.f;
    }
}

template <typename T>
__device__ __forceinline__ void store8(uint16_t* p, size_t base, size_t numel, const float (&v)[8],
                                       float scale) {
    if (base + 8 <= numel) {
        uint32_t w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
            w[i] = (uint32_t)scale_round<T>(v[2 * i], scale) | ((uint32_t)scale_round<T>(v[2 * i + 1], scale) << 16);
        *reinterpret_cast<uint4*>(p + base) = make_uint4(w[0], w[1], w[2], w[3]);
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (base + i < numel) p[base + i] = scale_round<T>(v[i], scale);
    }
}

// h <= 512: a wave transforms 512 contiguous elements = 512/h independent blocks.
template <typename T>
__global__ __launch_bounds__(256) void fwht_wave_kernel(const uint16_t* __restrict__ in,
                                                        uint16_t* __restrict__ out, size_t numel,
                                                        int log_h, float scale) {
    const size_t base = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 8;
    if (base >= numel) {
        // whole lane out of range; it still has to take part in the shuffles
    }
    float v[8];
    load8<T>(in, base < numel ? base : numel, numel, v);
    const int lane = threadIdx.x & 63;
    reg_stages(v, log_h < 3 ? log_h : 3);
    lane_stages(v, lane, log_h > 3 ? log_h - 3 : 0);
    if (base < numel) store8<T>(out, base, numel, v, scale);
}

// 512 < h <= 32768: one workgroup per block of h elements, R vectors per thread.
//   element index e = ((r*nthr + tid) << 3) | i
//   phase 1: bits 0..8 (register + lane), bits >= 9 + log2(nwaves) (the r bits)
//   LDS transpose: position bits [0, L-9) <-> index bits [9, L)
//   phase 2: the former wave bits, now register/lane bits
template <typename T, int R>
__global__ __launch_bounds__(1024) void fwht_block_kernel(const uint16_t* __restrict__ in,
                                                          uint16_t* __restrict__ out, size_t numel,
                                                          int log_h, float scale) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* lds = reinterpret_cast<float*>(smem);
    const int tid = threadIdx.x;
    const int nthr = blockDim.x;                      // h / (8 R)
    const int lane = tid & 63;
    const size_t blk = (size_t)blockIdx.x << log_h;
    const int hi_bits = log_h - 9;                    // 1..6

    float v[R][8];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        load8<T>(in, blk + ((size_t)(r * nthr + tid) << 3), numel, v[r]);
        reg_stages(v[r], 3);
        lane_stages(v[r], lane, 6);
    }
    // r bits are the top bits of e: butterflies between whole vectors
    if constexpr (R >= 2) {
#pragma unroll
        for (int s = 1; s < R; s <<= 1)
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (!(r & s)) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float a = v[r][i], b = v[r + s][i];
                        v[r][i] = a + b; v[r + s][i] = a - b;
                    }
                }
    }
    // remaining bits: 9 .. 9+log2(nwaves)-1  (count = hi_bits - log2(R))
    int rbits = 0;
    for (int t = R; t > 1; t >>= 1) ++rbits;
    const int wbits = hi_bits - rbits;

    auto phys = [](int e) { return e ^ ((e >> 9) & 31); };
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) lds[phys((((r * nthr + tid) << 3) | i))] = v[r][i];
    __syncthreads();

    const int ymask = (1 << hi_bits) - 1;
    if (wbits > 0) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float t[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int pos = ((r * nthr + tid) << 3) | i;
                const int e = ((pos & ymask) << 9) | (pos >> hi_bits);
                t[i] = lds[phys(e)];
            }
            // the wave bits sit at position bits [0, wbits); r bits (already done) above them
            reg_stages(t, wbits < 3 ? wbits : 3);
            lane_stages(t, lane, wbits > 3 ? wbits - 3 : 0);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int pos = ((r * nthr + tid) << 3) | i;
                const int e = ((pos & ymask) << 9) | (pos >> hi_bits);
                v[r][i] = t[i];
                (void)e;
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int pos = ((r * nthr + tid) << 3) | i;
                const int e = ((pos & ymask) << 9) | (pos >> hi_bits);
                lds[phys(e)] = v[r][i];
            }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) v[r][i] = lds[phys((((r * nthr + tid) << 3) | i))];
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
        store8<T>(out, blk + ((size_t)(r * nthr + tid) << 3), numel, v[r], scale);
}

template <typename T>
static int launch_fwht(const void* in, void* out, size_t numel, uint32_t h, hipStream_t stream) {
    int log_h = 0;
    while ((1u << log_h) < h) ++log_h;
    const float scale = 1.0f / sqrtf((float)h);
    const uint16_t* i16 = reinterpret_cast<const uint16_t*>(in);
    uint16_t* o16 = reinterpret_cast<uint16_t*>(out);
    if (h <= 512) {
        const size_t threads = (numel + 7) / 8;
        const unsigned grid = (unsigned)((threads + 255) / 256);
        hipLaunchKernelGGL((fwht_wave_kernel<T>), dim3(grid), dim3(256), 0, stream, i16, o16, numel,
                           log_h, scale);
    } else {
        if (numel % h) return -4;
        const unsigned grid = (unsigned)(numel / h);
        const size_t lds = (size_t)h * 4;
        if (h <= 8192) {
            auto k = fwht_block_kernel<T, 1>;
            if (lds > 65536) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(k, dim3(grid), dim3(h / 8), lds, stream, i16, o16, numel, log_h, scale);
        } else if (h == 16384) {
            auto k = fwht_block_kernel<T, 2>;
            (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(k, dim3(grid), dim3(1024), lds, stream, i16, o16, numel, log_h, scale);
        } else {
            auto k = fwht_block_kernel<T, 4>;
            (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(k, dim3(grid), dim3(1024), lds, stream, i16, o16, numel, log_h, scale);
        }
    }
    return hipGetLastError() == hipSuccess ? 0 : -6;
}

int hadamard_dispatch(int dtype, const void* in, void* out, size_t numel, uint32_t h,
                      hipStream_t stream) {
    if (h == 0 || (h & (h - 1)) || h > (1u << 15)) return -8;
    if (numel == 0) return 0;
    if (numel % h) return -4;
    if (dtype == 0) return launch_fwht<F16>(in, out, numel, h, stream);
    if (dtype == 1) return launch_fwht<BF16>(in, out, numel, h, stream);
    return -7;
}

}  // namespace flute_amd
