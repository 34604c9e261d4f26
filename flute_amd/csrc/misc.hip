// Small kernels around the hot path: split-K second pass and the native unpacker.
#include "kernels.h"
#include "mfma.h"

namespace flute_amd {

int splitk_reduce_dispatch(int dtype, const float* partial, void* D, size_t mn, int splitk,
                           hipStream_t stream) {
    const unsigned grid = (unsigned)((mn / 4 + 255) / 256);
    uint16_t* d16 = reinterpret_cast<uint16_t*>(D);
    if (dtype == 0)
        hipLaunchKernelGGL((splitk_reduce_kernel<F16>), dim3(grid), dim3(256), 0, stream, partial,
                           d16, mn, splitk);
    else
        hipLaunchKernelGGL((splitk_reduce_kernel<BF16>), dim3(grid), dim3(256), 0, stream, partial,
                           d16, mn, splitk);
    return hipGetLastError() == hipSuccess ? 0 : -6;
}

// Q[P,K] -> codes W[K,N] (uint8).  The reference recovers codes by running
// qgemm on an identity matrix with an arange table (flute/utils.py:347-407);
// this reads the bit fields directly (layout: common.h).  One thread per
// (unit, kappa): consecutive threads take consecutive units so the byte
// stores to W[k, n0 + j*TileP + t] are contiguous in t.
template <int BITS, int TILEP>
__global__ __launch_bounds__(256) void unpack_kernel(const uint32_t* __restrict__ Q,
                                                     uint8_t* __restrict__ W, int N, int K) {
    using L = Layout<BITS>;
    constexpr int J = L::J;
    constexpr int NP = L::NPLANES;
    const int units = N / J;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int u = (int)(idx % units);
    const size_t kap = idx / units;
    if (kap >= (size_t)(K >> 1)) return;
    uint32_t w[NP];
#pragma unroll
    for (int pl = 0; pl < NP; ++pl)
        w[pl] = Q[(size_t)unit_row<BITS, TILEP>(u, pl, N) * (K >> 1) + kap];
    const int n0 = unit_col0<BITS, TILEP>(u);
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const uint32_t f = field<BITS>(w, j);
        W[(2 * kap) * N + n0 + j * TILEP] = (uint8_t)(f >> BITS);
        W[(2 * kap + 1) * N + n0 + j * TILEP] = (uint8_t)(f & ((1u << BITS) - 1));
    }
}

int unpack_dispatch(int num_bits, int tile_p, int N, int K, const void* Q, void* W,
                    hipStream_t stream) {
    const int J = (num_bits == 3) ? 16 : 16 / num_bits;
    const size_t total = (size_t)(N / J) * (K >> 1);
    const unsigned grid = (unsigned)((total + 255) / 256);
    const uint32_t* q = reinterpret_cast<const uint32_t*>(Q);
    uint8_t* w = reinterpret_cast<uint8_t*>(W);
#define FLUTE_UNPACK(B, TP) \
    hipLaunchKernelGGL((unpack_kernel<B, TP>), dim3(grid), dim3(256), 0, stream, q, w, N, K)
    if (num_bits == 4 && tile_p == 32) FLUTE_UNPACK(4, 32);
    else if (num_bits == 4 && tile_p == 64) FLUTE_UNPACK(4, 64);
    else if (num_bits == 2 && tile_p == 32) FLUTE_UNPACK(2, 32);
    else if (num_bits == 2 && tile_p == 64) FLUTE_UNPACK(2, 64);
    else if (num_bits == 3 && tile_p == 32) FLUTE_UNPACK(3, 32);
    else return -3;
#undef FLUTE_UNPACK
    return hipGetLastError() == hipSuccess ? 0 : -6;
}

// ---- calibration: what a pure read of `bytes` costs on this chip ---------------------
// Same access shape as the decode kernel (each wave streams contiguous KiB bursts,
// `U` dwordx4 loads in flight per lane) with one XOR per load instead of the dequant.
template <int U>
__global__ __launch_bounds__(1024) void stream_read_kernel(const uint4* __restrict__ src,
                                                           uint32_t* __restrict__ sink,
                                                           size_t n16_per_wave, int waves_total) {
    const int wave = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const int lane = threadIdx.x & 63;
    uint32_t acc = 0;
    for (int w = wave; w < waves_total; w += (int)((gridDim.x * blockDim.x) >> 6)) {
        const uint4* p = src + (size_t)w * n16_per_wave + lane;
        for (size_t i = 0; i < n16_per_wave; i += 64 * U) {
            uint4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = p[i + (size_t)u * 64];
#pragma unroll
            for (int u = 0; u < U; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;       // never true for random data; keeps the loads live
}

int stream_read_dispatch(const void* src, void* sink, size_t bytes, int bytes_per_wave, int grid,
                         int block, hipStream_t stream) {
    const size_t n16_per_wave = (size_t)bytes_per_wave / 16;
    const int waves_total = (int)(bytes / (size_t)bytes_per_wave);
    hipLaunchKernelGGL((stream_read_kernel<8>), dim3(grid), dim3(block), 0, stream,
                       reinterpret_cast<const uint4*>(src), reinterpret_cast<uint32_t*>(sink),
                       n16_per_wave, waves_total);
    return hipGetLastError() == hipSuccess ? 0 : -6;
}

// Measurement only: one lane writes the chip-wide 100 MHz clock (s_memrealtime) to *dst.  Captured into a hipGraph in front
// of and behind the timed launches, the difference brackets exactly those launches (bench.py): HIP events around a graph
// replay add a fixed 16 - 19 us per replay (profiles/r05/graph_replay_fixed_cost_probe.json).
__global__ void timestamp_kernel(unsigned long long* dst) {
    if (threadIdx.x == 0) *dst = (unsigned long long)wall_clock64();
}
int timestamp_dispatch(void* dst, hipStream_t stream) {
    hipLaunchKernelGGL(timestamp_kernel, dim3(1), dim3(64), 0, stream, reinterpret_cast<unsigned long long*>(dst));
    return hipGetLastError() == hipSuccess ? 0 : -6;
}

}  // namespace flute_amd
