// Small kernels around the hot path: split-K second pass and the native unpacker.
#include "kernels.h"
#include "qgemm_mfma.h"

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

}  // namespace flute_amd
