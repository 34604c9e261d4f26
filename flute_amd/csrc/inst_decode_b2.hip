// Explicit instantiations of the decode kernel for num_bits = 2 (generated layout: one
// translation unit per bit width so that `make -j` compiles them in parallel).
#include "kernels.h"
#include "qgemm_decode.h"
namespace flute_amd {
QGemmKernel decode_kernel_b2(int dtype, int tile_p, int mb) {
    if (tile_p == 32 && mb == 1) return dtype == 0 ? (QGemmKernel)qgemv_kernel<F16, 2, 32, 1> : (QGemmKernel)qgemv_kernel<BF16, 2, 32, 1>;
    if (tile_p == 32 && mb == 2) return dtype == 0 ? (QGemmKernel)qgemv_kernel<F16, 2, 32, 2> : (QGemmKernel)qgemv_kernel<BF16, 2, 32, 2>;
    if (tile_p == 32 && mb == 4) return dtype == 0 ? (QGemmKernel)qgemv_kernel<F16, 2, 32, 4> : (QGemmKernel)qgemv_kernel<BF16, 2, 32, 4>;
    if (tile_p == 32 && mb == 8) return dtype == 0 ? (QGemmKernel)qgemv_kernel<F16, 2, 32, 8> : (QGemmKernel)qgemv_kernel<BF16, 2, 32, 8>;
    if (tile_p == 64 && mb == 1) return dtype == 0 ? (QGemmKernel)qgemv_kernel<F16, 2, 64, 1> : (QGemmKernel)qgemv_kernel<BF16, 2, 64, 1>;
    if (tile_p == 64 && mb == 2) return dtype == 0 ? (QGemmKernel)qgemv_kernel<F16, 2, 64, 2> : (QGemmKernel)qgemv_kernel<BF16, 2, 64, 2>;
    if (tile_p == 64 && mb == 4) return dtype == 0 ? (QGemmKernel)qgemv_kernel<F16, 2, 64, 4> : (QGemmKernel)qgemv_kernel<BF16, 2, 64, 4>;
    if (tile_p == 64 && mb == 8) return dtype == 0 ? (QGemmKernel)qgemv_kernel<F16, 2, 64, 8> : (QGemmKernel)qgemv_kernel<BF16, 2, 64, 8>;
    return nullptr;
}
}  // namespace flute_amd
