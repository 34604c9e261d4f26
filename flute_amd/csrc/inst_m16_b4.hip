// Explicit instantiations of the small-batch MFMA kernel (M <= 16) for num_bits = 4.
#include "kernels.h"
#include "qgemm_m16.h"
namespace flute_amd {
QGemmKernel m16_kernel_b4(int dtype, int tile_p, int r) {
    if (tile_p == 32 && r == 1) return dtype == 0 ? (QGemmKernel)qgemm_m16_kernel<F16, 4, 32, 1> : (QGemmKernel)qgemm_m16_kernel<BF16, 4, 32, 1>;
    if (tile_p == 32 && r == 2) return dtype == 0 ? (QGemmKernel)qgemm_m16_kernel<F16, 4, 32, 2> : (QGemmKernel)qgemm_m16_kernel<BF16, 4, 32, 2>;
    if (tile_p == 32 && r == 4) return dtype == 0 ? (QGemmKernel)qgemm_m16_kernel<F16, 4, 32, 4> : (QGemmKernel)qgemm_m16_kernel<BF16, 4, 32, 4>;
    if (tile_p == 64 && r == 1) return dtype == 0 ? (QGemmKernel)qgemm_m16_kernel<F16, 4, 64, 1> : (QGemmKernel)qgemm_m16_kernel<BF16, 4, 64, 1>;
    if (tile_p == 64 && r == 2) return dtype == 0 ? (QGemmKernel)qgemm_m16_kernel<F16, 4, 64, 2> : (QGemmKernel)qgemm_m16_kernel<BF16, 4, 64, 2>;
    if (tile_p == 64 && r == 4) return dtype == 0 ? (QGemmKernel)qgemm_m16_kernel<F16, 4, 64, 4> : (QGemmKernel)qgemm_m16_kernel<BF16, 4, 64, 4>;
    return nullptr;
}
}  // namespace flute_amd
