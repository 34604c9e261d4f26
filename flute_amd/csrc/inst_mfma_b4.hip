// Explicit instantiations of the MFMA kernel for num_bits = 4.
#include "kernels.h"
#include "qgemm_mfma.h"
namespace flute_amd {
QGemmKernel mfma_kernel_b4(int dtype, int tile_p, int mt) {
    if (tile_p == 32 && mt == 1) return dtype == 0 ? (QGemmKernel)qgemm_mfma_kernel<F16, 4, 32, 1> : (QGemmKernel)qgemm_mfma_kernel<BF16, 4, 32, 1>;
    if (tile_p == 32 && mt == 2) return dtype == 0 ? (QGemmKernel)qgemm_mfma_kernel<F16, 4, 32, 2> : (QGemmKernel)qgemm_mfma_kernel<BF16, 4, 32, 2>;
    if (tile_p == 32 && mt == 4) return dtype == 0 ? (QGemmKernel)qgemm_mfma_kernel<F16, 4, 32, 4> : (QGemmKernel)qgemm_mfma_kernel<BF16, 4, 32, 4>;
    if (tile_p == 64 && mt == 1) return dtype == 0 ? (QGemmKernel)qgemm_mfma_kernel<F16, 4, 64, 1> : (QGemmKernel)qgemm_mfma_kernel<BF16, 4, 64, 1>;
    if (tile_p == 64 && mt == 2) return dtype == 0 ? (QGemmKernel)qgemm_mfma_kernel<F16, 4, 64, 2> : (QGemmKernel)qgemm_mfma_kernel<BF16, 4, 64, 2>;
    if (tile_p == 64 && mt == 4) return dtype == 0 ? (QGemmKernel)qgemm_mfma_kernel<F16, 4, 64, 4> : (QGemmKernel)qgemm_mfma_kernel<BF16, 4, 64, 4>;
    return nullptr;
}
}  // namespace flute_amd
