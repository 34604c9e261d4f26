// Explicit instantiations of the MFMA kernel for num_bits = 3.
#include "kernels.h"
#include "qgemm_mfma.h"
namespace flute_amd {
QGemmKernel mfma_kernel_b3(int dtype, int tile_p, int mt) {
    if (tile_p == 32 && mt == 1) return dtype == 0 ? (QGemmKernel)qgemm_mfma_kernel<F16, 3, 32, 1> : (QGemmKernel)qgemm_mfma_kernel<BF16, 3, 32, 1>;
    if (tile_p == 32 && mt == 2) return dtype == 0 ? (QGemmKernel)qgemm_mfma_kernel<F16, 3, 32, 2> : (QGemmKernel)qgemm_mfma_kernel<BF16, 3, 32, 2>;
    return nullptr;
}
}  // namespace flute_amd
