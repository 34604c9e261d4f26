// Explicit instantiations of the small-batch MFMA kernel (M <= 16) for num_bits = 3.
#include "kernels.h"
#include "qgemm_m16.h"
namespace flute_amd {
QGemmKernel m16_kernel_b3(int dtype, int tile_p, int r) {
    if (tile_p == 32 && r == 1) return dtype == 0 ? (QGemmKernel)qgemm_m16_kernel<F16, 3, 32, 1> : (QGemmKernel)qgemm_m16_kernel<BF16, 3, 32, 1>;
    return nullptr;
}
}  // namespace flute_amd
