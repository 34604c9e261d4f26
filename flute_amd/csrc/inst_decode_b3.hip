// Explicit instantiations of the decode kernel for num_bits = 3 (generated layout: one
// translation unit per bit width so that `make -j` compiles them in parallel).
#include "kernels.h"
#include "qgemm_decode.h"
namespace flute_amd {
QGemmKernel decode_kernel_b3(int dtype, int tile_p, int mb) {
    if (tile_p == 32 && mb == 1) return dtype == 0 ? (QGemmKernel)qgemv_kernel<F16, 3, 32, 1> : (QGemmKernel)qgemv_kernel<BF16, 3, 32, 1>;
    if (tile_p == 32 && mb == 2) return dtype == 0 ? (QGemmKernel)qgemv_kernel<F16, 3, 32, 2> : (QGemmKernel)qgemv_kernel<BF16, 3, 32, 2>;
    if (tile_p == 32 && mb == 4) return dtype == 0 ? (QGemmKernel)qgemv_kernel<F16, 3, 32, 4> : (QGemmKernel)qgemv_kernel<BF16, 3, 32, 4>;
    return nullptr;
}
}  // namespace flute_amd
