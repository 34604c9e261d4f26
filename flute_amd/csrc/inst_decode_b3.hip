// Explicit instantiations of the decode kernel for num_bits = 3 (one translation unit per
// bit width so that `make -j` compiles them in parallel).  PRE (per-pair scale rounding,
// the reference's exact contract) exists for fp16 only.
#include "kernels.h"
#include "qgemm_decode.h"
namespace flute_amd {
QGemmKernel decode_kernel_b3(int dtype, int tile_p, int mb, int pre) {
    if (tile_p == 32 && mb == 1) {
        if (dtype == 0) return pre ? (QGemmKernel)qgemv_kernel<F16, 3, 32, 1, true> : (QGemmKernel)qgemv_kernel<F16, 3, 32, 1, false>;
        return (QGemmKernel)qgemv_kernel<BF16, 3, 32, 1, false>;
    }
    if (tile_p == 32 && mb == 2) {
        if (dtype == 0) return pre ? (QGemmKernel)qgemv_kernel<F16, 3, 32, 2, true> : (QGemmKernel)qgemv_kernel<F16, 3, 32, 2, false>;
        return (QGemmKernel)qgemv_kernel<BF16, 3, 32, 2, false>;
    }
    return nullptr;
}
}  // namespace flute_amd
