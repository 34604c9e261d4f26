// Explicit instantiations of the decode kernel for num_bits = 4 (one translation unit per
// bit width so that `make -j` compiles them in parallel).  PRE (per-pair scale rounding,
// the reference's exact contract) exists for fp16 only.
#include "kernels.h"
#include "qgemm_decode.h"
namespace flute_amd {
QGemmKernel decode_kernel_b4(int dtype, int tile_p, int mb, int pre) {
    // ablation builds (tools/sweep.py ablate): pre = 100 + DBG
    if (pre == 101) return (QGemmKernel)qgemv_kernel<F16, 4, 32, 1, false, 1>;
    if (pre == 102) return (QGemmKernel)qgemv_kernel<F16, 4, 32, 1, false, 2>;
    if (pre == 103) return (QGemmKernel)qgemv_kernel<F16, 4, 32, 1, false, 3>;
    pre = pre > 0;
    if (tile_p == 32 && mb == 1) {
        if (dtype == 0) return pre ? (QGemmKernel)qgemv_kernel<F16, 4, 32, 1, true> : (QGemmKernel)qgemv_kernel<F16, 4, 32, 1, false>;
        return (QGemmKernel)qgemv_kernel<BF16, 4, 32, 1, false>;
    }
    if (tile_p == 32 && mb == 2) {
        if (dtype == 0) return pre ? (QGemmKernel)qgemv_kernel<F16, 4, 32, 2, true> : (QGemmKernel)qgemv_kernel<F16, 4, 32, 2, false>;
        return (QGemmKernel)qgemv_kernel<BF16, 4, 32, 2, false>;
    }
    if (tile_p == 32 && mb == 4) {
        if (dtype == 0) return pre ? (QGemmKernel)qgemv_kernel<F16, 4, 32, 4, true> : (QGemmKernel)qgemv_kernel<F16, 4, 32, 4, false>;
        return (QGemmKernel)qgemv_kernel<BF16, 4, 32, 4, false>;
    }
    if (tile_p == 64 && mb == 1) {
        if (dtype == 0) return pre ? (QGemmKernel)qgemv_kernel<F16, 4, 64, 1, true> : (QGemmKernel)qgemv_kernel<F16, 4, 64, 1, false>;
        return (QGemmKernel)qgemv_kernel<BF16, 4, 64, 1, false>;
    }
    if (tile_p == 64 && mb == 2) {
        if (dtype == 0) return pre ? (QGemmKernel)qgemv_kernel<F16, 4, 64, 2, true> : (QGemmKernel)qgemv_kernel<F16, 4, 64, 2, false>;
        return (QGemmKernel)qgemv_kernel<BF16, 4, 64, 2, false>;
    }
    if (tile_p == 64 && mb == 4) {
        if (dtype == 0) return pre ? (QGemmKernel)qgemv_kernel<F16, 4, 64, 4, true> : (QGemmKernel)qgemv_kernel<F16, 4, 64, 4, false>;
        return (QGemmKernel)qgemv_kernel<BF16, 4, 64, 4, false>;
    }
    return nullptr;
}
}  // namespace flute_amd
