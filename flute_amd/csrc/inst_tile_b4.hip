// Explicit instantiations of the LDS-DMA staged MFMA kernel for num_bits = 4 (R lanes share a
// unit, MT 16-row tiles per wave; (16/4... J/R)*MT <= 16 accumulator tiles).
#include "kernels.h"
#include "qgemm_tile.h"
namespace flute_amd {
QGemmKernel tile_kernel_b4(int dtype, int tile_p, int r, int mt, int sw) {
    if (sw == 2) {      // two slabs per wave: fp16 MT = 1 / 2 / 4, bf16 MT = 1 / 2
        if (r != 1) return nullptr;
        if (tile_p == 32 && mt == 1) return dtype == 0 ? (QGemmKernel)qgemm_tile_kernel<F16, 4, 32, 1, 1, 2> : (QGemmKernel)qgemm_tile_kernel<BF16, 4, 32, 1, 1, 2>;
        if (tile_p == 64 && mt == 1) return dtype == 0 ? (QGemmKernel)qgemm_tile_kernel<F16, 4, 64, 1, 1, 2> : (QGemmKernel)qgemm_tile_kernel<BF16, 4, 64, 1, 1, 2>;
        if (tile_p == 32 && mt == 2) return dtype == 0 ? (QGemmKernel)qgemm_tile_kernel<F16, 4, 32, 1, 2, 2> : (QGemmKernel)qgemm_tile_kernel<BF16, 4, 32, 1, 2, 2>;
        if (tile_p == 64 && mt == 2) return dtype == 0 ? (QGemmKernel)qgemm_tile_kernel<F16, 4, 64, 1, 2, 2> : (QGemmKernel)qgemm_tile_kernel<BF16, 4, 64, 1, 2, 2>;
        if (tile_p == 32 && mt == 4 && dtype == 0) return (QGemmKernel)qgemm_tile_kernel<F16, 4, 32, 1, 4, 2>;
        if (tile_p == 64 && mt == 4 && dtype == 0) return (QGemmKernel)qgemm_tile_kernel<F16, 4, 64, 1, 4, 2>;
        return nullptr;
    }
    if (tile_p == 32 && r == 1 && mt == 1) return dtype == 0 ? (QGemmKernel)qgemm_tile_kernel<F16, 4, 32, 1, 1> : (QGemmKernel)qgemm_tile_kernel<BF16, 4, 32, 1, 1>;
    if (tile_p == 32 && r == 2 && mt == 1) return dtype == 0 ? (QGemmKernel)qgemm_tile_kernel<F16, 4, 32, 2, 1> : (QGemmKernel)qgemm_tile_kernel<BF16, 4, 32, 2, 1>;
    if (tile_p == 32 && r == 4 && mt == 1) return dtype == 0 ? (QGemmKernel)qgemm_tile_kernel<F16, 4, 32, 4, 1> : (QGemmKernel)qgemm_tile_kernel<BF16, 4, 32, 4, 1>;
    if (tile_p == 32 && r == 1 && mt == 2) return dtype == 0 ? (QGemmKernel)qgemm_tile_kernel<F16, 4, 32, 1, 2> : (QGemmKernel)qgemm_tile_kernel<BF16, 4, 32, 1, 2>;
    if (tile_p == 32 && r == 2 && mt == 2) return dtype == 0 ? (QGemmKernel)qgemm_tile_kernel<F16, 4, 32, 2, 2> : (QGemmKernel)qgemm_tile_kernel<BF16, 4, 32, 2, 2>;
    if (tile_p == 32 && r == 1 && mt == 4) return dtype == 0 ? (QGemmKernel)qgemm_tile_kernel<F16, 4, 32, 1, 4> : (QGemmKernel)qgemm_tile_kernel<BF16, 4, 32, 1, 4>;
    if (tile_p == 32 && r == 2 && mt == 4) return dtype == 0 ? (QGemmKernel)qgemm_tile_kernel<F16, 4, 32, 2, 4> : (QGemmKernel)qgemm_tile_kernel<BF16, 4, 32, 2, 4>;
    if (tile_p == 64 && r == 1 && mt == 1) return dtype == 0 ? (QGemmKernel)qgemm_tile_kernel<F16, 4, 64, 1, 1> : (QGemmKernel)qgemm_tile_kernel<BF16, 4, 64, 1, 1>;
    if (tile_p == 64 && r == 2 && mt == 1) return dtype == 0 ? (QGemmKernel)qgemm_tile_kernel<F16, 4, 64, 2, 1> : (QGemmKernel)qgemm_tile_kernel<BF16, 4, 64, 2, 1>;
    if (tile_p == 64 && r == 4 && mt == 1) return dtype == 0 ? (QGemmKernel)qgemm_tile_kernel<F16, 4, 64, 4, 1> : (QGemmKernel)qgemm_tile_kernel<BF16, 4, 64, 4, 1>;
    if (tile_p == 64 && r == 1 && mt == 2) return dtype == 0 ? (QGemmKernel)qgemm_tile_kernel<F16, 4, 64, 1, 2> : (QGemmKernel)qgemm_tile_kernel<BF16, 4, 64, 1, 2>;
    if (tile_p == 64 && r == 2 && mt == 2) return dtype == 0 ? (QGemmKernel)qgemm_tile_kernel<F16, 4, 64, 2, 2> : (QGemmKernel)qgemm_tile_kernel<BF16, 4, 64, 2, 2>;
    if (tile_p == 64 && r == 1 && mt == 4) return dtype == 0 ? (QGemmKernel)qgemm_tile_kernel<F16, 4, 64, 1, 4> : (QGemmKernel)qgemm_tile_kernel<BF16, 4, 64, 1, 4>;
    if (tile_p == 64 && r == 2 && mt == 4) return dtype == 0 ? (QGemmKernel)qgemm_tile_kernel<F16, 4, 64, 2, 4> : (QGemmKernel)qgemm_tile_kernel<BF16, 4, 64, 2, 4>;
    return nullptr;
}
}  // namespace flute_amd
