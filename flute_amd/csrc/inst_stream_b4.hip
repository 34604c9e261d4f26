// Explicit instantiations of the streaming decode kernel (qgemm_stream.h) for num_bits = 4:
// dtype x TileP x rows per pass x ring depth (+ the one-shot variant).  One translation unit per bit
// width (`make -j`).
#include "kernels.h"
#include "qgemm_stream.h"
namespace flute_amd {
StreamKernel stream_kernel_b4(int dtype, int tile_p, int mb, int depth, int one_shot) {
    if (tile_p == 32 && mb == 1 && !one_shot && depth == 2) return dtype == 0 ? (StreamKernel)qgemv_stream_kernel<F16, 4, 32, 1, 2> : (StreamKernel)qgemv_stream_kernel<BF16, 4, 32, 1, 2>;
    if (tile_p == 32 && mb == 1 && !one_shot && depth == 4) return dtype == 0 ? (StreamKernel)qgemv_stream_kernel<F16, 4, 32, 1, 4> : (StreamKernel)qgemv_stream_kernel<BF16, 4, 32, 1, 4>;
    if (tile_p == 32 && mb == 2 && !one_shot && depth == 2) return dtype == 0 ? (StreamKernel)qgemv_stream_kernel<F16, 4, 32, 2, 2> : (StreamKernel)qgemv_stream_kernel<BF16, 4, 32, 2, 2>;
    if (tile_p == 32 && mb == 2 && !one_shot && depth == 4) return dtype == 0 ? (StreamKernel)qgemv_stream_kernel<F16, 4, 32, 2, 4> : (StreamKernel)qgemv_stream_kernel<BF16, 4, 32, 2, 4>;
    if (tile_p == 32 && mb == 4 && !one_shot && depth == 2) return dtype == 0 ? (StreamKernel)qgemv_stream_kernel<F16, 4, 32, 4, 2> : (StreamKernel)qgemv_stream_kernel<BF16, 4, 32, 4, 2>;
    if (tile_p == 32 && mb == 4 && !one_shot && depth == 4) return dtype == 0 ? (StreamKernel)qgemv_stream_kernel<F16, 4, 32, 4, 4> : (StreamKernel)qgemv_stream_kernel<BF16, 4, 32, 4, 4>;
    if (tile_p == 64 && mb == 1 && !one_shot && depth == 2) return dtype == 0 ? (StreamKernel)qgemv_stream_kernel<F16, 4, 64, 1, 2> : (StreamKernel)qgemv_stream_kernel<BF16, 4, 64, 1, 2>;
    if (tile_p == 64 && mb == 1 && !one_shot && depth == 4) return dtype == 0 ? (StreamKernel)qgemv_stream_kernel<F16, 4, 64, 1, 4> : (StreamKernel)qgemv_stream_kernel<BF16, 4, 64, 1, 4>;
    if (tile_p == 64 && mb == 2 && !one_shot && depth == 2) return dtype == 0 ? (StreamKernel)qgemv_stream_kernel<F16, 4, 64, 2, 2> : (StreamKernel)qgemv_stream_kernel<BF16, 4, 64, 2, 2>;
    if (tile_p == 64 && mb == 2 && !one_shot && depth == 4) return dtype == 0 ? (StreamKernel)qgemv_stream_kernel<F16, 4, 64, 2, 4> : (StreamKernel)qgemv_stream_kernel<BF16, 4, 64, 2, 4>;
    if (tile_p == 64 && mb == 4 && !one_shot && depth == 2) return dtype == 0 ? (StreamKernel)qgemv_stream_kernel<F16, 4, 64, 4, 2> : (StreamKernel)qgemv_stream_kernel<BF16, 4, 64, 4, 2>;
    if (tile_p == 64 && mb == 4 && !one_shot && depth == 4) return dtype == 0 ? (StreamKernel)qgemv_stream_kernel<F16, 4, 64, 4, 4> : (StreamKernel)qgemv_stream_kernel<BF16, 4, 64, 4, 4>;
    return nullptr;
}
}  // namespace flute_amd
