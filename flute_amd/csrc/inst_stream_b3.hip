// Explicit instantiations of the streaming decode kernel (qgemm_stream.h) for num_bits = 3:
// dtype x TileP x rows per pass x ring depth (+ the one-shot variant).  One translation unit per bit
// width (`make -j`).
#include "kernels.h"
#include "qgemm_stream.h"
namespace flute_amd {
StreamKernel stream_kernel_b3(int dtype, int tile_p, int mb, int depth, int one_shot) {
    if (tile_p == 32 && mb == 1 && !one_shot && depth == 2) return dtype == 0 ? (StreamKernel)qgemv_stream_kernel<F16, 3, 32, 1, 2> : (StreamKernel)qgemv_stream_kernel<BF16, 3, 32, 1, 2>;
    if (tile_p == 32 && mb == 2 && !one_shot && depth == 2) return dtype == 0 ? (StreamKernel)qgemv_stream_kernel<F16, 3, 32, 2, 2> : (StreamKernel)qgemv_stream_kernel<BF16, 3, 32, 2, 2>;
    // four rows per pass (the 3-bit MFMA plans are 256 columns wide per wave and slow at M = 3, 4)
    if (tile_p == 32 && mb == 4 && !one_shot && depth == 2) return dtype == 0 ? (StreamKernel)qgemv_stream_kernel<F16, 3, 32, 4, 2> : (StreamKernel)qgemv_stream_kernel<BF16, 3, 32, 4, 2>;
    return nullptr;
}
}  // namespace flute_amd
