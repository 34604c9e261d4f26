// Explicit instantiations of the block-tiled prefill kernel (qgemm_block.h) for num_bits = 4.
#include "kernels.h"
#include "qgemm_block.h"
#include "qgemm_block2.h"
namespace flute_amd {
// cfg 0: 256 x 256 block (TM = 8 row tiles per wave, 2 x 4 waves); cfg 1: 128 x 256 (TM = 4)
BlockKernel block_kernel_b4(int dtype, int tile_p, int cfg) {
    if (tile_p == 32 && cfg == 0) return dtype == 0 ? (BlockKernel)qgemm_block_kernel<F16, 4, 32, 8, 2, 4> : (BlockKernel)qgemm_block_kernel<BF16, 4, 32, 8, 2, 4>;
    if (tile_p == 64 && cfg == 0) return dtype == 0 ? (BlockKernel)qgemm_block_kernel<F16, 4, 64, 8, 2, 4> : (BlockKernel)qgemm_block_kernel<BF16, 4, 64, 8, 2, 4>;
    if (tile_p == 32 && cfg == 1) return dtype == 0 ? (BlockKernel)qgemm_block_kernel<F16, 4, 32, 4, 2, 4> : (BlockKernel)qgemm_block_kernel<BF16, 4, 32, 4, 2, 4>;
    if (tile_p == 64 && cfg == 1) return dtype == 0 ? (BlockKernel)qgemm_block_kernel<F16, 4, 64, 4, 2, 4> : (BlockKernel)qgemm_block_kernel<BF16, 4, 64, 4, 2, 4>;
    // cfg 2 / 3: the same two blocks on the software-pipelined schedule
    if (tile_p == 32 && cfg == 2) return dtype == 0 ? (BlockKernel)qgemm_block_kernel<F16, 4, 32, 8, 2, 4, true> : (BlockKernel)qgemm_block_kernel<BF16, 4, 32, 8, 2, 4, true>;
    if (tile_p == 64 && cfg == 2) return dtype == 0 ? (BlockKernel)qgemm_block_kernel<F16, 4, 64, 8, 2, 4, true> : (BlockKernel)qgemm_block_kernel<BF16, 4, 64, 8, 2, 4, true>;
    if (tile_p == 32 && cfg == 3) return dtype == 0 ? (BlockKernel)qgemm_block_kernel<F16, 4, 32, 4, 2, 4, true> : (BlockKernel)qgemm_block_kernel<BF16, 4, 32, 4, 2, 4, true>;
    if (tile_p == 64 && cfg == 3) return dtype == 0 ? (BlockKernel)qgemm_block_kernel<F16, 4, 64, 4, 2, 4, true> : (BlockKernel)qgemm_block_kernel<BF16, 4, 64, 4, 2, 4, true>;
    // cfg 4: 256 x 256 block split 1 x 8 over the waves (qgemm_block2.h): every weight dequantised once per workgroup
    if (tile_p == 32 && cfg == 4) return dtype == 0 ? (BlockKernel)qgemm_block2_kernel<F16, 32> : (BlockKernel)qgemm_block2_kernel<BF16, 32>;
    if (tile_p == 64 && cfg == 4) return dtype == 0 ? (BlockKernel)qgemm_block2_kernel<F16, 64> : (BlockKernel)qgemm_block2_kernel<BF16, 64>;
    // cfg 5: the 1 x 8 split on 128-row blocks
    if (tile_p == 32 && cfg == 5) return dtype == 0 ? (BlockKernel)qgemm_block2_kernel<F16, 32, 8> : (BlockKernel)qgemm_block2_kernel<BF16, 32, 8>;
    if (tile_p == 64 && cfg == 5) return dtype == 0 ? (BlockKernel)qgemm_block2_kernel<F16, 64, 8> : (BlockKernel)qgemm_block2_kernel<BF16, 64, 8>;
    return nullptr;
}
}  // namespace flute_amd
