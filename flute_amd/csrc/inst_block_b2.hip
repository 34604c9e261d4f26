// Explicit instantiations of the block-tiled prefill kernel (qgemm_block2.h) for num_bits = 2.
#include "kernels.h"
#include "qgemm_block2.h"
namespace flute_amd {
// cfg 4: 256 x 256 blocks, cfg 5: 128 x 256 (the 1 x 8 wave split; the 2 x 4 split of qgemm_block.h is 4-bit only)
BlockKernel block_kernel_b2(int dtype, int tile_p, int cfg) {
    if (tile_p == 32 && cfg == 4) return dtype == 0 ? (BlockKernel)qgemm_block2_kernel<F16, 32, 16, 2> : (BlockKernel)qgemm_block2_kernel<BF16, 32, 16, 2>;
    if (tile_p == 64 && cfg == 4) return dtype == 0 ? (BlockKernel)qgemm_block2_kernel<F16, 64, 16, 2> : (BlockKernel)qgemm_block2_kernel<BF16, 64, 16, 2>;
    if (tile_p == 32 && cfg == 5) return dtype == 0 ? (BlockKernel)qgemm_block2_kernel<F16, 32, 8, 2> : (BlockKernel)qgemm_block2_kernel<BF16, 32, 8, 2>;
    if (tile_p == 64 && cfg == 5) return dtype == 0 ? (BlockKernel)qgemm_block2_kernel<F16, 64, 8, 2> : (BlockKernel)qgemm_block2_kernel<BF16, 64, 8, 2>;
    return nullptr;
}
}  // namespace flute_amd
