// Explicit instantiations of the block-tiled prefill kernel (qgemm_block2.h) for num_bits = 4.
#include "kernels.h"
#include "qgemm_block2.h"
namespace flute_amd {
// cfg 4: 256 x 256 block split 1 x 8 over the waves (every weight dequantised once per workgroup); cfg 5: the same on
// 128-row blocks.  (cfg 0..3 were the 2 x 4 split of round 2, removed.)
BlockKernel block_kernel_b4(int dtype, int tile_p, int cfg) {
    if (tile_p == 32 && cfg == 4) return dtype == 0 ? (BlockKernel)qgemm_block2_kernel<F16, 32> : (BlockKernel)qgemm_block2_kernel<BF16, 32>;
    if (tile_p == 64 && cfg == 4) return dtype == 0 ? (BlockKernel)qgemm_block2_kernel<F16, 64> : (BlockKernel)qgemm_block2_kernel<BF16, 64>;
    if (tile_p == 32 && cfg == 5) return dtype == 0 ? (BlockKernel)qgemm_block2_kernel<F16, 32, 8> : (BlockKernel)qgemm_block2_kernel<BF16, 32, 8>;
    if (tile_p == 64 && cfg == 5) return dtype == 0 ? (BlockKernel)qgemm_block2_kernel<F16, 64, 8> : (BlockKernel)qgemm_block2_kernel<BF16, 64, 8>;
    return nullptr;
}
}  // namespace flute_amd
