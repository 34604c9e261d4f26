// Explicit instantiations of the block-tiled prefill kernel for num_bits = 3 (qgemm_block3.h; TileP = 32 only,
// like every 3-bit template).
#include "kernels.h"
#include "qgemm_block3.h"
namespace flute_amd {
// cfg 5: 128 x 256 blocks; cfg 4: 256 x 256 blocks (the second / third plane pieces of waves 6, 7 in LDS)
BlockKernel block_kernel_b3(int dtype, int tile_p, int cfg) {
    if (tile_p == 32 && cfg == 4) return dtype == 0 ? (BlockKernel)qgemm_block3_kernel<F16, 16> : (BlockKernel)qgemm_block3_kernel<BF16, 16>;
    if (tile_p == 32 && cfg == 5) return dtype == 0 ? (BlockKernel)qgemm_block3_kernel<F16, 8> : (BlockKernel)qgemm_block3_kernel<BF16, 8>;
    // cfg 8 + RT: skinny blocks of RT = 1, 2, 4 row tiles for small batches (launched with a grid K split)
    if (tile_p == 32 && cfg == 9) return dtype == 0 ? (BlockKernel)qgemm_block3_kernel<F16, 1> : (BlockKernel)qgemm_block3_kernel<BF16, 1>;
    if (tile_p == 32 && cfg == 10) return dtype == 0 ? (BlockKernel)qgemm_block3_kernel<F16, 2> : (BlockKernel)qgemm_block3_kernel<BF16, 2>;
    if (tile_p == 32 && cfg == 12) return dtype == 0 ? (BlockKernel)qgemm_block3_kernel<F16, 4> : (BlockKernel)qgemm_block3_kernel<BF16, 4>;
    return nullptr;
}
}  // namespace flute_amd
