// Explicit instantiations of the mid-M kernel (qgemm_mid.h): bits x dtype x TileP x row tiles.
#include "kernels.h"
#include "qgemm_mid.h"
namespace flute_amd {
#define FLUTE_MID(B, TP, RT) (dtype == 0 ? (BlockKernel)qgemm_mid_kernel<F16, B, TP, RT> : (BlockKernel)qgemm_mid_kernel<BF16, B, TP, RT>)
BlockKernel mid_kernel_b4(int dtype, int tile_p, int rt) {
    if (tile_p == 32 && rt == 4) return FLUTE_MID(4, 32, 4);
    if (tile_p == 32 && rt == 8) return FLUTE_MID(4, 32, 8);
    if (tile_p == 64 && rt == 4) return FLUTE_MID(4, 64, 4);
    if (tile_p == 64 && rt == 8) return FLUTE_MID(4, 64, 8);
    return nullptr;
}
BlockKernel mid_kernel_b2(int dtype, int tile_p, int rt) {
    if (tile_p == 32 && rt == 4) return FLUTE_MID(2, 32, 4);
    if (tile_p == 32 && rt == 8) return FLUTE_MID(2, 32, 8);
    if (tile_p == 64 && rt == 4) return FLUTE_MID(2, 64, 4);
    if (tile_p == 64 && rt == 8) return FLUTE_MID(2, 64, 8);
    return nullptr;
}
}  // namespace flute_amd
