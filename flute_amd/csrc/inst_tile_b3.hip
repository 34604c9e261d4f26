// Explicit instantiations of the LDS-DMA staged MFMA kernel for num_bits = 3 (R lanes share a
// unit, MT 16-row tiles per wave; (16/3... J/R)*MT <= 16 accumulator tiles).
#include "kernels.h"
#include "qgemm_tile.h"
namespace flute_amd {
QGemmKernel tile_kernel_b3(int dtype, int tile_p, int r, int mt) {
    if (tile_p == 32 && r == 1 && mt == 1) return dtype == 0 ? (QGemmKernel)qgemm_tile_kernel<F16, 3, 32, 1, 1> : (QGemmKernel)qgemm_tile_kernel<BF16, 3, 32, 1, 1>;
    return nullptr;
}
}  // namespace flute_amd
