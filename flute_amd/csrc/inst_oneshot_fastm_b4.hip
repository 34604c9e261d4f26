// Explicit instantiations of the lean MFMA decode kernel (qgemm_fastm.h), num_bits = 4: dtype x TileP x (waves, macro-steps
// per wave) x group size x column groups per workgroup (1, 2, 3: round 6).  Built with -mllvm -amdgpu-kernarg-preload-count=14 (Makefile): the arguments arrive in SGPRs.
#include "kernels.h"
#include "qgemm_fastm.h"
namespace flute_amd {
#define FLUTE_FM(T, TP, W, NM, LG) \
    if (tile_p == TP && waves == W && nm == NM && lg == LG && ng == 1) return (FastMKernel)qgemm_fastm_kernel<T, TP, W, NM, LG, 1>; \
    if (tile_p == TP && waves == W && nm == NM && lg == LG && ng == 2) return (FastMKernel)qgemm_fastm_kernel<T, TP, W, NM, LG, 2>; \
    if (tile_p == TP && waves == W && nm == NM && lg == LG && ng == 3) return (FastMKernel)qgemm_fastm_kernel<T, TP, W, NM, LG, 3>;
// K = 4096: 8 waves x 512 k; K = 2048: 8 waves x 256 k (group size 256 would leave a wave one group: its scale word is not dword aligned)
#define FLUTE_FM_SHAPES(T, TP) \
    FLUTE_FM(T, TP, 8, 4, 6) FLUTE_FM(T, TP, 8, 4, 7) FLUTE_FM(T, TP, 8, 4, 8) FLUTE_FM(T, TP, 8, 2, 6) FLUTE_FM(T, TP, 8, 2, 7)
FastMKernel fastm_kernel_b4(int dtype, int tile_p, int waves, int nm, int lg, int ng) {
    if (dtype == 0) { FLUTE_FM_SHAPES(F16, 32) FLUTE_FM_SHAPES(F16, 64) }
    else { FLUTE_FM_SHAPES(BF16, 32) FLUTE_FM_SHAPES(BF16, 64) }
    return nullptr;
}
}  // namespace flute_amd
