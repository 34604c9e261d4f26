// Explicit instantiations of the persistent MFMA decode kernel (qgemm_persistm.h), num_bits = 2, f16: TileP x group size (64 / 128) x
// column groups per set (1, 2, 3) x activation requests per macro-step (1, 2, 4); activations resident in LDS where they fit (K * xr <= 8192).
// Built with -mllvm -amdgpu-kernarg-preload-count=14 (Makefile): the arguments arrive in SGPRs.
#include "kernels.h"
#include "qgemm_persistm.h"
namespace flute_amd {
#define FLUTE_PM1(TP, LG, NG, XR) if (tile_p == TP && lg == LG && ng == NG && xr == XR && waves == 8 && !xres) return (PersistMKernel)qgemm_persistm_kernel<F16, TP, LG, NG, XR, 8, false, 2>;
#define FLUTE_PMR(TP, LG, NG, XR) if (tile_p == TP && lg == LG && ng == NG && xr == XR && waves == 8 && xres) return (PersistMKernel)qgemm_persistm_kernel<F16, TP, LG, NG, XR, 8, true, 2>;
#define FLUTE_PM(TP, LG) \
    FLUTE_PM1(TP, LG, 1, 1) FLUTE_PM1(TP, LG, 1, 2) FLUTE_PM1(TP, LG, 1, 4) FLUTE_PM1(TP, LG, 2, 1) FLUTE_PM1(TP, LG, 2, 2) FLUTE_PM1(TP, LG, 2, 4) \
    FLUTE_PM1(TP, LG, 3, 1) FLUTE_PM1(TP, LG, 3, 2) FLUTE_PM1(TP, LG, 3, 4) \
    FLUTE_PMR(TP, LG, 1, 1) FLUTE_PMR(TP, LG, 2, 1) FLUTE_PMR(TP, LG, 1, 2) FLUTE_PMR(TP, LG, 2, 2) FLUTE_PMR(TP, LG, 3, 2)
PersistMKernel persistm_kernel_b2_f16(int tile_p, int lg, int ng, int xr, int waves, int xres) {
    FLUTE_PM(32, 6) FLUTE_PM(32, 7) FLUTE_PM(64, 6) FLUTE_PM(64, 7)
    return nullptr;
}
}  // namespace flute_amd
