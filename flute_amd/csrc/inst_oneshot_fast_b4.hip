// Explicit instantiations of the lean one-row decode kernel (qgemm_fast.h), num_bits = 4: dtype x TileP x (waves per
// workgroup, waves per unit row, pieces per wave) - the shapes api.hip's plan_fast hands out.  Built with
// -mllvm -amdgpu-kernarg-preload-count=14 (Makefile): the kernel's arguments arrive in SGPRs.
#include "kernels.h"
#include "qgemm_fast.h"
namespace flute_amd {
#define FLUTE_FAST(T, TP, W, KW, D) \
    if (tile_p == TP && waves == W && kw == KW && depth == D) return (FastKernel)qgemv_fast_kernel<T, TP, W, KW, D, 0, 0>;
#define FLUTE_FAST_SHAPES(T, TP) FLUTE_FAST(T, TP, 4, 1, 4) FLUTE_FAST(T, TP, 4, 1, 8) FLUTE_FAST(T, TP, 8, 2, 4) FLUTE_FAST(T, TP, 8, 2, 8)
FastKernel fast_kernel_b4(int dtype, int tile_p, int waves, int kw, int depth) {
    if (dtype == 0) { FLUTE_FAST_SHAPES(F16, 32) FLUTE_FAST_SHAPES(F16, 64) }
    else { FLUTE_FAST_SHAPES(BF16, 32) FLUTE_FAST_SHAPES(BF16, 64) }
    return nullptr;
}
}  // namespace flute_amd
