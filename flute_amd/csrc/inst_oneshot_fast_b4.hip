// Explicit instantiations of the lean decode kernel (qgemm_fast.h), num_bits = 4: dtype x TileP x (waves per workgroup,
// waves per unit row, pieces per wave) x rows per pass - the shapes api.hip's plan_fast hands out.  Built with
// -mllvm -amdgpu-kernarg-preload-count=14 (Makefile): the kernel's arguments arrive in SGPRs.
#include "kernels.h"
#include "qgemm_fast.h"
namespace flute_amd {
#define FLUTE_FAST(T, TP, W, KW, D, MB) \
    if (tile_p == TP && waves == W && kw == KW && depth == D && mb == MB) return (FastKernel)qgemv_fast_kernel<T, TP, W, KW, D, MB, 0>;
// K = 2048: (4, 1, 4); 3584: (4, 1, 7); 4096: (4, 1, 8) / (8, 2, 4); 8192: (8, 2, 8).  Four rows of K = 8192 would need 64 KB of
// activations beside the table image: not instantiated.
#define FLUTE_FAST_SHAPES(T, TP) \
    FLUTE_FAST(T, TP, 4, 1, 4, 1) FLUTE_FAST(T, TP, 4, 1, 8, 1) FLUTE_FAST(T, TP, 8, 2, 4, 1) FLUTE_FAST(T, TP, 8, 2, 8, 1) FLUTE_FAST(T, TP, 4, 1, 7, 1) \
    FLUTE_FAST(T, TP, 4, 1, 4, 2) FLUTE_FAST(T, TP, 4, 1, 8, 2) FLUTE_FAST(T, TP, 8, 2, 4, 2) FLUTE_FAST(T, TP, 8, 2, 8, 2) FLUTE_FAST(T, TP, 4, 1, 7, 2) \
    FLUTE_FAST(T, TP, 4, 1, 4, 4) FLUTE_FAST(T, TP, 4, 1, 8, 4) FLUTE_FAST(T, TP, 8, 2, 4, 4) FLUTE_FAST(T, TP, 4, 1, 7, 4)
FastKernel fast_kernel_b4(int dtype, int tile_p, int waves, int kw, int depth, int mb) {
    if (dtype == 0) { FLUTE_FAST_SHAPES(F16, 32) FLUTE_FAST_SHAPES(F16, 64) }
    else { FLUTE_FAST_SHAPES(BF16, 32) FLUTE_FAST_SHAPES(BF16, 64) }
    return nullptr;
}
}  // namespace flute_amd
