// Explicit instantiations of the skinny MFMA kernel (qgemm_skinny.h), num_bits = 4, M <= 16: dtype x TileP x k-steps per
// wave.  (Measured and not kept, profiles/r03/skinny_lab.jsonl: 2-bit layers - 128 columns per slab, no gain; two row
// tiles, M = 17..32 - as fast as the per-wave kernel; 16-wave workgroups - as 8.)  Built like the one-shot kernels
// (-mllvm -amdgpu-kernarg-preload-count=14).
#include "kernels.h"
#include "qgemm_skinny.h"
namespace flute_amd {
#define FLUTE_K(T, TP, D) (SkinnyKernel)qgemm_skinny_kernel<T, 4, TP, 1, D, 8>
#define FLUTE_ROW(TP, D) \
    if (tile_p == TP && depth == D) return dtype == 0 ? FLUTE_K(F16, TP, D) : FLUTE_K(BF16, TP, D);
SkinnyKernel skinny_kernel_b4(int dtype, int tile_p, int depth) {
    FLUTE_ROW(32, 4) FLUTE_ROW(32, 8) FLUTE_ROW(32, 16) FLUTE_ROW(64, 4) FLUTE_ROW(64, 8) FLUTE_ROW(64, 16)
    return nullptr;
}
}  // namespace flute_amd
