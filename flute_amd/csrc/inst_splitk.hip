// Explicit instantiations of the split-K block kernel (qgemm_splitk.h) for num_bits = 4 and 2; rt = 8 / 4: 128- / 64-row tiles;
// kp = 2 / 4: K parts per workgroup (4: 64-column tiles of 64 rows - round 6).  Eight compute + four loader waves each.
#include "kernels.h"
#include "qgemm_splitk.h"
namespace flute_amd {
#define FLUTE_SK(B, TP, R, KP) \
    if (bits == B && tile_p == TP && rt == R && kp == KP) return dtype == 0 ? (SplitKKernel)qgemm_splitk_kernel<F16, TP, B, R, KP> : (SplitKKernel)qgemm_splitk_kernel<BF16, TP, B, R, KP>;
SplitKKernel splitk_kernel(int bits, int dtype, int tile_p, int rt, int kp) {
    FLUTE_SK(4, 32, 8, 2) FLUTE_SK(4, 64, 8, 2) FLUTE_SK(2, 32, 8, 2) FLUTE_SK(2, 64, 8, 2)
    FLUTE_SK(4, 32, 4, 2) FLUTE_SK(4, 64, 4, 2) FLUTE_SK(2, 32, 4, 2) FLUTE_SK(2, 64, 4, 2)
    FLUTE_SK(4, 32, 4, 4) FLUTE_SK(4, 64, 4, 4) FLUTE_SK(2, 32, 4, 4) FLUTE_SK(2, 64, 4, 4)
    return nullptr;
}
}  // namespace flute_amd
