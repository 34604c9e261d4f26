// Explicit instantiations of the split-K block kernel (qgemm_splitk.h) for num_bits = 4 and 2; ldw = 4: with loader waves;
// rt = 8 / 4: 128- / 64-row tiles.
#include "kernels.h"
#include "qgemm_splitk.h"
namespace flute_amd {
#define FLUTE_SK(B, TP, L, R) \
    if (bits == B && tile_p == TP && ldw == L && rt == R) return dtype == 0 ? (SplitKKernel)qgemm_splitk_kernel<F16, TP, B, L, R> : (SplitKKernel)qgemm_splitk_kernel<BF16, TP, B, L, R>;
SplitKKernel splitk_kernel(int bits, int dtype, int tile_p, int ldw, int rt) {
    FLUTE_SK(4, 32, 0, 8) FLUTE_SK(4, 64, 0, 8) FLUTE_SK(2, 32, 0, 8) FLUTE_SK(2, 64, 0, 8)
    FLUTE_SK(4, 32, 4, 8) FLUTE_SK(4, 64, 4, 8) FLUTE_SK(2, 32, 4, 8) FLUTE_SK(2, 64, 4, 8)
    FLUTE_SK(4, 32, 4, 4) FLUTE_SK(4, 64, 4, 4) FLUTE_SK(2, 32, 4, 4) FLUTE_SK(2, 64, 4, 4)
    FLUTE_SK(4, 32, 0, 4) FLUTE_SK(4, 64, 0, 4) FLUTE_SK(2, 32, 0, 4) FLUTE_SK(2, 64, 0, 4)
    return nullptr;
}
}  // namespace flute_amd
