// Explicit instantiations of the split-K block kernel (qgemm_splitk.h) for num_bits = 4 and 2; ldw = 4: with loader waves.
#include "kernels.h"
#include "qgemm_splitk.h"
namespace flute_amd {
#define FLUTE_SK(B, TP, L) \
    if (bits == B && tile_p == TP && ldw == L) return dtype == 0 ? (SplitKKernel)qgemm_splitk_kernel<F16, TP, B, L> : (SplitKKernel)qgemm_splitk_kernel<BF16, TP, B, L>;
SplitKKernel splitk_kernel(int bits, int dtype, int tile_p, int ldw) {
    FLUTE_SK(4, 32, 0) FLUTE_SK(4, 64, 0) FLUTE_SK(2, 32, 0) FLUTE_SK(2, 64, 0)
    FLUTE_SK(4, 32, 4) FLUTE_SK(4, 64, 4) FLUTE_SK(2, 32, 4) FLUTE_SK(2, 64, 4)
    return nullptr;
}
}  // namespace flute_amd
