// Explicit instantiations of the split-K block kernel (qgemm_splitk.h) for num_bits = 4 and 2.
#include "kernels.h"
#include "qgemm_splitk.h"
namespace flute_amd {
SplitKKernel splitk_kernel(int bits, int dtype, int tile_p) {
    if (bits == 4 && tile_p == 32) return dtype == 0 ? (SplitKKernel)qgemm_splitk_kernel<F16, 32, 4> : (SplitKKernel)qgemm_splitk_kernel<BF16, 32, 4>;
    if (bits == 4 && tile_p == 64) return dtype == 0 ? (SplitKKernel)qgemm_splitk_kernel<F16, 64, 4> : (SplitKKernel)qgemm_splitk_kernel<BF16, 64, 4>;
    if (bits == 2 && tile_p == 32) return dtype == 0 ? (SplitKKernel)qgemm_splitk_kernel<F16, 32, 2> : (SplitKKernel)qgemm_splitk_kernel<BF16, 32, 2>;
    if (bits == 2 && tile_p == 64) return dtype == 0 ? (SplitKKernel)qgemm_splitk_kernel<F16, 64, 2> : (SplitKKernel)qgemm_splitk_kernel<BF16, 64, 2>;
    return nullptr;
}
}  // namespace flute_amd
