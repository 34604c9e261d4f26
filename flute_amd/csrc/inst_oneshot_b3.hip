// Explicit instantiations of the one-shot decode kernel (qgemm_oneshot.h), num_bits = 3 (TileP 32 only,
// utils.py:137-139): dtype x rows per pass x pieces per wave x fused Hadamard x (plain | software-pipelined) loop.
// Built with -mllvm -amdgpu-kernarg-preload-count=14 (Makefile).
#include "kernels.h"
#include "qgemm_oneshot.h"
namespace flute_amd {
#define FLUTE_ONE(T, MB, D, H, O) (OneKernel)qgemv_oneshot_kernel<T, 3, 32, MB, D, 2, H, O>
#define FLUTE_ROW(MB, D) \
    if (mb == MB && depth == D) { \
        if (dtype == 0) { \
            if constexpr (MB == 1) { if (pipe) return had ? FLUTE_ONE(F16, MB, D, true, 49) : FLUTE_ONE(F16, MB, D, false, 49); } \
            return had ? FLUTE_ONE(F16, MB, D, true, 33) : FLUTE_ONE(F16, MB, D, false, 33); \
        } \
        if constexpr (MB == 1) { if (pipe) return had ? FLUTE_ONE(BF16, MB, D, true, 49) : FLUTE_ONE(BF16, MB, D, false, 49); } \
        return had ? FLUTE_ONE(BF16, MB, D, true, 33) : FLUTE_ONE(BF16, MB, D, false, 33); \
    }
OneKernel oneshot_kernel_b3(int dtype, int tile_p, int mb, int depth, int had, int pipe) {
    if (tile_p != 32) return nullptr;
    FLUTE_ROW(1, 2) FLUTE_ROW(1, 4) FLUTE_ROW(2, 2) FLUTE_ROW(2, 4)
    return nullptr;
}
}  // namespace flute_amd
