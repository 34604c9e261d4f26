// Explicit instantiations of the one-shot decode kernel (qgemm_oneshot.h), num_bits = 4, F16: TileP x rows per
// pass x pieces per wave x fused Hadamard (+ the software-pipelined 4-bit single-row loop).  Built with
// -mllvm -amdgpu-kernarg-preload-count=14 (Makefile): the leading kernel arguments arrive in SGPRs.
#include "kernels.h"
#include "qgemm_oneshot.h"
namespace flute_amd {
#define FLUTE_ONE(TP, MB, D, H, O) (OneKernel)qgemv_oneshot_kernel<F16, 4, TP, MB, D, (MB == 4 ? 1 : 2), H, O>
#define FLUTE_ROW(TP, MB, D) \
    if (tile_p == TP && mb == MB && depth == D) return had ? FLUTE_ONE(TP, MB, D, true, 33) : FLUTE_ONE(TP, MB, D, false, 33);
#define FLUTE_ROW_PIPE(TP, D) \
    if (tile_p == TP && mb == 1 && depth == D && pipe) return had ? FLUTE_ONE(TP, 1, D, true, 49) : FLUTE_ONE(TP, 1, D, false, 49);
OneKernel oneshot_kernel_b4_f16(int tile_p, int mb, int depth, int had, int pipe) {
    (void)pipe;
    FLUTE_ROW_PIPE(32, 4) FLUTE_ROW_PIPE(32, 8) FLUTE_ROW_PIPE(64, 4) FLUTE_ROW_PIPE(64, 8)
    FLUTE_ROW(32, 1, 4) FLUTE_ROW(32, 1, 8) FLUTE_ROW(32, 2, 4) FLUTE_ROW(32, 2, 8) FLUTE_ROW(32, 4, 4) FLUTE_ROW(32, 4, 8)
    FLUTE_ROW(64, 1, 4) FLUTE_ROW(64, 1, 8) FLUTE_ROW(64, 2, 4) FLUTE_ROW(64, 2, 8) FLUTE_ROW(64, 4, 4) FLUTE_ROW(64, 4, 8)
    return nullptr;
}
}  // namespace flute_amd
