// Explicit instantiations of the one-shot decode kernel (qgemm_oneshot.h), num_bits = 4, BF16: TileP x rows per
// pass x pieces per wave x fused Hadamard x (plain | software-pipelined) piece loop.  OPT 33 / 49 = nt weight loads +
// interleaved prologue (+ pipelined loop).  Built with -mllvm -amdgpu-kernarg-preload-count=14 (Makefile): the leading
// kernel arguments arrive in SGPRs.
#include "kernels.h"
#include "qgemm_oneshot.h"
namespace flute_amd {
#define FLUTE_ONE(TP, MB, D, H, O) (OneKernel)qgemv_oneshot_kernel<BF16, 4, TP, MB, D, (MB == 4 ? 1 : 2), H, O>
// the pipelined loop is instantiated for one row only (two and four rows spill the in-flight registers at 128 VGPRs:
// tools/audit_asm_loads.py flags the scratch stores)
#define FLUTE_ROW(TP, MB, D) \
    if (tile_p == TP && mb == MB && depth == D) { \
        if constexpr (MB == 1) { if (pipe) return had ? FLUTE_ONE(TP, MB, D, true, 49) : FLUTE_ONE(TP, MB, D, false, 49); } \
        return had ? FLUTE_ONE(TP, MB, D, true, 33) : FLUTE_ONE(TP, MB, D, false, 33); \
    }
OneKernel oneshot_kernel_b4_bf16(int tile_p, int mb, int depth, int had, int pipe) {
    FLUTE_ROW(32, 1, 4) FLUTE_ROW(32, 1, 8) FLUTE_ROW(32, 2, 4) FLUTE_ROW(32, 2, 8) FLUTE_ROW(32, 4, 4) FLUTE_ROW(32, 4, 8)
    FLUTE_ROW(64, 1, 4) FLUTE_ROW(64, 1, 8) FLUTE_ROW(64, 2, 4) FLUTE_ROW(64, 2, 8) FLUTE_ROW(64, 4, 4) FLUTE_ROW(64, 4, 8)
    return nullptr;
}
}  // namespace flute_amd
