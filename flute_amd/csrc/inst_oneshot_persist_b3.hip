// Explicit instantiations of the persistent one-shot decode kernel (qgemm_persist.h), num_bits = 3: dtype x TileP x rows per pass
// (1, 2; four rows were measured - 1.7x the one-row time, no faster than the MFMA kernel - and are not built) x (pieces per segment, register sets) x fused Hadamard.  Two sets of 4 (3 bits: 2) pieces measured best
// (profiles/r03/persist_lab.txt: three / four sets and 8-piece segments were 3 - 10 % slower); 2-piece segments also
// serve K that is not a multiple of 2048.  Built like the one-shot kernels (-mllvm
// -amdgpu-kernarg-preload-count=14).
#include "kernels.h"
#include "qgemm_persist.h"
namespace flute_amd {
#define FLUTE_P(T, TP, MB, D, NS, H) (PersistKernel)qgemv_persist_kernel<T, 3, TP, MB, D, NS, H>
#define FLUTE_ROW1(TP, MB, D, NS) \
    if (tile_p == TP && mb == MB && depth == D && nsets == NS) { \
        if (dtype == 0) return had ? FLUTE_P(F16, TP, MB, D, NS, true) : FLUTE_P(F16, TP, MB, D, NS, false); \
        return had ? FLUTE_P(BF16, TP, MB, D, NS, true) : FLUTE_P(BF16, TP, MB, D, NS, false); \
    }
#define FLUTE_ROW(TP, D, NS) FLUTE_ROW1(TP, 1, D, NS) FLUTE_ROW1(TP, 2, D, NS)
PersistKernel persist_kernel_b3(int dtype, int tile_p, int mb, int depth, int nsets, int had) {
    FLUTE_ROW(32, 2, 2)                                         // 3-bit packing exists for TileP = 32 only
    return nullptr;
}
}  // namespace flute_amd
