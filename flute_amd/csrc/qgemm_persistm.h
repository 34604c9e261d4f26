// Persistent MFMA decode kernel (round 6): 3 <= M <= 16 rows of a 4-bit layer too large or too deep for one round of the lean MFMA decode
// kernel (qgemm_fastm.h: K <= 4096, <= 3 x 256 column groups) - the 70B-class layers (8192 x 28672, 28672 x 8192, 8192^2) and the
// 8B-class MLP (4096 x 14336, 14336 x 3584) at the batch sizes between the dot-product decode kernels and the MFMA tiles.
//
// Until this kernel those launches ran on the per-wave MFMA kernel (qgemm_tile.h) or the skinny kernel (qgemm_skinny.h): 34 - 40 us
// on 8192 x 28672 / 28672 x 8192 at M = 4 .. 16 where M = 1 streams the same weights in 22.5 us (5.5 TB/s) - every MFMA operand
// travelled through its own request (an activation fragment per 256 B of weights: 4 x the weight bytes through the CU's addresser).
// Here the launch is ONE round of workgroups that STREAM: a workgroup owns column-group sets (NG groups of 4 unit rows = 16 columns
// each) s = blockIdx, blockIdx + grid, ... for the whole of K, its eight waves take the 128-k macro-steps w, w + 8, ... of every set
// (interleaved: the workgroup reads 2 KB contiguous per unit row and round), and each wave runs ONE flat software pipeline over all
// its (set, macro-step) pairs - the requests of a set's first macro-steps are in flight while the previous set is still multiplied:
//   * weights: one 16-B request per lane, group and macro-step (4 unit rows x 256 contiguous bytes: whole lines, non-temporal),
//     SIX macro-steps ahead, into a register ring; lane mapping, DPP quad broadcast and v_perm lookup addresses as qgemm_fastm.h;
//   * activations: the rows x 256 B of a macro-step by XR = 1, 2, 4 LDS-DMA requests (rows 4 r .. 4 r + 3 x 256 contiguous bytes: whole
//     lines; XR = ceil(M / 4) rounded up - the MFMA rows past 4 XR alias the requested ones, their products are never stored; rows
//     >= M lie past the descriptor: zeros without a trip to L2) into a wave-private ring of XR-KB slots - no VGPRs, no ds_write, no
//     barrier.  The ring is SIX slots deep, as the weight ring, where the LDS has room (XR <= 2: 48 / 96 KB), three otherwise.  The DMA
//     writes lane-linearly, so the swizzle is applied to what a lane ASKS for: lane l of request r fetches chunk
//     (l % 16) ^ 4 (l / 16) ^ g(r), g = (0, 3, 2, 1), of row 4 r + l / 16, and the fragment read of MFMA row m = 4 r + mm, k-chunk
//     4 s + kg finds it at position 16 mm + 4 (s ^ mm) + (kg ^ g(r)) of request r's KB: the 16 lanes of every ds_read_b128 lane group
//     hit 16 different 16-B slots (tests/test_host.py);
//   * group scales: the 2 (g = 64) or 1 (g = 128) groups of a macro-step for the 16 NG columns by ONE 4-B LDS-DMA request per
//     macro-step (lane = column: [group][unit][column of the unit]), read back as one ds_read_b128 per group: the four columns of
//     the lane's output unit;
//   * every macro-step issues the same NG + 1 + XR requests (positions past the wave's last pair go through zero-byte descriptors), so
//     the counted vmcnt waits are compile-time constants; the cursors of the three request streams are countdowns and additive
//     offsets in SGPRs (the first version's per-request multiplications and compares were ~170 scalar instructions per macro-step,
//     more than its VALU work: 34.4 -> 32.7 us on 28672 x 8192);
//   * a set's end: the waves' 16 x 16 partial tiles meet in LDS (two barriers per set), every wave sums and stores 32 outputs per
//     group in a fixed order.
// What bounds it (ablation builds, profiles/r06/call29_persistm_ablation.log, 28672 x 8192 at M = 4, first version 34.4 us): NOT HBM - with
// the weights behind a zero-byte descriptor 31.5 us; without lookups / MFMAs 28.5; without activation / scale requests 28.0; the loop and
// the weight requests alone 11.6.  A wave's in-order stream (2 waves per SIMD: the activation rings fill the LDS) is the bound: ~1 us per
// macro-step of two groups.  Hence the short activation requests (XR), the deep rings and the SGPR cursors; M = 4: 34.4 -> 31.1.
// Arithmetic contract: as the decode kernels (include/flute_amd.h): fp32 group scale on the group's partial sum (one-hot rows bit-exact).
// Reference: qgemm_device's main loop for small M (flute/csrc/qgemm_kernel.hpp:617-712); its Stream-K schedule over tiles
// (tile_scheduler_utils.hpp:460-481) is the flat (set, macro-step) stream here, its fix-up the in-workgroup reduction.
// Round 6, last: a 2-bit member (BITS = 2: a group = two unit rows of eight columns; the lookup address is v_bfe + v_lshl_or on a 16-entry image) and the
// activations RESIDENT in LDS where 4 XR rows x K fit in 64 KB (XRES: staged once per workgroup, no activation request in the loop).
// Host contract (api.hip: plan_persistm): num_bits = 4 or 2, M <= 16, K % 128 == 0, K >= 1024, group size 64 or 128 (128: K % 256 == 0 - a
// column's scale row is a whole number of dwords), N % 16 == 0, N * K / 2, N * (K / g) * 2 and M * K * 2 below 2^32 bytes, grid <= sets,
// XR = 1 / 2 / 4 for M <= 4 / 8 / 16, LDS = persistm_lds_bytes(NG, XR): 32 KB table + rings + 8 NG KB of partial tiles.
#pragma once
#include "qgemm_block.h"
#include "qgemm_fastm.h"

namespace flute_amd {

constexpr int PM_DW = 6;
// xr = activation requests per macro-step = ceil(M / 4) rounded up to 1, 2, 4 (a request = 4 rows x 256 B); the activation / scale rings
// are as deep as the weight ring (six macro-steps) where the LDS has room, three deep otherwise
// w = waves per workgroup: 8 (16 - four per SIMD, rings three deep, xr <= 2 - compiles and is correct but measured slower: not instantiated)
// xres (round 6, last step): the activations RESIDENT in LDS - all 4 xr rows x K staged once per workgroup (<= 64 KB: K <= 8192 / xr), no activation
// request in the loop, the scale ring six deep
__host__ __device__ constexpr int persistm_dx(int ng, int xr, int w = 8, bool xres = false) { return (!xres && (w == 16 || xr == 4 || (xr == 2 && ng == 3))) ? 3 : 6; }
// M <= 4 (xr = 1): the table image at a 256-B entry stride (64 KB, the upper half of every entry unused) - the v_perm that extracts a lane's
// byte then IS the lookup address, no shift: one VALU instruction less per pair
// (2 bits: 16 pair entries x 32 copies = 2 KB)
__host__ __device__ constexpr size_t persistm_table_bytes(int xr, int bits = 4) { return bits == 2 ? 2048 : (xr == 1 ? 65536 : 32768); }
__host__ __device__ constexpr size_t persistm_lds_bytes(int ng, int xr, int w = 8, bool xres = false, int bits = 4) {
    return persistm_table_bytes(xr, bits) + (xres ? (size_t)65536 + (size_t)w * 6 * 256 : (size_t)w * persistm_dx(ng, xr, w) * (xr * 1024 + 256)) + (size_t)w * ng * 1024;
}
// the swizzle constant of activation request r (rows 4 r .. 4 r + 3)
__host__ __device__ constexpr int pm_g(int r) { return (4 - r) & 3; }

// LDS-DMA of 4 B per lane (as dma16_buf): base + voff (per lane, range-checked) + soff (wave-uniform) -> LDS byte m0 + 4 * lane
__device__ __forceinline__ void dma4_buf(uint32_t voff, srd_t srd, uint32_t soff, uint32_t lds_addr) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 4\n\tbuffer_load_dword %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(srd), "s"(soff), "s"(lds_addr) : "memory");
}

template <typename T, int TILEP, int LG, int NG, int XR, int W = 8, bool XRES = false, int BITS = 4>
__global__ __launch_bounds__(W * 64) void qgemm_persistm_kernel(
    const uint32_t* __restrict__ Qp, const void* __restrict__ Sp, const void* __restrict__ Ap,
    const uint32_t* __restrict__ QM2, void* __restrict__ Dp, int N, int K, int M, int nsets) {
    using NT = Num<T>;
    constexpr int DX = persistm_dx(NG, XR, W, XRES), DW = PM_DW, XQ = XR;
    static_assert(W == 8 || W == 16, "waves per workgroup");
    static_assert(persistm_lds_bytes(NG, XR, W, XRES, BITS) <= 160 * 1024, "rings + partial tiles beside the table image");
    // 2-bit member: a unit row = 8 columns (a word = eight 4-bit pair indices), a group = TWO unit rows; the eight lanes of a unit share its words -
    // both of their quads hold steps 0 .. 3, so the quad broadcast is the 4-bit one - and a lookup address is v_bfe + v_lshl_or on a 16-entry table
    static_assert(BITS == 4 || BITS == 2, "3-bit layers: the per-wave MFMA kernel");
    constexpr int UPG = BITS == 4 ? 4 : 2, CPU = 16 / UPG;          // unit rows per group, columns per unit
    constexpr uint32_t MS_STRIDE = (uint32_t)W * 256u;               // bytes between a wave's consecutive macro-steps in a row
    static_assert(XR == 1 || XR == 2 || XR == 4, "activation requests per macro-step");
    static_assert(LG == 6 || LG == 7, "group size 64 or 128");
    static_assert(NG >= 1 && NG <= 3, "column groups per set");
    static_assert(DW % DX == 0 && DW % 2 == 0, "one unrolled body per weight-ring slot");
    constexpr int SPG = (1 << LG) / 32;                             // 32-k steps per group: 2 / 4
#ifdef FLUTE_PM_ABLATE   // development builds (tools/build_variant.sh): 1 activations / 2 scales / 8 weights through zero-byte descriptors (the requests
    constexpr int dbg = FLUTE_PM_ABLATE;                            // stay, their data does not travel), 4 no lookups / MFMAs, 16 / 32 no activation / scale requests at all
#else
    constexpr int dbg = 0;
#endif
    constexpr int NXR = ((dbg & 16) || XRES) ? 0 : XQ, NSR = (dbg & 32) ? 0 : 1;
    constexpr int NREQ = NG + NSR + NXR;                            // requests per macro-step: weights, scales, activations
    constexpr uint32_t XSLOT = (uint32_t)XR * 1024u;
    constexpr bool WIDE = XR == 1 && BITS == 4;                     // table image at a 256-B entry stride
    constexpr uint32_t X_BASE = (uint32_t)persistm_table_bytes(XR, BITS), XREG = XRES ? 65536u / W : (uint32_t)DX * XSLOT;
    constexpr uint32_t S_BASE = X_BASE + (uint32_t)W * XREG, SREG = (uint32_t)DX * 256u;
    constexpr uint32_t R_BASE = S_BASE + (uint32_t)W * SREG;
    constexpr int ENT = 256 / W, RUNS = 32 / W;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (lds_base_of(smem) != 0) __builtin_trap();                  // absolute LDS addresses
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15;                                      // MFMA row (weights) / column (activations)
    const int kg = lane >> 4;                                       // k-chunk of a 32-k step; also: unit of the lane's outputs
    const int ju = lane & 3;                                        // lane of its quad = the step whose words it requests; 4 bits: also the byte of the packed word = column of the unit
    const int uu = i16 / CPU;                                       // unit of the group
    const int jcol = i16 % CPU;                                     // column of the unit
    const int nunits = N / CPU;
    const int nms = K >> 7;                                         // macro-steps of the layer
    const int n_w = (nms - wave + W - 1) / W;                       // ... of this wave per set: w, w + 8, ...
    const int bid = blockIdx.x, grid = gridDim.x;
    const int n_items = (nsets - bid + grid - 1) / grid;            // sets of this workgroup
    const int P = n_items * n_w;                                    // the wave's (set, macro-step) pairs
    const uint32_t row2k = (uint32_t)K * 2u;                        // bytes of a unit row = of an activation row

    const srd_t lut_srd = make_srd(QM2, BITS == 4 ? 1024u : 64u);
    if constexpr (XRES) {
        // resident activations: macro-step ms of rows 4 r .. 4 r + 3 at X_BASE + ms * XSLOT + r KB (the rings' layout, indexed by the macro-step); the
        // waves share the requests, which are the launch's oldest: every counted wait below covers them, the prologue's barrier publishes them
        const srd_t xs0 = make_srd(Ap, (uint32_t)M * (uint32_t)K * 2u);
        for (int ms = wave; ms < (K >> 7); ms += W)
#pragma unroll
            for (int r = 0; r < XR; ++r)
                dma16_buf((uint32_t)(4 * r + (lane >> 4)) * (uint32_t)K * 2u + (uint32_t)((lane & 15) ^ (4 * (lane >> 4)) ^ pm_g(r)) * 16u, xs0, (uint32_t)ms * 256u,
                          X_BASE + (uint32_t)ms * XSLOT + (uint32_t)r * 1024u);
    }
    uint32_t lut_v = BITS == 4 ? buf_load4((uint32_t)(wave * ENT + (lane & (ENT - 1))) * 4u, lut_srd) : buf_load4((uint32_t)(lane & 15) * 4u, lut_srd);
    const srd_t q_srd = make_srd(Qp, (uint32_t)((size_t)nunits * row2k));
    const srd_t x_srd = make_srd(Ap, (uint32_t)M * row2k);
    const srd_t s_srd = make_srd(Sp, (uint32_t)((size_t)N * (size_t)(K >> LG) * 2));

    // ---- per-lane request offsets ----
    const uint32_t q_vo = (uint32_t)uu * row2k + (uint32_t)(ju * 32 + kg * 8) * 2u;
    const int xrs = lane >> 4, xb16 = lane & 15;
    uint32_t x_vo[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) x_vo[r] = (uint32_t)(4 * r + xrs) * row2k + (uint32_t)(xb16 ^ (4 * xrs) ^ pm_g(r)) * 16u;
    const uint32_t xreg = X_BASE + (uint32_t)wave * XREG;
    const uint32_t sreg = S_BASE + (uint32_t)wave * SREG;
    const uint32_t s_row = (uint32_t)(K >> LG) * 2u;                // bytes of a column's scale row
    auto scale_voff = [&](int set) {                                // lane = column 4 (4 g + u) + r of the set
        const int unit = set * (UPG * NG) + (lane >> 4) * UPG + (lane & 15) / CPU;
        const int col = (unit / TILEP) * (CPU * TILEP) + (unit % TILEP) + ((lane & 15) % CPU) * TILEP;
        return (lane < 16 * NG && unit < nunits) ? (uint32_t)col * s_row : 0xfffffff0u;
    };

    // ---- the three cursors of the flat stream (request weights / request scales + activations / compute), kept as countdowns and
    // additive offsets: every quantity below is wave-uniform and lives in SGPRs (the first version's multiplications and comparisons per
    // request were ~170 scalar instructions per macro-step, more than the VALU work) ----
    const uint32_t set_units = (uint32_t)grid * (UPG * NG);         // unit rows between a workgroup's consecutive sets
    const uint32_t w_jump = set_units * row2k - (uint32_t)n_w * MS_STRIDE;      // from a set's last macro-step to the next set's first
    uint32_t w_so = (uint32_t)(bid * (UPG * NG)) * row2k + (uint32_t)wave * 256u;
    int w_left = n_w, w_u0 = bid * (UPG * NG);
    auto groups_at = [&](int u0, int sets_left) { const int gl = (nunits - u0) / UPG; return sets_left > 0 ? (gl < NG ? gl : NG) : 0; };
    int w_sets = n_items, w_ng = groups_at(w_u0, w_sets);           // valid groups of the cursor's set (0: past the last pair)
    uint32_t x_so = (uint32_t)wave * 256u;
    int x_left = n_w, x_sets = n_items, x_set = bid;
    uint32_t s_vo = scale_voff(bid);
    int c_left = n_w, c_set = bid;
    // XRES: LDS offset of the macro-step whose steps 1 .. 3 are being issued / of the next pair's macro-step (whose step 0 is issued at step 3)
    uint32_t xo_cur = (uint32_t)wave * XSLOT, xo_nxt = n_w > 1 ? (uint32_t)(wave + W) * XSLOT : (uint32_t)wave * XSLOT;
    int xo_left = n_w > 1 ? n_w - 1 : n_w;                          // macro-steps from xo_nxt's to its set's end, itself included

    ring16_t q[DW][NG];
    auto request_w = [&](auto slot_tag) {
        constexpr int slot = decltype(slot_tag)::value;
        // (the set and the macro-step travel in the scalar offset, which the range check does not cover: a group past the layer's last
        // unit row - the last set of a layer whose groups NG does not divide - or a pair past the wave's last gets the zero-byte descriptor)
        static_for<NG>([&](auto g_tag) {
            constexpr int g = decltype(g_tag)::value;
            srd_t d = q_srd;
            d.z = (!(dbg & 8) && g < w_ng) ? d.z : 0;
            const uint32_t so = w_so + (uint32_t)(g * UPG) * row2k;
            ring16_t& dst = q[slot][g];
            const uint32_t vo = q_vo;
            asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen nt" : "=v"(dst) : "v"(vo), "s"(d), "s"(so) : "memory");
        });
        w_so += MS_STRIDE;
        if (--w_left == 0) { w_left = n_w; w_so += w_jump; w_u0 += (int)set_units; --w_sets; w_ng = groups_at(w_u0, w_sets); }
    };
    auto request_sx = [&](auto slot_tag) {
        constexpr int slot = decltype(slot_tag)::value;             // activation / scale ring slot
        srd_t ds = s_srd, dx = x_srd;
        ds.z = (x_sets > 0 && !(dbg & 2)) ? ds.z : 0;
        dx.z = (x_sets > 0 && !(dbg & 1)) ? dx.z : 0;
        const uint32_t sso = LG == 6 ? (x_so >> 6) : ((x_so >> 9) << 2);    // macro-step ms = x_so / 256: its groups' bytes in a scale row
        const uint32_t l_s = sreg + (uint32_t)slot * 256u, l_x = xreg + (uint32_t)slot * XSLOT;
        const uint32_t v0 = x_vo[0], v1 = x_vo[1], v2 = x_vo[2], v3 = x_vo[3], vs = s_vo;
        uint32_t keep;                                             // M0 is compiler-reserved: saved and restored (one wait state between its write and the LDS-DMA)
        if constexpr (NSR && NXR == 4)
            asm volatile("s_mov_b32 %0, m0\n\t"
                         "s_mov_b32 m0, %9\n\ts_nop 0\n\tbuffer_load_dword %1, %6, %8 offen lds\n\t"
                         "s_mov_b32 m0, %10\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %7, %11 offen lds\n\t"
                         "s_add_u32 m0, %10, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %3, %7, %11 offen lds\n\t"
                         "s_add_u32 m0, %10, 0x800\n\ts_nop 0\n\tbuffer_load_dwordx4 %4, %7, %11 offen lds\n\t"
                         "s_add_u32 m0, %10, 0xc00\n\ts_nop 0\n\tbuffer_load_dwordx4 %5, %7, %11 offen lds\n\t"
                         "s_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(vs), "v"(v0), "v"(v1), "v"(v2), "v"(v3), "s"(ds), "s"(dx), "s"(sso), "s"(l_s), "s"(l_x), "s"(x_so) : "memory", "scc");
        else if constexpr (NSR && NXR == 2)
            asm volatile("s_mov_b32 %0, m0\n\t"
                         "s_mov_b32 m0, %7\n\ts_nop 0\n\tbuffer_load_dword %1, %4, %6 offen lds\n\t"
                         "s_mov_b32 m0, %8\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %5, %9 offen lds\n\t"
                         "s_add_u32 m0, %8, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %3, %5, %9 offen lds\n\t"
                         "s_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(vs), "v"(v0), "v"(v1), "s"(ds), "s"(dx), "s"(sso), "s"(l_s), "s"(l_x), "s"(x_so) : "memory", "scc");
        else if constexpr (NSR && NXR == 1)
            asm volatile("s_mov_b32 %0, m0\n\t"
                         "s_mov_b32 m0, %6\n\ts_nop 0\n\tbuffer_load_dword %1, %3, %5 offen lds\n\t"
                         "s_mov_b32 m0, %7\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %4, %8 offen lds\n\t"
                         "s_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(vs), "v"(v0), "s"(ds), "s"(dx), "s"(sso), "s"(l_s), "s"(l_x), "s"(x_so) : "memory");
        else {
            if constexpr (NSR) dma4_buf(vs, ds, sso, l_s);
            if constexpr (NXR >= 1) dma16_buf(v0, dx, x_so, l_x);
            if constexpr (NXR >= 2) dma16_buf(v1, dx, x_so, l_x + 1024u);
            if constexpr (NXR == 4) { dma16_buf(v2, dx, x_so, l_x + 2048u); dma16_buf(v3, dx, x_so, l_x + 3072u); }
        }
        x_so += MS_STRIDE;
        if (--x_left == 0) {                                       // (wave-uniform) the next set's columns
            x_left = n_w; x_so = (uint32_t)wave * 256u; --x_sets; x_set += grid;
            s_vo = scale_voff(x_set);
        }
    };

    // ---- prologue requests, in the loop's order: body p refills the weight slot with W(p + 6) and the activation / scale slot with
    // S, X(p + DX) ----
    static_for<DW>([&](auto p_tag) {
        constexpr int pp = decltype(p_tag)::value;
        request_w(p_tag);
        if constexpr (pp >= DW - DX) request_sx(std::integral_constant<int, pp - (DW - DX)>{});
    });
    __builtin_amdgcn_sched_barrier(0);

    // ---- table image: entry e at [128 e, 128 e + 128) (WIDE: [256 e, 256 e + 128)): a wave writes RUNS runs of 8 entries (1 KiB, lane-linear) ----
    vm_wait_regs<DW * NG + DX * (NSR + NXR)>(lut_v);
    if constexpr (BITS == 2) {                                      // 16 entries: waves 0 / 1 write eight each
        const uint32_t te = (uint32_t)__builtin_amdgcn_ds_bpermute(((wave & 1) * 8 + (lane >> 3)) * 4, (int)lut_v);
        if (wave < 2) *reinterpret_cast<uint4*>(smem + (uint32_t)wave * 1024u + (uint32_t)lane * 16u) = make_uint4(te, te, te, te);
    } else {
        uint32_t te[RUNS];
#pragma unroll
        for (int u = 0; u < RUNS; ++u) te[u] = (uint32_t)__builtin_amdgcn_ds_bpermute((u * 8 + (lane >> 3)) * 4, (int)lut_v);
#pragma unroll
        for (int u = 0; u < RUNS; ++u)
            *reinterpret_cast<uint4*>(smem + (WIDE ? (uint32_t)((wave * RUNS + u) * 8 + (lane >> 3)) * 256u + (uint32_t)(lane & 7) * 16u
                                                   : (uint32_t)(wave * RUNS + u) * 1024u + (uint32_t)lane * 16u)) = make_uint4(te[u], te[u], te[u], te[u]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                  // the table image is complete

    // ---- per-lane LDS addresses ----
    const uint32_t lane_off2 = (uint32_t)(lane & 31) * (WIDE ? 4u : 8u);     // twice the copy offset (the address is halved after the v_perm); WIDE: the offset itself
    const uint32_t sel = 0x0c0c0400u | ((uint32_t)ju << 8);         // {copy offset x 2, byte ju of the word, 0, 0}
    const uint32_t nib_shift = (uint32_t)jcol * 4u, lane_off4 = (uint32_t)(lane & 31) * 4u;      // 2 bits: the column's nibble, the copy offset
    // activation fragment of step s: row i16 = 4 r + mm, chunk 4 s + kg -> request r's KB, position 16 mm + 4 (s ^ mm) + (kg ^ g(r))
    uint32_t xa[4];
    {
        // (rows past the requested 4 XR alias the requested ones: their products are never stored)
        const int r = (i16 % (4 * XR)) >> 2, mm = i16 & 3;
#pragma unroll
        for (int s = 0; s < 4; ++s) xa[s] = (XRES ? X_BASE : xreg) + (uint32_t)r * 1024u + (uint32_t)(16 * mm + 4 * (s ^ mm) + (kg ^ pm_g(r))) * 16u;
    }
    const uint32_t sc_a = sreg + (uint32_t)kg * 16u;                // the four columns of the lane's output unit kg: + 64 g

    uint32_t v[2][NG][4];
    ring16_t xb[2];
    ring16_t sc[2][NG];
    f32x4_t accf[NG], part[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) { accf[g] = f32x4_t{0.f, 0.f, 0.f, 0.f}; part[g] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }

    // the LDS reads of step s of the macro-step in ring slots (qs, xs): 4 NG lookups + the fragment (+ NG scale reads at s = 0)
    auto issue_step = [&](auto qs_tag, auto s_tag, uint32_t xoff = 0u) {       // xoff (XRES): the macro-step's place in the resident activations
        constexpr int qs = decltype(qs_tag)::value, s = decltype(s_tag)::value;
        constexpr int xs = qs % DX;
        if constexpr (!(dbg & 4)) static_for<NG>([&](auto g_tag) {
            constexpr int g = decltype(g_tag)::value;
            uint32_t ad[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const uint32_t wsrc = (uint32_t)__builtin_amdgcn_mov_dpp((int)q[qs][g][c], s * 0x55, 0xF, 0xF, true);       // quad_perm [s, s, s, s]
                if constexpr (BITS == 2) ad[c] = (__builtin_amdgcn_ubfe(wsrc, nib_shift, 4u) << 7) | lane_off4;
                else ad[c] = WIDE ? __builtin_amdgcn_perm(wsrc, lane_off2, sel) : __builtin_amdgcn_perm(wsrc, lane_off2, sel) >> 1;
            }
            asm volatile("" : "+v"(ad[0]), "+v"(ad[1]), "+v"(ad[2]), "+v"(ad[3]));
#pragma unroll
            for (int c = 0; c < 4; ++c) v[s & 1][g][c] = lds_lookup32(ad[c]);
        });
        ring16_t& dst = xb[s & 1];
        if constexpr (XRES) {
            const uint32_t xaddr = xa[s] + xoff;
            asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(xaddr) : "memory");
        } else {
            const uint32_t xaddr = xa[s];
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(xaddr), "n"(xs * (int)XSLOT) : "memory");
        }
        if constexpr (s == 0) {
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                ring16_t& sd = sc[qs & 1][g];
                const uint32_t saddr = sc_a + (uint32_t)g * 64u;
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(sd) : "v"(saddr), "n"(xs * 256) : "memory");
            }
        }
    };
    // (Measured and dropped, profiles/r06/call35_persistm_two_step_lookahead_dropped.log: the LDS reads TWO steps ahead of their MFMA at one group per
    // set - three buffers, one lgkmcnt counting two steps' reads: equal within the noise, 29.5 against 30.1 us on 28672 x 8192, 30.4 against 29.2 on
    // 8192 x 28672 - the LDS latency is not what a wave waits for.)
    // step s's reads have returned once at most Y younger LDS operations are outstanding
    auto wait_step = [&](auto s_tag, auto younger_tag) {
        constexpr int s = decltype(s_tag)::value, Y = decltype(younger_tag)::value;
        uint32_t (&vv)[NG][4] = v[s & 1];
        ring16_t& xx = xb[s & 1];
        if constexpr (NG == 1)
            asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(vv[0][0]), "+v"(vv[0][1]), "+v"(vv[0][2]), "+v"(vv[0][3]), "+v"(xx) : "n"(Y) : "memory");
        else if constexpr (NG == 2)
            asm volatile("s_waitcnt lgkmcnt(%9)" : "+v"(vv[0][0]), "+v"(vv[0][1]), "+v"(vv[0][2]), "+v"(vv[0][3]),
                         "+v"(vv[1][0]), "+v"(vv[1][1]), "+v"(vv[1][2]), "+v"(vv[1][3]), "+v"(xx) : "n"(Y) : "memory");
        else
            asm volatile("s_waitcnt lgkmcnt(%13)" : "+v"(vv[0][0]), "+v"(vv[0][1]), "+v"(vv[0][2]), "+v"(vv[0][3]),
                         "+v"(vv[1][0]), "+v"(vv[1][1]), "+v"(vv[1][2]), "+v"(vv[1][3]),
                         "+v"(vv[2 % NG][0]), "+v"(vv[2 % NG][1]), "+v"(vv[2 % NG][2]), "+v"(vv[2 % NG][3]), "+v"(xx) : "n"(Y) : "memory");
    };
    auto name_scales = [&](auto par_tag) {                          // (the scale reads ride in step 0's group: returned with it)
        constexpr int par = decltype(par_tag)::value;
        static_for<NG>([&](auto g_tag) { ring16_t& sd = sc[par][decltype(g_tag)::value]; asm volatile("" : "+v"(sd) : : "memory"); });
    };
    // the data of the macro-step in weight slot qs has landed once at most Y younger requests are outstanding
    auto wait_requests = [&](auto qs_tag, auto younger_tag) {
        constexpr int qs = decltype(qs_tag)::value, Y = decltype(younger_tag)::value;
        ring16_t (&qq)[NG] = q[qs];
        if constexpr (NG == 1) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(qq[0]) : "n"(Y) : "memory");
        else if constexpr (NG == 2) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(qq[0]), "+v"(qq[1]) : "n"(Y) : "memory");
        else asm volatile("s_waitcnt vmcnt(%3)" : "+v"(qq[0]), "+v"(qq[1]), "+v"(qq[2 % NG]) : "n"(Y) : "memory");
    };

    // ---- a set's end: partial tiles -> LDS -> one barrier -> every wave sums 32 outputs per group -> a second barrier ----
    auto finish_set = [&](int set) {
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            *reinterpret_cast<float4*>(smem + R_BASE + (uint32_t)(wave * NG + g) * 1024u + (uint32_t)lane * 16u) = make_float4(accf[g][0], accf[g][1], accf[g][2], accf[g][3]);
            accf[g] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // (not __syncthreads(): its fence would drain the request pipeline)
        __builtin_amdgcn_s_barrier();
        constexpr int OPW = 256 / W;                                // outputs per wave and column group
        if (lane < OPW) {
            const int f = wave * OPW + lane;                        // float f of a partial tile: lane f / 4 of the MFMA layout, register f % 4
            const int ls = f >> 2, r = f & 3;
            const int m = ls & 15;
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                float sum = 0.f;
#pragma unroll
                for (int w2 = 0; w2 < W; ++w2) sum += __builtin_bit_cast(float, lds_ld32(R_BASE + (uint32_t)(w2 * NG + g) * 1024u + (uint32_t)f * 4u));
                const int c16 = 4 * (ls >> 4) + r;                   // MFMA weight row = column 16 g + c16 of the set
                const int unit = set * (UPG * NG) + UPG * g + c16 / CPU;
                const int col = (unit / TILEP) * (CPU * TILEP) + (unit % TILEP) + (c16 % CPU) * TILEP;
                if (m < M && unit < nunits) reinterpret_cast<uint16_t*>(Dp)[(size_t)m * N + col] = NT::from_float(sum);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                              // the partial tiles may be overwritten
    };

    // ---- the flat loop: one body per weight-ring slot; step 3 of a macro-step waits for the next macro-step's data and issues its
    // step-0 reads, the requests of six / three macro-steps ahead follow the last read of the slots they refill ----
    wait_requests(std::integral_constant<int, 0>{}, std::integral_constant<int, (DX - 1) * NREQ>{});
    issue_step(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, xo_cur);
    auto body = [&](auto qs_tag) {
        constexpr int qs = decltype(qs_tag)::value;
        constexpr int nq = (qs + 1) % DW;
        using nq_t = std::integral_constant<int, nq>;
        static_for<4>([&](auto s_tag) {
            constexpr int s = decltype(s_tag)::value;
            if constexpr (s < 3) {
                issue_step(qs_tag, std::integral_constant<int, s + 1>{}, xo_cur);
            } else {
                wait_requests(nq_t{}, std::integral_constant<int, (DX - 2) * NREQ>{});
                issue_step(nq_t{}, std::integral_constant<int, 0>{}, xo_nxt);
            }
            wait_step(s_tag, std::integral_constant<int, (s < 3) ? 4 * NG + 1 : (5 * NG + 1 < 15 ? 5 * NG + 1 : 15)>{});   // (lgkmcnt counts to 15)
            if constexpr (s == 0) name_scales(std::integral_constant<int, qs & 1>{});
            const u32x4_t b = {xb[s & 1][0], xb[s & 1][1], xb[s & 1][2], xb[s & 1][3]};
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const u32x4_t a = {v[s & 1][g][0], v[s & 1][g][1], v[s & 1][g][2], v[s & 1][g][3]};
                if constexpr (dbg & 4) part[g][0] += __builtin_bit_cast(float, b[0] ^ q[qs][g][0]);
                else if constexpr (s % SPG == 0) part[g] = Mfma<T>::run(a, b, f32x4_t{0.f, 0.f, 0.f, 0.f});
                else part[g] = Mfma<T>::run(a, b, part[g]);
                if constexpr (s % SPG == SPG - 1) {
                    // g = 64: group s / 2 of the macro-step's two; g = 128: the dword holds the groups of macro-steps (ms & ~1, ms | 1)
                    const bool hi = (LG == 6) ? (s / SPG) != 0 : ((wave & 1) != 0);    // (ms = wave + 8 i: its parity is the wave's)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const uint32_t w = sc[qs & 1][g][r];
                        accf[g][r] = __builtin_fmaf(part[g][r], scale_to_float<T>(hi ? (w >> 16) : (w & 0xffffu)), accf[g][r]);
                    }
                }
            }
        });
        // every read of the slots (qs, qs % 3) has returned: refill them
        request_w(qs_tag);
        request_sx(std::integral_constant<int, qs % DX>{});
        if (--c_left == 0) { c_left = n_w; finish_set(c_set); c_set += grid; }     // (wave-uniform)
        if constexpr (XRES) {
            xo_cur = xo_nxt;
            if (--xo_left == 0) { xo_left = n_w; xo_nxt = (uint32_t)wave * XSLOT; } else xo_nxt += (uint32_t)W * XSLOT;
        }
    };
    static_assert(DW == 6, "the loop below spells the six bodies out");
    for (int left = P;;) {
        body(std::integral_constant<int, 0>{}); if (--left == 0) break;
        body(std::integral_constant<int, 1>{}); if (--left == 0) break;
        body(std::integral_constant<int, 2>{}); if (--left == 0) break;
        body(std::integral_constant<int, 3>{}); if (--left == 0) break;
        body(std::integral_constant<int, 4>{}); if (--left == 0) break;
        body(std::integral_constant<int, 5>{}); if (--left == 0) break;
    }
    // the reads and requests past the end (zero-byte descriptors) are dead; nothing may land in LDS after the wave has gone
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
}

}  // namespace flute_amd
