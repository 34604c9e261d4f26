// Split-K block kernel (round 4; reworked in round 6): 128 / 64 x 128 / 64 output tiles, K split over workgroups, partial tiles
// combined INSIDE the launch.
//
// The chip-filling schedule for 33 <= M <= ~1024 that rounds 2 and 3 lacked: the reference's Stream-K hands every CTA an equal
// share of tiles_M x tiles_N x tiles_K for any M (flute/csrc/tile_scheduler_utils.hpp:460-481, fix-up :58-211, main loop
// qgemm_kernel.hpp:617-712).  Here the output is cut into tiles of RT x 16 rows x (256 / KP) columns, each cut `splitk` ways in K -
// one workgroup per (tile, slice); the partial tiles of an output tile meet through the workspace with write-through
// stores and one arrival word (xwg.h): no second launch, no release fence.  M = 256 x 4096 x 11008: 172 tiles of 128 x 128 x 1
// slice; M = 1024 x 4096^2: 256 tiles; M = 256 x 4096^2: 256 tiles of 64 x 64 over all of K (KP = 4, round 6; before: 128 tiles of
// 64 x 128 x 2 slices).
//
// Workgroup = 8 compute waves = 8 / KP column groups (32 columns = two MFMA column tiles each, as qgemm_block2.h) x KP K parts,
// + 4 loader waves (768 threads).  The waves of K part p run the block2 pipeline on the 64-k chunks p, p + KP, p + 2 KP, ... of the
// workgroup's K range (interleaved since round 6: the waves of a column group, kept in step by the barriers, walk neighbouring
// chunks) with their own three 64-k activation stages; every weight of the workgroup's range is dequantised exactly once (8
// lookups + 8 multiplies per RT x 2 MFMAs of a 32-k half step), a wave holds RT row tiles x 2 column tiles of accumulators.
// What differs from qgemm_block2.h (each step measured: DESIGN.md 3.2d, profiles/r04/, profiles/r06/):
//   * activation pieces are 8 rows x 128 B - WHOLE cache lines (a request is priced per line it touches); fragment
//     swizzle sk_swz (= qgemm_tile.h's swz_x); every piece is issued by a loader wave (RT 8 / 4 per loader and step) - the compute
//     waves' in-order streams hold one weight request per step and nothing else;
//   * the scales of the workgroup's whole K range are fetched once, by the prologue: one image of eight 8-group blocks x 32 columns
//     per column group (4 KB), shared by its KP waves: no scale request - and no sink request - per step;
//   * ONE whole-line weight request per compute wave and step, 8 unit rows x 128 B, into registers.  The texture addresser is paid per
//     line a QUAD of lanes touches (round 6: the same bytes asked for as 2 rows x 512 B or lane-linearly - results wrong - ran M = 256
//     on 4096^2 in 15.5 / 15.7 instead of 16.55 us, profiles/r06/call8_*.log, call12_*.log).  So lane bit 1 of an MFMA weight row is a
//     FIELD bit, not a unit bit: the four lanes of a quad are two units x two fields, each loads ONE of the two half steps of its
//     unit - two lines per quad instead of four - and takes the other from lane ^ 2 by DPP quad_perm (rounds 4 / 5: partner lane
//     r16 ^ 8 by row_ror:8, four lines per quad): 16.33 -> 15.53 us, same box.  (The lane-linear form through a wave-private LDS slot
//     and two ds_read_b128 per step was built too: 15.8 - 16.1; removed.)  The epilogue's v_permlane16_swap puts four consecutive
//     columns back into every lane;
//   * the operand registers are double-buffered: the next half step's LDS reads are issued behind the first RT / 2 row
//     tiles' MFMAs;
//   * a batch past the end of the K range goes through a zero-byte descriptor (returns zeros without a trip to L2): the counted
//     waits keep their shape, the drain behind the last step does not wait for re-reads;
//   * XCD-aware tile order when K is not split: groups of E row tiles that share a column tile's weights run on one XCD.
// Epilogue: (1) the K parts exchange row tiles through LDS (every wave ends with RT / KP row tiles x 2 column tiles of the
// workgroup's K range), (2) splitk > 1: the E form of xwg.h when the slices divide a wave's row tiles (2 slices: RT / KP even; 4:
// RT = 8, KP = 2; slice s owns row tiles s, s + nsh, ... of every wave), the L form otherwise; sums are taken in a fixed order (K
// parts ascending; owner first, then the other slices ascending; L: all slices ascending), so the result does not depend on arrival order.
// Arithmetic contract as qgemm_block2.h: w^ = round_T(lut * s) (packbits_utils.hpp:139), fp32 accumulation, one rounding.
// Host contract (api.hip, plan_splitk): 2 or 4 bits, G % 8 == 0, K % k_per_split == 0, k_per_split % (KP * max(64, g)) == 0,
// at most 64 scale groups (+ alignment slack: eight 8-group blocks) per workgroup K range, splitk * tiles * RT / KP * 16 KB of slabs < 2^31.
#pragma once
#include <utility>

#include "qgemm_block.h"
#include "xwg.h"

namespace flute_amd {

struct SplitKArgs {
    const void* A;          // [M,K] T
    const uint32_t* Q;      // [P,K/2] packed
    void* D;                // [M,N] T
    const void* S;          // [N,G] T
    const uint32_t* QM2;    // [4^b] pair table
    float* partial;         // [splitk][tile] x 8 RT / KP x 2 KB fp32 partial tiles in fragment order (write-through), splitk > 1
    uint32_t* state;        // two words per output tile, zero before and after the launch (xwg.h)
    int M, N, K, G, lg;
    int tiles_m;            // row tiles (fastest in the block order: the row tiles of a column tile are neighbours)
    int splitk, k_per_split;
    // XCD-aware tile order (splitk == 1, tiles_m = E P with E = 2^pair_e row tiles per group and P a power of two): pair_lg =
    // log2(P), pair_c8 = (P x column tiles) rounded down to a multiple of 8; pair_lg < 0: tiles in their natural order
    int pair_lg, pair_c8, pair_e;
};

// RT row tiles per workgroup: 8 (128-row tiles) or - round 4, for outputs whose 128-row tiles leave most of the chip idle
// - 4 (64-row tiles: twice the tiles, half the slab per slice, every weight dequantised by twice as many workgroups)
__host__ __device__ constexpr int splitk_stage_bytes(int rt) { return rt * 2 * 1024; }     // RT x 16 rows x 64 k: 1-KB pieces
constexpr int SK_SCALE_GROUP = 4096;                               // a column group's scale image: eight 8-group blocks x 32 columns x 16 B
constexpr int SK_LOADERS = 4;                                      // loader waves beside the eight compute waves
constexpr int SK_THREADS = 512 + 64 * SK_LOADERS;
// position swizzle of an 8-row x 8-chunk activation piece (rows 8 rh .. 8 rh + 7 of a 16-row tile): the 16 lanes of every
// ds_read_b128 lane group hit 16 different 16-B slots of the 256-B bank row (as qgemm_tile.h's swz_x; checked for the
// lane groups of MI355X_MICROARCH.md's LDS table by tests/test_host.py)
__host__ __device__ constexpr int sk_swz(int row8, int rh) { return (row8 >> 1) | (rh << 2); }
__host__ __device__ constexpr int splitk_lds_bytes(int bits, int rt = 8, int kp = 2) {
    return (128 << (2 * bits)) + kp * BLK_STAGES * splitk_stage_bytes(rt) + (8 / kp) * SK_SCALE_GROUP;
}

// A hidden load's value, handed over in a NEW register once at most N younger loads are outstanding.  (The wait and the
// move are one asm statement: with the usual "+v" wait hipcc was seen copying the destination into the first register
// of the quad the table store wants BEFORE the wait - the audit caught it in the 768-thread variant.)
template <int N> __device__ __forceinline__ uint32_t lut_word_after(const uint32_t& hidden) {
    uint32_t v;
    asm volatile("s_waitcnt vmcnt(%2)\n\tv_mov_b32 %0, %1" : "=&v"(v) : "v"(hidden), "n"(N) : "memory");
    return v;
}

// KP (round 6): K parts per workgroup.  2: four column groups x two K parts - a 128-column tile (rounds 4 / 5).  4: TWO column
// groups x FOUR K parts - a 64-column tile over twice the K range per workgroup: M = 256 on 4096 x 4096 becomes 4 x 64 tiles of
// 64 x 64 over ALL of K, 256 workgroups and NO cross-workgroup seam (the E-form seam of the 2-slice plan cost 2.9 us of a 19.4-us
// launch: drain, arrival, poll, load - a chain of fabric round trips); the price is twice the activation bytes per workgroup
// (64 rows x 4096 k instead of 64 x 2048), pulled by the same four loader waves, one per K part.  A wave's work is the same in
// both forms: 32 columns x RT row tiles x 1/KP of the workgroup's K range.
template <typename T, int TILEP, int BITS = 4, int RT = 8, int KP = 2>
__global__ __launch_bounds__(SK_THREADS) void qgemm_splitk_kernel(const SplitKArgs args) {
    using NT = Num<T>;
    static_assert(BITS == 4 || BITS == 2, "3-bit layers: qgemm_block3.h / qgemm_tile.h");
    constexpr int J = 16 / BITS;
    constexpr int U = 32 / J;                                      // units per wave (32 columns)
    constexpr int FPT = 16 / U;                                    // fields per column tile and unit
    constexpr int FB = 2 * BITS;
    static_assert(RT == 8 || RT == 4, "row tiles per workgroup");
    static_assert(KP == 2 || KP == 4, "K parts per workgroup");
    static_assert(RT % KP == 0, "every K part keeps RT / KP row tiles");
    constexpr int NT2 = 2, NWN = 8 / KP;                           // column tiles per wave, column groups (waves per K part)
    constexpr int HR = RT / KP;                                    // row tiles a wave keeps after the K parts' exchange
    constexpr int HF = RT / 2;                                     // the row tiles of a half step behind which the next half step's operands are fetched
    constexpr int SK_STAGE = splitk_stage_bytes(RT);
    constexpr int LPR = 8 / HF;                                    // next-half lookups issued behind each of the first HF row tiles
    constexpr int LDW = SK_LOADERS, NTHR = SK_THREADS;
    constexpr int LPP = LDW / KP;                                  // loader waves per K part (2 / 1)
    constexpr int LPW = RT * 2 / LPP;                              // activation pieces per loader wave and step (8 RT / 4)
#ifdef FLUTE_SK_ABLATE   // development builds (tools/build_variant.sh): 1 no activation requests in the loop, 2 no weight requests,
    constexpr int dbg = FLUTE_SK_ABLATE;                           // 4 no MFMA, 8 no table lookups, 16 no fragment reads, 32 no barriers, 64 one step only, 128 no seam
#else
    constexpr int dbg = 0;
#endif
    // Request distances on the three ring slots.  Activations: XA = 3 - batch t + 3 goes into stage t % 3 right behind barrier (B) of
    // step t (every wave has read the stage's second fragments in front of it) - or 2 - batch t + 2 behind barrier (A) into the stage
    // step t - 1 left (rounds 4 / 5).  Measured per form (profiles/r06/call7_*.log, same box, M = 256 on 4096^2): two K parts x 2 slices
    // 18.25 -> 17.78 us with 3, four K parts 16.3 -> 16.65 - twice the activation bytes in flight per CU did not pay there.  Weights in
    // registers: the same distance, w[t % 3] is dead once half step 0 of step t has shuffled its second half out of it.
#ifdef FLUTE_SK_XA
    constexpr int XA = FLUTE_SK_XA;
#else
    constexpr int XA = (KP == 4) ? 2 : 3;
#endif
    static_assert(XA == 2 || XA == 3, "activation request distance");
#ifdef FLUTE_SK_TWO_BARRIERS   // development A/B: both barriers of a step at request distance 3 too (rounds 4 / 5)
    constexpr bool ONE_BARRIER = false;
#else
    constexpr bool ONE_BARRIER = true;
#endif
    // (Measured and dropped in round 6, profiles/r06/call22_lookups_two_half_steps_ahead_dropped.log: the table lookups TWO half steps ahead with
    // their scale multiplies between the MFMAs of the half step before they are needed - no LDS drain and no VALU in front of a half step's
    // first MFMA; M = 256 on 4096^2 15.57 -> 15.9 us, bf16 16.7 -> 18.1: VALU between a wave's MFMAs delays its in-order MFMA issue by more
    // than the multiplies cost in front of the barrier, where they overlap the wait for the slowest wave.)
    constexpr int WA = XA;                                         // weight request distance
    constexpr int NW = (dbg & 2) ? 0 : 1;                          // weight requests per compute wave and step in the loop
    constexpr int LUT_BYTES = (1 << (2 * BITS)) * 128;
    constexpr int X_BASE = LUT_BYTES;
    constexpr int SC_BASE = X_BASE + KP * BLK_STAGES * SK_STAGE;

    SplitKArgs a = args;
    {
#define FLUTE_OPAQUE(x) asm volatile("" : "+s"(x))
        FLUTE_OPAQUE(a.A); FLUTE_OPAQUE(a.Q); FLUTE_OPAQUE(a.D); FLUTE_OPAQUE(a.S); FLUTE_OPAQUE(a.QM2);
        FLUTE_OPAQUE(a.partial); FLUTE_OPAQUE(a.state); FLUTE_OPAQUE(a.M); FLUTE_OPAQUE(a.N); FLUTE_OPAQUE(a.K);
        FLUTE_OPAQUE(a.G); FLUTE_OPAQUE(a.lg); FLUTE_OPAQUE(a.tiles_m); FLUTE_OPAQUE(a.splitk); FLUTE_OPAQUE(a.k_per_split);
        FLUTE_OPAQUE(a.pair_lg); FLUTE_OPAQUE(a.pair_c8); FLUTE_OPAQUE(a.pair_e);
#undef FLUTE_OPAQUE
    }
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (lds_base_of(smem) != 0) __builtin_trap();
#ifdef FLUTE_STAMPS   // development build (tools/stamps_splitk.py): 100 MHz wall-clock stamps per wave behind the slabs
    uint64_t stamp[12];
    for (int i = 0; i < 12; ++i) stamp[i] = 0;
    stamp[0] = wall_clock64();
#define FLUTE_SKSTAMP(i) stamp[i] = wall_clock64()
#define FLUTE_SKSTAMP_FLUSH() do { __builtin_amdgcn_s_waitcnt(0); stamp[8] = wall_clock64(); \
        if ((threadIdx.x & 63) == 0) { uint64_t* o = reinterpret_cast<uint64_t*>(a.partial + (size_t)a.splitk * a.M * a.N) + ((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 12; \
            for (int i = 0; i < 12; ++i) o[i] = stamp[i]; } } while (0)
#else
#define FLUTE_SKSTAMP(i)
#define FLUTE_SKSTAMP_FLUSH()
#endif

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int r16 = lane & 15;
    const int q4 = lane >> 4;
    // Weight row r16 of a column tile (= MFMA row) is unit u8, field fsel.  Round 6: lane bit 1 is a FIELD bit (it was a unit bit) -
    // r16 = (b3 b2 b1 b0): unit = (b3 b2 b0) [U = 4: (b2 b0)], field = b1 [U = 4: (b3 b1)] - so that the lanes of a quad that share a
    // unit can hold its two half steps and a weight request touches TWO lines per quad of lanes instead of four (below).
    const int u8 = (((r16 >> 2) % (U / 2)) << 1) | (r16 & 1);
    const int fsel = ((r16 >> 2) / (U / 2)) * 2 + ((r16 >> 1) & 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wg = wave % NWN;                                     // column group
    const int kh = wave / NWN;                                     // K part

    int tile = blockIdx.x, split = 0;
    if (a.splitk > 1) { split = tile % a.splitk; tile /= a.splitk; }
    // Block b runs on XCD b % 8 (observed; speed only).  Natural order (row tiles fastest) puts the row tiles of a column
    // tile on neighbouring XCDs: each fetches the column tile's weights for itself (FETCH 1.9x the algorithmic bytes at
    // M = 256, profiles/r04_rocprof).  Here a GROUP of E = 2^pair_e row tiles x column tile = "group column" c goes to XCD
    // c % 8 as that XCD's consecutive blocks: the weights are fetched once per group, an XCD reads E row tiles of X.
    int tm_idx, tn_idx;
    if (a.pair_lg >= 0) {
        const int em = (1 << a.pair_e) - 1;
        int c, e;
        if (tile < (a.pair_c8 << a.pair_e)) { const int i = tile >> 3; c = (i >> a.pair_e) * 8 + (tile & 7); e = i & em; }
        else { c = tile >> a.pair_e; e = tile & em; }              // the last, incomplete group of eight: natural order
        tn_idx = c >> a.pair_lg;
        tm_idx = ((c & ((1 << a.pair_lg) - 1)) << a.pair_e) + e;
    } else {
        tm_idx = tile % a.tiles_m;
        tn_idx = tile / a.tiles_m;
    }
    const int m0 = tm_idx * (RT * 16);
    const int unit0 = (tn_idx * NWN + wg) * U;
    // Round 6: the K parts of a workgroup are INTERLEAVED - step t of part p is the 64-k chunk t KP + p of the workgroup's K range
    // (rounds 4 / 5: part p owned the contiguous p-th share): the waves of a column group ask for KP x 128 contiguous bytes of
    // every unit row per step, the loader waves for KP x 128 B of every activation row, and ONE scale image serves a column group.
    const int kpart = a.k_per_split / KP;                          // k per part
    const int kwg = split * a.k_per_split;                         // the workgroup's K range: [kwg, kwg + k_per_split)
    const int kbeg = kwg + kh * 64;                                // this wave's step t: k = kbeg + t * (64 KP) .. + 63
    constexpr int KSTEP = 64 * KP;
    const int nsteps = (dbg & 64) ? 1 : (kpart >> 6);              // (ablation 64: ONE step - what a launch costs around its main loop)
    const uint32_t row_bytes = (uint32_t)a.K * 2u;

    const srd_t x_srd = make_srd(a.A, (uint32_t)min((size_t)a.M * a.K * 2, (size_t)0xfffffff0u));
    const srd_t w_srd = make_srd(reinterpret_cast<const char*>(a.Q) + (size_t)unit0 * row_bytes, (uint32_t)U * row_bytes);
    const srd_t s_srd = make_srd(a.S, (uint32_t)min((size_t)a.N * a.G * 2, (size_t)0xfffffff0u));
    // A batch past the end of the K range goes through a descriptor of ZERO bytes: every lane is out of range, the request returns
    // zeros from the texture unit without a trip to L2 / HBM - the counted waits keep their shape, and the drain behind the last step
    // (vmcnt(0) before the K parts' exchange) does not wait for re-reads of the last step (rounds 4 / 5).  Every K offset is wave-uniform
    // and travels in the scalar offset.
    auto live = [&](srd_t d, int u) { d.z = (u < nsteps) ? d.z : 0; return d; };

    // ---- activation pieces (loader waves): piece p = 2 rt + rh of a stage = rows 16 rt + 8 rh .. + 7, 128 B (the 64 k of the step) each;
    // lane L fetches the 16-B chunk (L & 7) ^ sk_swz(L >> 3, rh) of row L >> 3 and the DMA writes it lane-linearly, so LDS
    // position pos of row8 holds chunk pos ^ swz.  Rows past M: their byte offset is past the descriptor's range and reads as
    // zero (voffset is what is checked).
    const int row8 = lane >> 3;
    const uint32_t x_grp = (uint32_t)X_BASE + (uint32_t)kh * (BLK_STAGES * SK_STAGE);      // this K part's three stages
    // ---- weights: ONE request per compute wave and step, 8 unit rows x 128 B (whole cache lines): lane (r16, q4) fetches chunk
    // 4 ((r16 >> 1) & 1) + q4 of its unit, i.e. the half step its field bit names - a quad of lanes = two units x two half steps = two
    // cache lines.  The half a lane lacks sits in lane ^ 2 of its quad: DPP quad_perm, four moves per half step.
    const uint32_t w_voff = (uint32_t)u8 * row_bytes + (uint32_t)(((r16 >> 1) & 1) * 4 + q4) * 16u;

    u32x4_t w[BLK_STAGES];                                         // the weight pieces of three steps
    auto issue_w = [&](auto slot_tag, int u) {                     // batch u = the weight piece of this wave's K step u
        constexpr int slot = decltype(slot_tag)::value;
        const uint32_t k0 = (uint32_t)(kbeg + min(u, nsteps - 1) * KSTEP);
        w[slot] = buf_load16(w_voff, live(w_srd, u), k0 * 2u);
    };
    // the words of half step h out of a step's piece: quad_perm [2h, 2h + 1, 2h, 2h + 1] - every lane reads the lane of its quad that
    // holds half step h of its unit (itself or lane ^ 2)
    auto half_words = [&](const u32x4_t& own, auto h_tag) {
        constexpr int h = decltype(h_tag)::value;
        u32x4_t r;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            r[j] = (uint32_t)__builtin_amdgcn_update_dpp((int)own[j], (int)own[j], h == 0 ? 0x44 : 0xee, 0xf, 0xf, false);
        return r;
    };

    // ---- pair-table words first (oldest in the queue: their wait below leaves everything else in flight) ----
    constexpr int ENT = 1 << (2 * BITS);
    constexpr int LUT_R = (ENT * 8 + NTHR - 1) / NTHR;             // 16-B table pieces per thread
    const srd_t lut_srd = make_srd(a.QM2, (uint32_t)(4 * ENT));
    uint32_t lutw[LUT_R];
#pragma unroll
    for (int r = 0; r < LUT_R; ++r) lutw[r] = buf_load4((uint32_t)((tid + NTHR * r) >> 3) * 4u, lut_srd);   // past the table: reads 0, not written
    if (wave >= 8) {
        // ---- loader wave L: K part L / LPP, pieces LPW (L % LPP) .. + LPW - 1 of that part's stage, every step ----
        const int L = wave - 8, lkh = L / LPP, lp0 = (L % LPP) * LPW;
        const int lkbeg = kwg + lkh * 64;
        uint32_t lx[LPW];
#pragma unroll
        for (int i = 0; i < LPW; ++i) {
            const int rt = (lp0 + i) >> 1, rh = (lp0 + i) & 1;
            lx[i] = (uint32_t)(((size_t)(m0 + rt * 16 + rh * 8 + row8) * a.K + (((lane & 7) ^ sk_swz(row8, rh)) * 8)) * 2);
        }
        const uint32_t ldst = (uint32_t)X_BASE + (uint32_t)lkh * (BLK_STAGES * SK_STAGE) + (uint32_t)lp0 * 1024u;
        auto lbatch = [&](int u) {
            const uint32_t k0 = (uint32_t)(lkbeg + min(u, nsteps - 1) * KSTEP);
            const uint32_t dst = ldst + (uint32_t)(u % BLK_STAGES) * SK_STAGE;
            const srd_t d = live(x_srd, u);
#pragma unroll
            for (int i = 0; i < LPW; ++i) dma16_buf(lx[i], d, k0 * 2u, dst + (uint32_t)i * 1024u);
        };
        // (Measured and dropped at request distance 2, profiles/r06/call24_issue_points_dropped.log, M = 256 on 4096^2: a third batch in the
        // prologue 15.0 us, half of a batch behind each barrier 15.0, the batch behind barrier (B) with one barrier per step 15.4 - against 14.8 - 15.4.)
        lbatch(0);
        lbatch(1);
        if constexpr (XA == 3) lbatch(2);
        [&]<int... R>(std::integer_sequence<int, R...>) {
            ([&] {
                const uint32_t lv = lut_word_after<LUT_R - 1 - R + XA * LPW>(lutw[R]);
                const int p = tid + NTHR * R;
                if (p < ENT * 8) *reinterpret_cast<uint4*>(smem + (size_t)(p >> 3) * 128 + (p & 7) * 16) = make_uint4(lv, lv, lv, lv);
            }(), ...);
        }(std::make_integer_sequence<int, LUT_R>{});
        asm volatile("s_waitcnt vmcnt(%0)" : : "n"((XA - 1) * LPW) : "memory");   // batch 0 has landed
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        for (int t = 0; t < nsteps; ++t) {
            if constexpr (XA == 2 || !ONE_BARRIER) __builtin_amdgcn_s_barrier();      // (A) XA = 2: stage t-1 is free
            if constexpr (!(dbg & 1) && XA == 2) lbatch(t + 2);
            asm volatile("s_waitcnt vmcnt(%0)" : : "n"(LPW) : "memory");   // batch t+1 has landed (one younger batch may be on its way)
            __builtin_amdgcn_s_barrier();                          // (B) stage t+1 is complete; stage t has been read by every wave
            if constexpr (!(dbg & 1) && XA == 3) lbatch(t + 3);    // ... and takes batch t+3
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;                                                    // the epilogue's barriers count the live waves only
    }
    // ---- scales of the workgroup's whole K range, once, ONE image per column group (its KP waves walk the same groups, interleaved):
    // eight 8-group blocks from the block that holds the range's first group, 8 / KP of them fetched by each wave of the group.
    // Request r of wave part kh, lane L: block (8 / KP) kh + 2 r + L / 32 of column (unit L % U, field (L & 31) / U); image
    // [block][column] x 16 B, lane-linear; complete at the prologue's barrier ----
    const int g0e = (kwg >> a.lg) & ~7;
    const uint32_t sc_base = (uint32_t)SC_BASE + (uint32_t)wg * SK_SCALE_GROUP;
    constexpr int BPW = 8 / KP;                                    // scale blocks per wave
    {
        const int cl = lane & 31;
        const uint32_t s_v = (uint32_t)(((size_t)(unit_col0<BITS, TILEP>(unit0 + (cl % U)) + (cl / U) * TILEP) * a.G + g0e + (kh * BPW + (lane >> 5)) * 8) * 2);
#pragma unroll
        for (int r = 0; r < BPW / 2; ++r) dma16_buf(s_v, s_srd, (uint32_t)r * 32u, sc_base + (uint32_t)(kh * BPW + 2 * r) * 512u);
    }
    issue_w(std::integral_constant<int, 0>{}, 0);
    issue_w(std::integral_constant<int, 1>{}, 1);
    if constexpr (WA == 3) issue_w(std::integral_constant<int, 2>{}, 2);
    // the pair table (entry e: 32 copies of its word at [128 e, 128 e + 128)) is written while the scale blocks and the
    // first weight pieces travel
    [&]<int... R>(std::integer_sequence<int, R...>) {
        ([&] {
            const uint32_t lv = lut_word_after<LUT_R - 1 - R + BPW / 2 + WA>(lutw[R]);
            const int p = tid + NTHR * R;
            if (p < ENT * 8) *reinterpret_cast<uint4*>(smem + (size_t)(p >> 3) * 128 + (p & 7) * 16) = make_uint4(lv, lv, lv, lv);
        }(), ...);
    }(std::make_integer_sequence<int, LUT_R>{});
    const uint32_t lane_off = (uint32_t)(lane & 31) * 4u;
    // fragment of row tile R, half step h, stage slot: x_grp + slot * SK_STAGE + R * 2048 + piece (r16 >> 3) * 1024 + row
    // (r16 & 7) * 128 + position ((4 h + q4) ^ swz) * 16; the lane part lives in two base registers, the rest is immediate
    const uint32_t frag_b0 = x_grp + (uint32_t)((r16 >> 3) * 1024 + (r16 & 7) * 128 + ((q4 ^ sk_swz(r16 & 7, r16 >> 3)) * 16));
    const uint32_t frag_b1 = x_grp + (uint32_t)((r16 >> 3) * 1024 + (r16 & 7) * 128 + (((4 + q4) ^ sk_swz(r16 & 7, r16 >> 3)) * 16));
    const uint32_t sc_lane = sc_base + (uint32_t)(fsel * U + u8) * 16u;
    const uint32_t shift0 = (uint32_t)(fsel * FB);

    f32x4_t acc[RT][NT2];
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int t = 0; t < NT2; ++t) acc[r][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // Operands of a half step live in register set h (half step 0 of every K step: set 0, half step 1: set 1): while set h
    // is multiplied, the LDS reads of the NEXT half step fill set h ^ 1 - all of them issued behind the first RT / 2 row
    // tiles' MFMAs, i.e. at least half a half step before they are needed (the first version reused ONE set in place: the
    // last fragment / lookup of a half step was requested by its last row tile and the wave paid the LDS latency at every
    // half-step boundary - 36 us instead of ~26 at K = 4096, profiles/r04/splitk_lab_run1*.jsonl).
    uint32_t v[2][8];                                              // hidden lookups: [set][tile * 4 + word]
    u32x4_t af[2][RT];                                             // activation fragments: [set][row tile]
    uint32_t scn[2][NT2];                                          // scales: [set][column tile]

    auto scales = [&](auto set_tag, int t, int h) {
        constexpr int set = decltype(set_tag)::value;
        const int rel = ((kbeg + t * KSTEP + h * 32) >> a.lg) - g0e;
        const uint32_t sb = sc_lane + (uint32_t)(rel >> 3) * 512u + (uint32_t)(rel & 7) * 2u;
        uint32_t& d0 = scn[set][0];
        uint32_t& d1 = scn[set][1];
        asm volatile("ds_read_u16 %0, %1" : "=v"(d0) : "v"(sb) : "memory");
        asm volatile("ds_read_u16 %0, %1 offset:256" : "=v"(d1) : "v"(sb) : "memory");     // field + FPT = 16 image columns on
    };
    auto lookup = [&](auto set_tag, const u32x4_t& qw, auto n_tag) {
        constexpr int set = decltype(set_tag)::value;
        constexpr int n = decltype(n_tag)::value;                  // tile n / 4, word n % 4
        const uint32_t idx = __builtin_amdgcn_ubfe(qw[n & 3], shift0 + (uint32_t)(FB * FPT * (n >> 2)), (uint32_t)FB);
        if constexpr (dbg & 8) v[set][n] = idx; else v[set][n] = lds_lookup32((idx << 7) | lane_off);
    };
    auto frag = [&](auto set_tag, auto slot_tag, auto h_tag, auto r_tag) {
        constexpr int set = decltype(set_tag)::value;
        constexpr int R = decltype(r_tag)::value;
        constexpr int h = decltype(h_tag)::value;
        constexpr int off = decltype(slot_tag)::value * SK_STAGE + R * 2048;
        static_assert(off < 65536, "ds_read_b128 immediate offset");
        u32x4_t& dst = af[set][R];
        const uint32_t addr = h == 0 ? frag_b0 : frag_b1;          // (named outside the asm: a generic lambda captures no variable it only meets as an asm operand)
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off) : "memory");
    };
    auto wait_lds = [&](auto set_tag) {                             // every LDS read issued so far has returned; names set `set`
        constexpr int set = decltype(set_tag)::value;
        uint32_t (&vv)[8] = v[set];                                // (references: a generic lambda captures no variable it only meets as an asm operand)
        u32x4_t (&aa)[RT] = af[set];
        uint32_t (&ss)[NT2] = scn[set];
        if constexpr (RT == 8) {
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(vv[0]), "+v"(vv[1]), "+v"(vv[2]), "+v"(vv[3]), "+v"(vv[4]), "+v"(vv[5]), "+v"(vv[6]), "+v"(vv[7]),
                           "+v"(aa[0]), "+v"(aa[1]), "+v"(aa[2]), "+v"(aa[3]), "+v"(aa[4 % RT]), "+v"(aa[5 % RT]), "+v"(aa[6 % RT]), "+v"(aa[7 % RT]),
                           "+v"(ss[0]), "+v"(ss[1])
                         : : "memory");
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(vv[0]), "+v"(vv[1]), "+v"(vv[2]), "+v"(vv[3]), "+v"(vv[4]), "+v"(vv[5]), "+v"(vv[6]), "+v"(vv[7]),
                           "+v"(aa[0]), "+v"(aa[1]), "+v"(aa[2]), "+v"(aa[3]), "+v"(ss[0]), "+v"(ss[1])
                         : : "memory");
        }
    };

    auto half = [&](auto slot_tag, auto h_tag, int t) {
        constexpr int slot = decltype(slot_tag)::value;
        constexpr int h = decltype(h_tag)::value;
        constexpr int nslot = h ? (slot + 1) % BLK_STAGES : slot;
        constexpr int nh = h ^ 1;
        using cur_t = std::integral_constant<int, h>;
        using nxt_t = std::integral_constant<int, nh>;
        wait_lds(cur_t{});
        // (B) the piece of step t+1 has landed once at most the younger pieces are outstanding
        if constexpr (h == 1) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(w[nslot]) : "n"((WA - 1) * NW) : "memory");
        // Round 5: everything of this half step that touches only the wave's own registers and its private scale image - the scale
        // multiplies of the weights, the next half step's scale reads and word shuffle - is done BEFORE the workgroup meets: it overlaps
        // the tail of the previous half step's MFMAs and the wait for the last wave, and after the barrier every wave starts with MFMAs
        // (64-row tiles: M = 256 x 4096^2 19.45 -> 19.25 us, M = 64 x 8192^2 20.05 -> 19.74; 128-row tiles: equal - profiles/r05/call34_*.log).
        u32x4_t bf[NT2];
#pragma unroll
        for (int c = 0; c < NT2; ++c) {
            const uint32_t sj = scn[h][c];
            const uint32_t vin[4] = {v[h][c * 4], v[h][c * 4 + 1], v[h][c * 4 + 2], v[h][c * 4 + 3]};
            uint32_t o[4];
            NT::mul_scale4(vin, sj, o);
            bf[c] = u32x4_t{o[0], o[1], o[2], o[3]};
        }
        scales(nxt_t{}, t + h, nh);
        const u32x4_t qw = half_words(w[nslot], nxt_t{});
        asm volatile("" : "+v"(bf[0]), "+v"(bf[1]) : : "memory");     // (keeps hipcc from sinking the multiplies below the barrier)
        // (A) [h = 0]; (B) [h = 1] stage t+1 is complete (and stage t free).  At request distance 3 nobody needs (A): the stage half step 0
        // reads next was complete at the previous (B), and the loaders refill a stage behind (B) only
        if constexpr (!(dbg & 32) && !(h == 0 && XA == 3 && ONE_BARRIER)) __builtin_amdgcn_s_barrier();
        auto row = [&](auto r_tag) {
            constexpr int R = decltype(r_tag)::value;
#pragma unroll
            for (int c = 0; c < NT2; ++c) {
                if constexpr (dbg & 4) acc[R][c][0] += __builtin_bit_cast(float, bf[c][0] ^ af[h][R][0]);
                else acc[R][c] = Mfma<T>::run(bf[c], af[h][R], acc[R][c]);
            }
            // the next half step's operands: two fragments and 8 / HF lookups behind each of the first HF row tiles
            if constexpr (R < HF) {
                if constexpr (!(dbg & 16)) {
                    frag(nxt_t{}, std::integral_constant<int, nslot>{}, nxt_t{}, std::integral_constant<int, 2 * R>{});
                    frag(nxt_t{}, std::integral_constant<int, nslot>{}, nxt_t{}, std::integral_constant<int, 2 * R + 1>{});
                }
                [&]<int... L>(std::integer_sequence<int, L...>) {
                    (lookup(nxt_t{}, qw, std::integral_constant<int, R * LPR + L>{}), ...);
                }(std::make_integer_sequence<int, LPR>{});
            }
            // the weight piece of a later step behind the last row tile of half step 0: three ahead into the slot of step t (its second
            // half was shuffled out in front of barrier (A)), or two ahead into the slot step t - 1 left
            if constexpr (h == 0 && R == RT - 1 && NW > 0) {
                if constexpr (WA == 3) issue_w(std::integral_constant<int, slot>{}, t + 3);
                else issue_w(std::integral_constant<int, (slot + 2) % BLK_STAGES>{}, t + 2);
            }
        };
        [&]<int... R>(std::integer_sequence<int, R...>) {
            (row(std::integral_constant<int, R>{}), ...);
        }(std::make_integer_sequence<int, RT>{});
    };

    // scales, the first weight piece and the pair table before anyone reads them
    asm volatile("s_waitcnt vmcnt(%1)" : "+v"(w[0]) : "n"(WA - 1) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    FLUTE_SKSTAMP(1);
    {
        using set0 = std::integral_constant<int, 0>;
        scales(set0{}, 0, 0);
        const u32x4_t qw = half_words(w[0], set0{});
        [&]<int... R>(std::integer_sequence<int, R...>) {
            (frag(set0{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, R>{}), ...);
        }(std::make_integer_sequence<int, RT>{});
        [&]<int... L>(std::integer_sequence<int, L...>) {
            (lookup(set0{}, qw, std::integral_constant<int, L>{}), ...);
        }(std::make_integer_sequence<int, 8>{});
    }
    // static priority for the second-dispatched half of the compute waves (MI355X_MICROARCH.md, "two waves per SIMD": the younger wave of
    // a SIMD loses the VALU arbitration on every segment): 128-row tiles M = 256 on 4096 x 11008 32.2 -> 31.8 us, 64-row tiles equal
    // (profiles/r06/call21_wave_priority.log; the loader waves above the compute waves: nothing)
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);
    auto step = [&](auto slot_tag, int t) {
        half(slot_tag, std::integral_constant<int, 0>{}, t);
        half(slot_tag, std::integral_constant<int, 1>{}, t);
    };
    for (int t0 = 0;; t0 += BLK_STAGES) {
        step(std::integral_constant<int, 0>{}, t0);
        if (t0 + 1 >= nsteps) break;
        step(std::integral_constant<int, 1>{}, t0 + 1);
        if (t0 + 2 >= nsteps) break;
        step(std::integral_constant<int, 2>{}, t0 + 2);
        if (t0 + 3 >= nsteps) break;
    }
    wait_lds(std::integral_constant<int, 0>{});                    // the prefetch past the end (set 0: the last half step is a half step 1)
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]) : : "memory");

    FLUTE_SKSTAMP(2);
    // ---- epilogue 1: the K parts exchange row tiles through LDS: part p keeps row tiles HR p .. HR p + HR - 1 and hands the others
    // to the waves of its column group that keep them; own[i][t] = row tile HR kh + i over the workgroup's whole K range, summed in
    // ascending part order whoever holds it ----
    f32x4_t own[HR][NT2];
    {
        __syncthreads();                                           // every wave is done with the stages
        float4* xb = reinterpret_cast<float4*>(smem + X_BASE);     // [keeping wave][source part, own skipped][HR][2 column tiles][64 lanes] x 16 B: 8 (KP - 1) HR x 2 KB
        static_assert(8 * (KP - 1) * HR * NT2 * 1024 <= KP * BLK_STAGES * SK_STAGE, "the exchange lives in the stages");
        auto slot = [&](int keeper_part, int src_part, int i, int t) {
            const int keeper = wg + NWN * keeper_part;
            const int sp = src_part - (src_part > keeper_part ? 1 : 0);
            return (((keeper * (KP - 1) + sp) * HR + i) * NT2 + t) * 64 + lane;
        };
        // Every register index below is a constant (row tile r of the hand-over, the candidates of the keeper's own term): kh only
        // enters addresses and wave-uniform selects.  (A branch chain over kh with per-part bodies was tail-merged by hipcc into ONE
        // body with run-time accumulator indices: 96 B of scratch per lane.)
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            if (r / HR != kh) {                                    // wave-uniform: row tile r is kept by part r / HR
#pragma unroll
                for (int t = 0; t < NT2; ++t) {
                    const f32x4_t s4 = acc[r][t];
                    xb[slot(r / HR, kh, r % HR, t)] = make_float4(s4[0], s4[1], s4[2], s4[3]);
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < HR; ++i)
#pragma unroll
            for (int t = 0; t < NT2; ++t) {
                f32x4_t mine = acc[i][t];                          // row tile HR kh + i of this wave's own K part
#pragma unroll
                for (int P = 1; P < KP; ++P)
                    if (kh == P) mine = acc[P * HR + i][t];
                f32x4_t sum = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int q = 0; q < KP; ++q) {
                    f32x4_t term = mine;
                    if (q != kh) { const float4 g4 = xb[slot(kh, q, i, t)]; term = f32x4_t{g4.x, g4.y, g4.z, g4.w}; }
                    sum = (q == 0) ? term : sum + term;
                }
                own[i][t] = sum;
            }
    }

    FLUTE_SKSTAMP(3);
    // ---- epilogue 2: accumulator register j of lane (r16, q4) = weight row 4 q4 + j of the column tile = unit 2 (q4 % (U / 2)) + (j & 1),
    // field 2 (q4 / (U / 2)) + (j >> 1) + FPT t: two pairs of columns TILEP apart; the lane's output row is r16.  v_permlane16_swap of
    // (j0, j2) and (j1, j3) between the lane rows q4, q4 ^ 1 leaves four consecutive units (= columns) of one field in every lane:
    // even q4 the field 2 (q4 / (U / 2)), odd q4 that + 1 ----
    const int c_unit = unit0 + ((q4 & ~1) % (U / 2)) * 2;
    const int c_field = (q4 / (U / 2)) * 2 + (q4 & 1);
    uint32_t col[NT2];
#pragma unroll
    for (int t = 0; t < NT2; ++t) col[t] = (uint32_t)(unit_col0<BITS, TILEP>(c_unit) + (c_field + FPT * t) * TILEP);
    const int row_base = m0 + kh * (HR * 16) + r16;                // + 16 i
    auto store_d = [&](int i, int t, const f32x4_t o4) {
        // (scalars first: __builtin_bit_cast applied to o4[j] directly took element 0 four times - hipcc 7.2)
        const float e0 = o4[0], e1 = o4[1], e2 = o4[2], e3 = o4[3];
        uint32_t f[4] = {__builtin_bit_cast(uint32_t, e0), __builtin_bit_cast(uint32_t, e1), __builtin_bit_cast(uint32_t, e2), __builtin_bit_cast(uint32_t, e3)};
        // every lane of the wave is here (the row predicate follows); wait states on both sides of the lane swaps as fwht.h
        asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %2\n\tv_permlane16_swap_b32 %1, %3\n\ts_nop 1" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]));
        const int row = row_base + 16 * i;
        if (row < a.M) {
            uint2 o;
            o.x = (uint32_t)NT::from_float(__builtin_bit_cast(float, f[0])) | ((uint32_t)NT::from_float(__builtin_bit_cast(float, f[1])) << 16);
            o.y = (uint32_t)NT::from_float(__builtin_bit_cast(float, f[2])) | ((uint32_t)NT::from_float(__builtin_bit_cast(float, f[3])) << 16);
            // (write-through - sc1 - output stores, so that the bytes leave the L2 when stored and not at the end-of-kernel write-back: equal on
            // 64-row tiles, 128-row tiles 31.5 -> 33.0 us; profiles/r06/call25_write_through_output_stores_dropped.log)
            *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(a.D) + (size_t)row * a.N + col[t]) = o;
        }
    };
    if (a.splitk == 1 || (dbg & 128)) {                             // (ablation 128: every slice stores its partial as the result - the seam's price)
#pragma unroll
        for (int i = 0; i < HR; ++i)
#pragma unroll
            for (int t = 0; t < NT2; ++t) store_d(i, t, own[i][t]);
        FLUTE_SKSTAMP_FLUSH();
        return;
    }

    // Slabs in FRAGMENT order: [slice][tile][wave][row tile i][column tile t][lane] x 16 B - every store / load instruction of
    // a wave moves one contiguous KB, and the same lane of the same wave of another slice finds its counterpart at the same
    // place (the reference's BlockStripedReduce does the same, tile_scheduler_utils.hpp:80-83).  The first version kept the
    // slabs as [M][N] images: 32-B runs per row, write-through one fabric write each - 10 us of seam (profiles/r04).
    const uint32_t ntiles = gridDim.x / (uint32_t)a.splitk;
    constexpr uint32_t TILE_SLAB = 8u * HR * NT2 * 1024u;            // 8 waves x HR x 2 fragments of 1 KB: 64 KB (RT = 8) / 32 KB
    const __amdgpu_buffer_rsrc_t slab = xwg_rsrc(a.partial, (uint32_t)a.splitk * ntiles * TILE_SLAB);
    const uint32_t slab_lane = (uint32_t)tile * TILE_SLAB + (uint32_t)wave * (HR * NT2 * 1024u) + (uint32_t)lane * 16u;
    auto slab_off = [&](int slice, int i, int t) {
        return (uint32_t)slice * (ntiles * TILE_SLAB) + slab_lane + (uint32_t)(i * NT2 + t) * 1024u;
    };
    xwg_word* st = xwg_state(a.state + 2 * tile);
    const uint32_t bcast = 0;                                      // LDS dword 0 (the pair table is dead)
    const int nsl = a.splitk;

    // E form for NSH = 2, 4 shares (share q = row tiles q, q + NSH, ... of every wave), PER = HR / NSH row tiles per share
    auto e_form = [&]<int NSH>(std::integral_constant<int, NSH>) {
        constexpr int PER = HR / NSH;
        const int me = split;
        f32x4_t mine[PER][NT2];
#pragma unroll
        for (int i = 0; i < HR; ++i) {
            if (i % NSH == me) {                                   // wave-uniform
#pragma unroll
                for (int t = 0; t < NT2; ++t) mine[i / NSH][t] = own[i][t];
            } else {
#pragma unroll
                for (int t = 0; t < NT2; ++t) xwg_store(own[i][t], slab, slab_off(me, i, t));
            }
        }
        // share `sh`: first + the other slices' partials in ascending slice order -> D
        auto combine = [&](int sh, const f32x4_t (&first)[PER][NT2]) {
            f32x4_t ld[PER][NT2][NSH - 1];
#pragma unroll
            for (int jj = 0; jj < PER; ++jj)
#pragma unroll
                for (int t = 0; t < NT2; ++t)
#pragma unroll
                    for (int o = 0; o < NSH - 1; ++o) {
                        const int s2 = o + (o >= sh ? 1 : 0);      // the o-th slice other than the owner
                        ld[jj][t][o] = xwg_load(slab, slab_off(s2, sh + jj * NSH, t));
                    }
#pragma unroll
            for (int jj = 0; jj < PER; ++jj)
#pragma unroll
                for (int t = 0; t < NT2; ++t) {
                    f32x4_t s = first[jj][t];
#pragma unroll
                    for (int o = 0; o < NSH - 1; ++o) s += ld[jj][t][o];
                    FLUTE_SKSTAMP(7);
                    store_d(sh + jj * NSH, t, s);
                }
        };
        FLUTE_SKSTAMP(4);
        const uint32_t before = xwg_arrive(st, bcast, tid);
        FLUTE_SKSTAMP(5);
#ifdef FLUTE_STAMPS
        stamp[9] = before;
#endif
        if (before == (uint32_t)(NSH - 1)) {
            FLUTE_SKSTAMP(6);
            combine(me, mine);
            const uint32_t ab = xwg_sweep(st, NSH, me, bcast, tid);
            for (int q = 0; q < NSH; ++q) {
                if (!((ab >> q) & 1u)) continue;                   // abandoned by its owner: every slice's partial of it is in place
                f32x4_t first[PER][NT2];
#pragma unroll
                for (int jj = 0; jj < PER; ++jj)
#pragma unroll
                    for (int t = 0; t < NT2; ++t) first[jj][t] = xwg_load(slab, slab_off(q, q + jj * NSH, t));
                combine(q, first);
            }
            xwg_reset(st, tid);
        } else if (xwg_wait_all(st, NSH, bcast, tid)) {
            FLUTE_SKSTAMP(6);
            xwg_claim(st, me, tid);
            combine(me, mine);
        } else {
#pragma unroll
            for (int jj = 0; jj < PER; ++jj)
#pragma unroll
                for (int t = 0; t < NT2; ++t) xwg_store(mine[jj][t], slab, slab_off(me, me + jj * NSH, t));
            xwg_abandon(st, me, tid);
        }
    };

    bool met = false;                                              // the E form needs whole row tiles per share
    if constexpr (HR % 4 == 0) { if (nsl == 4) { e_form(std::integral_constant<int, 4>{}); met = true; } }
    if constexpr (HR % 2 == 0) { if (nsl == 2) { e_form(std::integral_constant<int, 2>{}); met = true; } }
    if (!met) {
        // L form: every slice publishes its whole partial; the last arriver sums ALL slices in ascending order (its own
        // from the slab as well: one order whoever is last)
#pragma unroll
        for (int i = 0; i < HR; ++i)
#pragma unroll
            for (int t = 0; t < NT2; ++t) xwg_store(own[i][t], slab, slab_off(split, i, t));
        FLUTE_SKSTAMP(4);
        const uint32_t before = xwg_arrive(st, bcast, tid);
        FLUTE_SKSTAMP(5);
#ifdef FLUTE_STAMPS
        stamp[9] = before;
#endif
        if (before == (uint32_t)(nsl - 1)) {
            f32x4_t s[HR][NT2];
#pragma unroll
            for (int i = 0; i < HR; ++i)
#pragma unroll
                for (int t = 0; t < NT2; ++t) s[i][t] = xwg_load(slab, slab_off(0, i, t));
            for (int s2 = 1; s2 < nsl; ++s2) {
                f32x4_t ld[HR][NT2];
#pragma unroll
                for (int i = 0; i < HR; ++i)
#pragma unroll
                    for (int t = 0; t < NT2; ++t) ld[i][t] = xwg_load(slab, slab_off(s2, i, t));
#pragma unroll
                for (int i = 0; i < HR; ++i)
#pragma unroll
                    for (int t = 0; t < NT2; ++t) s[i][t] += ld[i][t];
            }
#pragma unroll
            for (int i = 0; i < HR; ++i)
#pragma unroll
                for (int t = 0; t < NT2; ++t) store_d(i, t, s[i][t]);
            xwg_reset(st, tid);
        }
    }
    FLUTE_SKSTAMP_FLUSH();
#undef FLUTE_SKSTAMP
#undef FLUTE_SKSTAMP_FLUSH
}

}  // namespace flute_amd
