// Split-K block kernel (round 4): 128 (or 64) x 128 output tiles, K split over workgroups, partial tiles combined INSIDE the launch.
//
// The chip-filling schedule for 128 <= M <= ~1024 that rounds 2 and 3 lacked: the reference's Stream-K hands every CTA an equal
// share of tiles_M x tiles_N x tiles_K for any M (flute/csrc/tile_scheduler_utils.hpp:460-481, fix-up :58-211, main loop
// qgemm_kernel.hpp:617-712).  Here the output is cut into tiles of RT x 16 rows x 128 columns, each cut `splitk` ways in K -
// one workgroup per (tile, slice); the partial tiles of an output tile meet through the workspace with write-through
// stores and one arrival word (xwg.h): no second launch, no release fence.  M = 256 x 4096 x 11008: 172 tiles x 1 slice;
// M = 1024 x 4096^2: 256 tiles; M = 256 x 4096^2: 128 tiles of 64 rows x 2 slices.
//
// Workgroup = 8 compute waves = 4 column groups (32 columns = two MFMA column tiles each, as qgemm_block2.h) x 2 K halves
// (+ LDW = 4 loader waves): the waves of K half g run the block2 pipeline on the contiguous half [g kps / 2, (g + 1) kps / 2)
// of the workgroup's K range with their own three 64-k activation stages; every weight of the workgroup's range is
// dequantised exactly once (8 lookups + 8 multiplies per RT x 2 MFMAs of a 32-k half step), a wave holds RT row tiles x 2
// column tiles of accumulators.  What differs from qgemm_block2.h (each step measured: DESIGN.md 3.2d, profiles/r04/):
//   * activation pieces are 8 rows x 128 B - WHOLE cache lines (a request is priced per line it touches); fragment
//     swizzle sk_swz (= qgemm_tile.h's swz_x);
//   * the wave's scales for its whole K half are fetched once, by the prologue (<= 32 groups x 32 columns = 2 KB per
//     wave): no scale request - and no sink request - per step;
//   * ONE whole-line weight request per compute wave and step (the half step a lane lacks comes from its partner lane
//     by DPP row_ror:8);
//   * the operand registers are double-buffered: the next half step's LDS reads are issued behind the first RT / 2 row
//     tiles' MFMAs;
//   * the loader waves issue every activation piece (RT per loader and step, both K halves); without them (LDW = 0) a
//     compute wave issues RT / 2 pieces per step between the MFMAs of half step 0;
//   * XCD-aware tile order when K is not split: the row-tile pairs that share a column tile's weights run on one XCD.
// Epilogue: (1) the two K halves exchange half of their row tiles through LDS (every wave ends with RT / 2 row tiles x 2
// column tiles of the workgroup's K range), (2) splitk > 1: the E form of xwg.h when splitk is 2 (or 4 with RT = 8: slice s
// owns row tiles s, s + nsh, ... of every wave), the L form otherwise; sums are taken in a fixed order (owner first, then
// the other slices ascending; L: all slices ascending), so the result does not depend on arrival order.
// Arithmetic contract as qgemm_block2.h: w^ = round_T(lut * s) (packbits_utils.hpp:139), fp32 accumulation, one rounding.
// Host contract (api.hip, plan_splitk): 2 or 4 bits, G % 8 == 0, K % k_per_split == 0, k_per_split % (2 * max(64, g)) == 0,
// at most 32 scale groups (+ alignment slack: four 8-group blocks) per K half, splitk * tiles * RT * 8 KB of slabs < 2^31.
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
    float* partial;         // [splitk][tile] x 64 KB fp32 partial tiles in fragment order (write-through), splitk > 1
    uint32_t* state;        // two words per output tile, zero before and after the launch (xwg.h)
    int M, N, K, G, lg;
    int tiles_m;            // 128-row tiles (fastest in the block order: the row tiles of a column tile are neighbours)
    int splitk, k_per_split;
    // XCD-aware tile order (splitk == 1, tiles_m = 2 P with P a power of two): pair_lg = log2(P), pair_c8 = (P x column tiles)
    // rounded down to a multiple of 8; pair_lg < 0: tiles in their natural order
    int pair_lg, pair_c8;
};

// RT row tiles per workgroup: 8 (128-row tiles) or - round 4, for outputs whose 128-row tiles leave most of the chip idle
// (M = 256 on 4096 x 4096: 64 tiles) - 4 (64-row tiles: twice the tiles, half the slab per slice, every weight
// dequantised by twice as many workgroups)
__host__ __device__ constexpr int splitk_stage_bytes(int rt) { return rt * 2 * 1024; }     // RT x 16 rows x 64 k: 1-KB pieces
constexpr int SK_SCALE_WAVE = 2048;                                // four 8-group blocks x 32 columns x 16 B
// position swizzle of an 8-row x 8-chunk activation piece (rows 8 rh .. 8 rh + 7 of a 16-row tile): the 16 lanes of every
// ds_read_b128 lane group hit 16 different 16-B slots of the 256-B bank row (as qgemm_tile.h's swz_x; checked for the
// lane groups of MI355X_MICROARCH.md's LDS table by tests/test_host.py)
__host__ __device__ constexpr int sk_swz(int row8, int rh) { return (row8 >> 1) | (rh << 2); }
__host__ __device__ constexpr int splitk_lds_bytes(int bits, int rt = 8) {
    return (128 << (2 * bits)) + 2 * BLK_STAGES * splitk_stage_bytes(rt) + 8 * SK_SCALE_WAVE;
}

// LDW = 4: four LOADER waves beside the eight compute waves (block of 768 threads).  A loader issues the activation
// pieces of both K halves (8 per step) and nothing else; the compute waves are left with one weight request per step -
// their in-order instruction streams no longer stall in the addresser's queue between MFMAs.  Same barriers, same stages.
// A hidden load's value, handed over in a NEW register once at most N younger loads are outstanding.  (The wait and the
// move are one asm statement: with the usual "+v" wait hipcc was seen copying the destination into the first register
// of the quad the table store wants BEFORE the wait - the audit caught it in the 768-thread variant.)
template <int N> __device__ __forceinline__ uint32_t lut_word_after(const uint32_t& hidden) {
    uint32_t v;
    asm volatile("s_waitcnt vmcnt(%2)\n\tv_mov_b32 %0, %1" : "=&v"(v) : "v"(hidden), "n"(N) : "memory");
    return v;
}

template <typename T, int TILEP, int BITS = 4, int LDW = 0, int RT = 8>
__global__ __launch_bounds__(512 + 64 * LDW) void qgemm_splitk_kernel(const SplitKArgs args) {
    using NT = Num<T>;
    static_assert(BITS == 4 || BITS == 2, "3-bit layers: qgemm_block3.h / qgemm_tile.h");
    constexpr int J = 16 / BITS;
    constexpr int U = 32 / J;                                      // units per wave (32 columns)
    constexpr int FPT = 16 / U;                                    // fields per column tile and unit
    constexpr int FB = 2 * BITS;
    static_assert(RT == 8 || RT == 4, "row tiles per workgroup");
    constexpr int NT2 = 2, NWN = 4;
    constexpr int HR = RT / 2;                                     // row tiles a wave keeps after the K halves' exchange
    constexpr int SK_STAGE = splitk_stage_bytes(RT);
    constexpr int LPR = 8 / HR;                                    // next-half lookups issued behind each of the first HR row tiles
    static_assert(LDW == 0 || LDW == 4, "loader waves");
    constexpr int NTHR = 512 + 64 * LDW;
    constexpr int PPW = LDW ? 0 : RT * 2 / NWN;                    // activation pieces per COMPUTE wave and step (4; none with loaders)
    constexpr int LPW = LDW ? 2 * RT * 2 / LDW : 0;                // ... per LOADER wave and step (both K halves: 32 / 4)
#ifdef FLUTE_SK_ABLATE   // development builds (tools/splitk_ablate.sh): 1 no activation requests in the loop, 2 no weight requests,
    constexpr int dbg = FLUTE_SK_ABLATE;                           // 4 no MFMA, 8 no table lookups, 16 no fragment reads, 32 no barriers, 128 no seam
#else
    constexpr int dbg = 0;
#endif
    // A step's batch: 4 activation pieces + 1 weight piece.  (Measured and dropped, profiles/r04/splitk_lab_run4_line_touch_
    // prefetch_dropped.jsonl: one extra 4-B LDS-DMA per wave and step whose 64 lanes touch the 40 cache lines the wave will
    // want eight steps later - the requests are priced per LINE, so it doubled the addresser's work: 24.5 -> 27.8 us.)
    constexpr int NWQ = 1;                                         // weight requests per wave and step (one whole-line piece)
    constexpr int BATCH = ((dbg & 1) ? 0 : PPW) + ((dbg & 2) ? 0 : NWQ);
    static_assert(BATCH <= RT, "one request per row tile of half step 0");
    constexpr int LUT_BYTES = (1 << (2 * BITS)) * 128;
    constexpr int X_BASE = LUT_BYTES;
    constexpr int SC_BASE = X_BASE + 2 * BLK_STAGES * SK_STAGE;

    SplitKArgs a = args;
    {
#define FLUTE_OPAQUE(x) asm volatile("" : "+s"(x))
        FLUTE_OPAQUE(a.A); FLUTE_OPAQUE(a.Q); FLUTE_OPAQUE(a.D); FLUTE_OPAQUE(a.S); FLUTE_OPAQUE(a.QM2);
        FLUTE_OPAQUE(a.partial); FLUTE_OPAQUE(a.state); FLUTE_OPAQUE(a.M); FLUTE_OPAQUE(a.N); FLUTE_OPAQUE(a.K);
        FLUTE_OPAQUE(a.G); FLUTE_OPAQUE(a.lg); FLUTE_OPAQUE(a.tiles_m); FLUTE_OPAQUE(a.splitk); FLUTE_OPAQUE(a.k_per_split);
        FLUTE_OPAQUE(a.pair_lg); FLUTE_OPAQUE(a.pair_c8);
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
    const int u8 = r16 % U;
    const int fsel = r16 / U;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wg = wave & 3;                                       // column group
    const int kh = wave >> 2;                                      // K half

    int tile = blockIdx.x, split = 0;
    if (a.splitk > 1) { split = tile % a.splitk; tile /= a.splitk; }
    // Block b runs on XCD b % 8 (observed; speed only).  Natural order (row tiles fastest) puts the two row tiles of a column
    // tile on neighbouring XCDs: each fetches the column tile's weights for itself (FETCH 1.9x the algorithmic bytes at
    // M = 256, profiles/r04_rocprof).  Here a PAIR of row tiles (2 p, 2 p + 1) x column tile = "pair column" c goes to XCD
    // c % 8 as that XCD's consecutive blocks: the weights are fetched once per pair, an XCD reads two row tiles of X.
    int tm_idx, tn_idx;
    if (a.pair_lg >= 0) {
        int c, e;
        if (tile < 2 * a.pair_c8) { const int i = tile >> 3; c = (i >> 1) * 8 + (tile & 7); e = i & 1; }
        else { c = tile >> 1; e = tile & 1; }                      // the last, incomplete group of eight: natural order
        tn_idx = c >> a.pair_lg;
        tm_idx = 2 * (c & ((1 << a.pair_lg) - 1)) + e;
    } else {
        tm_idx = tile % a.tiles_m;
        tn_idx = tile / a.tiles_m;
    }
    const int m0 = tm_idx * (RT * 16);
    const int unit0 = (tn_idx * NWN + wg) * U;
    const int khalf = a.k_per_split >> 1;
    const int kbeg = split * a.k_per_split + kh * khalf;           // this wave's K range: [kbeg, kbeg + khalf)
    const int nsteps = khalf >> 6;
    const uint32_t row_bytes = (uint32_t)a.K * 2u;

    const srd_t x_srd = make_srd(a.A, (uint32_t)min((size_t)a.M * a.K * 2, (size_t)0xfffffff0u));
    const srd_t w_srd = make_srd(reinterpret_cast<const char*>(a.Q) + (size_t)unit0 * row_bytes, (uint32_t)U * row_bytes);
    const srd_t s_srd = make_srd(a.S, (uint32_t)min((size_t)a.N * a.G * 2, (size_t)0xfffffff0u));

    // ---- activation pieces: piece p = 2 rt + rh of a stage = rows 16 rt + 8 rh .. + 7, 128 B (the 64 k of the step) each;
    // lane L fetches the 16-B chunk (L & 7) ^ sk_swz(L >> 3, rh) of row L >> 3 and the DMA writes it lane-linearly, so LDS
    // position pos of row8 holds chunk pos ^ swz.  This wave's pieces: 4 wg .. 4 wg + 3 (row tiles 2 wg, 2 wg + 1).
    // Rows past M: their byte offset is past the descriptor's range and reads as zero (voffset is what is checked).
    const int row8 = lane >> 3;
    uint32_t x_vo[PPW ? PPW : 1];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int rt = (wg * PPW + i) >> 1, rh = (wg * PPW + i) & 1;
        x_vo[i] = (uint32_t)(((size_t)(m0 + rt * 16 + rh * 8 + row8) * a.K + (((lane & 7) ^ sk_swz(row8, rh)) * 8)) * 2);
    }
    const uint32_t x_grp = (uint32_t)X_BASE + (uint32_t)kh * (BLK_STAGES * SK_STAGE);      // this K half's three stages
    const uint32_t x_lds0 = x_grp + (uint32_t)(wg * PPW) * 1024u;
    // Weights: ONE request per wave and step - 8 unit rows x 128 B, whole cache lines: lane (r16, q4) fetches chunk
    // 4 (r16 >> 3) + q4 of unit r16 % U, i.e. the lanes of weight rows 0..7 hold half step 0's words and those of rows 8..15
    // half step 1's (the same unit's other field).  The half a lane lacks comes from lane r16 ^ 8 of its 16-lane row by DPP
    // (row_ror:8, four moves per half step).  (First version: two requests per step, each 16 half lines with lanes r16 and
    // r16 + 8 reading the same 16 B; a request is priced per line it touches.)
    const uint32_t w_voff = (uint32_t)u8 * row_bytes + (uint32_t)((r16 >> 3) * 4 + q4) * 16u;

    u32x4_t w[BLK_STAGES];
    // batch u = the hidden loads of this wave's K step u; batches past the end (issued two steps ahead, never consumed)
    // re-read the last step.  Every K offset is wave-uniform and travels in the scalar offset.
    auto issue_one = [&](auto slot_tag, auto i_tag, int u) {
        constexpr int slot = decltype(slot_tag)::value;
        constexpr int i = decltype(i_tag)::value;
        const uint32_t k0 = (uint32_t)(kbeg + min(u, nsteps - 1) * 64);
        if constexpr (i < PPW) dma16_buf(x_vo[i], x_srd, k0 * 2u, x_lds0 + (uint32_t)i * 1024u + (uint32_t)slot * SK_STAGE);
        else w[slot] = buf_load16(w_voff, w_srd, k0 * 2u);
    };
    // request j of a step's batch -> piece index (ablation builds drop the activation or the weight requests)
    auto issue_nth = [&](auto slot_tag, auto j_tag, int u) {
        constexpr int j = decltype(j_tag)::value;
        issue_one(slot_tag, std::integral_constant<int, (dbg & 1) ? PPW + j : j>{}, u);
    };
    auto issue_batch = [&](auto slot_tag, int u) {
        [&]<int... I>(std::integer_sequence<int, I...>) {
            (issue_one(slot_tag, std::integral_constant<int, I>{}, u), ...);
        }(std::make_integer_sequence<int, PPW + NWQ>{});
    };
    // the words of half step h out of a step's piece: own where (r16 >> 3) == h, else the partner lane's
    auto half_words = [&](const u32x4_t& own, auto h_tag) {
        constexpr int h = decltype(h_tag)::value;
        u32x4_t r;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            r[j] = (uint32_t)__builtin_amdgcn_update_dpp((int)own[j], (int)own[j], 0x128 /* row_ror:8 */, 0xf, h == 0 ? 0xc : 0x3, false);
        return r;
    };

    // ---- pair-table words first (oldest in the queue: their wait below leaves everything else in flight) ----
    constexpr int ENT = 1 << (2 * BITS);
    constexpr int LUT_R = (ENT * 8 + NTHR - 1) / NTHR;             // 16-B table pieces per thread
    const srd_t lut_srd = make_srd(a.QM2, (uint32_t)(4 * ENT));
    uint32_t lutw[LUT_R];
#pragma unroll
    for (int r = 0; r < LUT_R; ++r) lutw[r] = buf_load4((uint32_t)((tid + NTHR * r) >> 3) * 4u, lut_srd);   // past the table: reads 0, not written
    if constexpr (LDW > 0) {
        if (wave >= 8) {
            // ---- loader wave L: K half L / 2, pieces 8 (L % 2) .. + 7 of that half's stage, every step ----
            const int L = wave - 8, lkh = L >> 1, lp0 = (L & 1) * LPW;
            const int lkbeg = split * a.k_per_split + lkh * khalf;
            uint32_t lx[LPW];
#pragma unroll
            for (int i = 0; i < LPW; ++i) {
                const int rt = (lp0 + i) >> 1, rh = (lp0 + i) & 1;
                lx[i] = (uint32_t)(((size_t)(m0 + rt * 16 + rh * 8 + row8) * a.K + (((lane & 7) ^ sk_swz(row8, rh)) * 8)) * 2);
            }
            const uint32_t ldst = (uint32_t)X_BASE + (uint32_t)lkh * (BLK_STAGES * SK_STAGE) + (uint32_t)lp0 * 1024u;
            auto lbatch = [&](int u) {
                const uint32_t k0 = (uint32_t)(lkbeg + min(u, nsteps - 1) * 64);
                const uint32_t dst = ldst + (uint32_t)(u % BLK_STAGES) * SK_STAGE;
#pragma unroll
                for (int i = 0; i < LPW; ++i) dma16_buf(lx[i], x_srd, k0 * 2u, dst + (uint32_t)i * 1024u);
            };
            lbatch(0);
            lbatch(1);
            [&]<int... R>(std::integer_sequence<int, R...>) {
                ([&] {
                    const uint32_t lv = lut_word_after<LUT_R - 1 - R + 2 * LPW>(lutw[R]);
                    const int p = tid + NTHR * R;
                    if (p < ENT * 8) *reinterpret_cast<uint4*>(smem + (size_t)(p >> 3) * 128 + (p & 7) * 16) = make_uint4(lv, lv, lv, lv);
                }(), ...);
            }(std::make_integer_sequence<int, LUT_R>{});
            asm volatile("s_waitcnt vmcnt(%0)" : : "n"(LPW) : "memory");       // batch 0 has landed
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            for (int t = 0; t < nsteps; ++t) {
                __builtin_amdgcn_s_barrier();                      // (A) stage t-1 is free
                lbatch(t + 2);
                asm volatile("s_waitcnt vmcnt(%0)" : : "n"(LPW) : "memory");   // (B) batch t+1 has landed
                __builtin_amdgcn_s_barrier();
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            return;                                                // the epilogue's barriers count the live waves only
        }
    }
    // ---- scales of the whole K half, once: 8-group blocks from the block that holds the first group.  Request r, lane L:
    // block 2 r + L / 32 of column (unit L % U, field (L & 31) / U); image [block][column] x 16 B, lane-linear ----
    const int g0e = (kbeg >> a.lg) & ~7;
    const uint32_t sc_base = (uint32_t)SC_BASE + (uint32_t)wave * SK_SCALE_WAVE;
    {
        const int cl = lane & 31;
        const uint32_t s_v = (uint32_t)(((size_t)(unit_col0<BITS, TILEP>(unit0 + (cl % U)) + (cl / U) * TILEP) * a.G + g0e + (lane >> 5) * 8) * 2);
#pragma unroll
        for (int r = 0; r < 2; ++r) dma16_buf(s_v, s_srd, (uint32_t)r * 32u, sc_base + (uint32_t)r * 1024u);
    }
    issue_batch(std::integral_constant<int, 0>{}, 0);
    issue_batch(std::integral_constant<int, 1>{}, 1);
    // the pair table (entry e: 32 copies of its word at [128 e, 128 e + 128)) is written while the scale blocks and the
    // first two batches travel
    [&]<int... R>(std::integer_sequence<int, R...>) {
        ([&] {
            const uint32_t lv = lut_word_after<LUT_R - 1 - R + 2 + 2 * (PPW + NWQ)>(lutw[R]);
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
    // is multiplied, the LDS reads of the NEXT half step fill set h ^ 1 - all of them issued behind the first four row
    // tiles' MFMAs, i.e. at least half a half step before they are needed (the first version reused ONE set in place: the
    // last fragment / lookup of a half step was requested by its last row tile and the wave paid the LDS latency at every
    // half-step boundary - 36 us instead of ~26 at K = 4096, profiles/r04/splitk_lab_run1*.jsonl)
    uint32_t v[2][8];                                              // hidden lookups: [set][tile * 4 + word]
    u32x4_t af[2][RT];                                             // activation fragments: [set][row tile]
    uint32_t scn[2][NT2];                                          // scales: [set][column tile]

    auto scales = [&](auto set_tag, int t, int h) {
        constexpr int set = decltype(set_tag)::value;
        const int rel = ((kbeg + t * 64 + h * 32) >> a.lg) - g0e;
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
        // (B) batch t+1 has landed once at most batch t+2 is outstanding
        if constexpr (h == 1) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(w[nslot]) : "n"(BATCH) : "memory");
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
        // (A) [h = 0] stage t-1 is free: batch t+2 follows, spread over the rows; (B) [h = 1] stage t+1 is complete
        if constexpr (!(dbg & 32)) __builtin_amdgcn_s_barrier();
        auto row = [&](auto r_tag) {
            constexpr int R = decltype(r_tag)::value;
#pragma unroll
            for (int c = 0; c < NT2; ++c) {
                if constexpr (dbg & 4) acc[R][c][0] += __builtin_bit_cast(float, bf[c][0] ^ af[h][R][0]);
                else acc[R][c] = Mfma<T>::run(bf[c], af[h][R], acc[R][c]);
            }
            // the next half step's operands: two fragments and 8 / HR lookups behind each of the first HR row tiles
            if constexpr (R < HR) {
                if constexpr (!(dbg & 16)) {
                    frag(nxt_t{}, std::integral_constant<int, nslot>{}, nxt_t{}, std::integral_constant<int, 2 * R>{});
                    frag(nxt_t{}, std::integral_constant<int, nslot>{}, nxt_t{}, std::integral_constant<int, 2 * R + 1>{});
                }
                [&]<int... L>(std::integer_sequence<int, L...>) {
                    (lookup(nxt_t{}, qw, std::integral_constant<int, R * LPR + L>{}), ...);
                }(std::make_integer_sequence<int, LPR>{});
            }
            // batch t+2 behind the last BATCH row tiles of half step 0, one request each
            if constexpr (h == 0 && R >= RT - BATCH)
                issue_nth(std::integral_constant<int, (slot + 2) % BLK_STAGES>{}, std::integral_constant<int, R - (RT - BATCH)>{}, t + 2);
        };
        [&]<int... R>(std::integer_sequence<int, R...>) {
            (row(std::integral_constant<int, R>{}), ...);
        }(std::make_integer_sequence<int, RT>{});
    };

    // scales, batch 0 and the pair table before anyone reads them
    asm volatile("s_waitcnt vmcnt(%1)" : "+v"(w[0]) : "n"(PPW + NWQ) : "memory");
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
    // ---- epilogue 1: the K halves swap half of their row tiles through LDS (K half 0 keeps row tiles 0..HR-1, K half 1
    // keeps HR..RT-1); own[i][t] = row tile HR kh + i over the workgroup's whole K range ----
    f32x4_t own[HR][NT2];
    {
        __syncthreads();                                           // every wave is done with the stages
        float4* xb = reinterpret_cast<float4*>(smem + X_BASE);     // [wave][4 row tiles][2 column tiles][64 lanes] x 16 B = 64 KB
        auto put = [&](int i, int t, const f32x4_t s) { xb[((wave * HR + i) * NT2 + t) * 64 + lane] = make_float4(s[0], s[1], s[2], s[3]); };
        auto get = [&](int i, int t) { const float4 p = xb[(((wave ^ 4) * HR + i) * NT2 + t) * 64 + lane]; return f32x4_t{p.x, p.y, p.z, p.w}; };
        if (kh == 0) {
#pragma unroll
            for (int i = 0; i < HR; ++i)
#pragma unroll
                for (int t = 0; t < NT2; ++t) put(i, t, acc[HR + i][t]);
        } else {
#pragma unroll
            for (int i = 0; i < HR; ++i)
#pragma unroll
                for (int t = 0; t < NT2; ++t) put(i, t, acc[i][t]);
        }
        __syncthreads();
        if (kh == 0) {
#pragma unroll
            for (int i = 0; i < HR; ++i)
#pragma unroll
                for (int t = 0; t < NT2; ++t) own[i][t] = acc[i][t] + get(i, t);
        } else {
#pragma unroll
            for (int i = 0; i < HR; ++i)
#pragma unroll
                for (int t = 0; t < NT2; ++t) own[i][t] = get(i, t) + acc[HR + i][t];       // K half 0 first in both waves: one order
        }
    }

    FLUTE_SKSTAMP(3);
    // ---- epilogue 2: accumulator register j of lane (r16, q4) = weight row 4 q4 + j of the column tile = unit
    // (4 q4 + j) % U, field (4 q4) / U + FPT t: four consecutive columns; the lane's output row is r16 ----
    const int c_unit = unit0 + (4 * q4) % U;
    uint32_t col[NT2];
#pragma unroll
    for (int t = 0; t < NT2; ++t) col[t] = (uint32_t)(unit_col0<BITS, TILEP>(c_unit) + ((4 * q4) / U + FPT * t) * TILEP);
    const int row_base = m0 + kh * (HR * 16) + r16;                // + 16 i
    auto store_d = [&](int i, int t, const f32x4_t o4) {
        const int row = row_base + 16 * i;
        if (row < a.M) {
            uint2 o;
            o.x = (uint32_t)NT::from_float(o4[0]) | ((uint32_t)NT::from_float(o4[1]) << 16);
            o.y = (uint32_t)NT::from_float(o4[2]) | ((uint32_t)NT::from_float(o4[3]) << 16);
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

    if (nsl == 4 && HR % 4 == 0) {
        if constexpr (HR % 4 == 0) e_form(std::integral_constant<int, 4>{});
    } else if (nsl == 2) {
        e_form(std::integral_constant<int, 2>{});
    } else {
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
