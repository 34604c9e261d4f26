// Prefill-regime kernel (M >= ~512): block-tiled LUT-dequant GEMM on MFMA.
//
// gfx950 replacement for qgemm_device's main loop at large M (flute/csrc/qgemm_kernel.hpp:617-712,
// config.hpp:187-558).  The round-1 MFMA kernel (qgemm_tile.h) gives every wave a private 64-row output tile
// with a private K range: no operand is shared between waves, and each weight is LUT-dequantised once per
// 64 rows (64x at M = 4096) - 31 % of the MFMA peak at M = 4096.  Here a WORKGROUP owns a
// (WM x TM x 16) x (WN x 16 units) output block (256 x 256 for 4-bit weights) over the whole K range:
//   * activations: one [BM][64 k] tile per K step, fetched ONCE per workgroup by LDS-DMA
//     (`buffer_load_dwordx4 ... lds`: descriptor + scalar K offset, no per-step VALU address arithmetic, rows
//     beyond M read as zero), three stages in LDS, ONE s_barrier per K step, shared by the WN waves of a row
//     group; the MFMA fragments come out of LDS with conflict-free ds_read_b128 (source-side XOR swizzle);
//   * weights: every wave dequantises the 16 units x J fields (64 columns for 4 bits) it multiplies, straight
//     from packed words in registers (hidden buffer loads, 3-step ring): 16 lookups + 16 v_pk_mul_f16 per
//     32-k half step feed TM x J MFMAs - one lookup per TM MFMA rows instead of one per 4 (TM = 8: a weight is
//     dequantised once per 128 rows, WM = 2 times per workgroup);
//   * scales: 8-group blocks per wave, LDS-DMA through the same in-order queue into a wave-private image;
//   * every hidden load is unconditional, every step issues the same number of them (padding steps read out
//     of range = zeros), every counted wait is one statement (tools/audit_asm_loads.py).
// Arithmetic contract: w^ = round_T(lut * s) (packbits_utils.hpp:139) - v_pk_mul_f16 for fp16, fp32 multiply +
// v_cvt_pk_bf16_f32 for bf16 - fp32 accumulation in the MFMA (config.hpp:323-325), one rounding of the output.
#pragma once
#include <utility>

#include "common.h"
#include "mfma.h"
#include "qgemm_stream.h"      // srd_t, make_srd, buf_load16, hidden LDS lookups

namespace flute_amd {

struct BlockArgs {
    const void* A;          // [M,K] T
    const uint32_t* Q;      // [P,K/2] packed
    void* D;                // [M,N] T
    const void* S;          // [N,G] T
    const uint32_t* QM2;    // [4^b] pair table
    float* partial;         // [splitk][M][N] fp32 when splitk > 1
    int M, N, K, G, lg;
    int tiles_m, tiles_n;   // workgroup tiles
    int splitk, k_per_split;    // k_per_split: multiple of lcm(64, 8 * group_size)
    int order;              // 0: M tiles fastest; 1: XCD x owns a contiguous range of M tiles; 2: of N tiles
};

constexpr int BLK_STAGES = 3;

// 16-B chunk swizzle of a 16-row x 64-B activation piece (an involution applied to the DMA source address and
// to the fragment read): the 16 lanes of every ds_read_b128 lane group then hit 16 different bank slots
__device__ __forceinline__ int blk_swz(int row) { return (4 - (row >> 2)) & 3; }

__host__ __device__ constexpr int block_lds_bytes(int bits, int tm, int wm, int wn) {
    const int lut = (1 << (2 * bits)) * 128;
    const int stage = wm * tm * 2 * 1024;                      // BM/16 row tiles x 2 half steps x 1 KB
    return lut + BLK_STAGES * stage + wm * wn * 3 * 1024 * (bits == 2 ? 2 : 1);     // + per wave: two scale blocks + a sink
}

// LDS-DMA through a buffer descriptor: 16 B per lane from base + voff (per lane, range-checked) + soff
// (wave-uniform) to LDS byte m0 + 16 * lane.  No VGPR destination: only the counted vmcnt orders it.
__device__ __forceinline__ void dma16_buf(uint32_t voff, srd_t srd, uint32_t soff, uint32_t lds_addr) {
    uint32_t keep;                                                 // M0 is compiler-reserved: saved and restored
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(srd), "s"(soff), "s"(lds_addr) : "memory");
}

// releases 8 hidden lookups once at most N younger LDS operations are outstanding
template <int N> __device__ __forceinline__ void blk_lookup_wait(uint32_t (&v)[8]) {
    asm volatile("s_waitcnt lgkmcnt(%8)"
                 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7])
                 : "n"(N) : "memory");
}

// SP (software-pipelined schedule): the LDS work of half step p+1 - activation fragments, pair-table lookups,
// scales - is issued BETWEEN the MFMAs of half step p (after the four MFMAs of a row tile: that row's next
// fragment and 16 / TM lookups), so a wave's matrix pipe time covers its own LDS latency and VALU address
// arithmetic; one s_barrier per 64-k step (mid step, when stage t has been read completely and stage t+1 is about
// to be).  The lockstep schedule (SP = false: dequantise, then multiply, per half step) measured 40 % MFMA-busy;
// separating the two parts by barriers and running the two waves of a SIMD half a phase apart was slower still
// (the load part, ~1000 cycles, is longer than the 512-cycle MFMA part: tools/block_diag.py).
template <typename T, int BITS, int TILEP, int TM, int WM, int WN, bool SP = false>
__global__ __launch_bounds__(WM * WN * 64) void qgemm_block_kernel(const BlockArgs args) {
    static_assert(BITS == 4 || BITS == 2, "3-bit layers use the per-wave MFMA kernel (qgemm_tile.h)");
    using NT = Num<T>;
    constexpr int J = 16 / BITS;                                   // fields per word = column tiles per wave
    constexpr int NW = WM * WN;
    constexpr int BM = WM * TM * 16;
    constexpr int RT = BM / 16;                                    // row tiles of the block
    constexpr int PIECES = RT * 2;                                 // 1-KB activation pieces per stage
    constexpr int PPW = PIECES / NW;                               // ... issued by each wave
    static_assert(PIECES % NW == 0, "activation pieces must divide over the waves");
    constexpr int SLD = (J + 3) / 4;                               // scale loads per lane and block (4 columns each)
    constexpr int BATCH = PPW + 2 + SLD;                           // hidden loads per K step and wave
    constexpr int LUT_BYTES = (1 << (2 * BITS)) * 128;
    constexpr int STAGE_BYTES = PIECES * 1024;
    constexpr bool PRE16 = __is_same(T, F16);

    BlockArgs a = args;
    {
#define FLUTE_OPAQUE(x) asm volatile("" : "+s"(x))
        FLUTE_OPAQUE(a.A); FLUTE_OPAQUE(a.Q); FLUTE_OPAQUE(a.D); FLUTE_OPAQUE(a.S); FLUTE_OPAQUE(a.QM2);
        FLUTE_OPAQUE(a.partial); FLUTE_OPAQUE(a.M); FLUTE_OPAQUE(a.N); FLUTE_OPAQUE(a.K); FLUTE_OPAQUE(a.G);
        FLUTE_OPAQUE(a.lg); FLUTE_OPAQUE(a.tiles_m); FLUTE_OPAQUE(a.tiles_n); FLUTE_OPAQUE(a.splitk);
        FLUTE_OPAQUE(a.k_per_split); FLUTE_OPAQUE(a.order);
#undef FLUTE_OPAQUE
    }
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (lds_base_of(smem) != 0) __builtin_trap();

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int r16 = lane & 15;
    const int q4 = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    // ---- block -> (M tile, N tile, K split).  order 1 / 2: the 8 XCDs (block id % 8, one L2 each) own
    // contiguous ranges of M / N tiles, so that the operand shared inside a range stays in that L2 ----
    int bid = blockIdx.x, split = 0;
    if (a.splitk > 1) { split = bid % a.splitk; bid /= a.splitk; }
    int tm_idx, tn_idx;
    if (a.order == 1) {
        const int per = a.tiles_m >> 3, x = bid & 7, i = bid >> 3;         // i enumerates (m in range, n)
        tm_idx = x * per + i % per;
        tn_idx = i / per;
    } else if (a.order == 2) {
        const int per = a.tiles_n >> 3, x = bid & 7, i = bid >> 3;
        tn_idx = x * per + i % per;
        tm_idx = i / per;
    } else {
        tm_idx = bid % a.tiles_m;
        tn_idx = bid / a.tiles_m;
    }
    const int m0 = tm_idx * BM;
    const int unit0 = (tn_idx * WN + wn) * 16;                     // this wave's 16 units (lane r16 <-> unit)
    const int kbeg = split * a.k_per_split;
    const int kend = min(a.K, kbeg + a.k_per_split);
    const int nsteps = (kend - kbeg) >> 6;                         // 64-k steps
    const int npad = (nsteps + BLK_STAGES - 1) / BLK_STAGES * BLK_STAGES;
    const uint32_t row_bytes = (uint32_t)a.K * 2u;

    // ---- descriptors and per-lane offsets of the three hidden streams ----
    const srd_t x_srd = make_srd(a.A, (uint32_t)min((size_t)a.M * a.K * 2, (size_t)0xfffffff0u));
    const srd_t w_srd = make_srd(reinterpret_cast<const char*>(a.Q) + (size_t)unit0 * row_bytes, 16u * row_bytes);
    const srd_t s_srd = make_srd(a.S, (uint32_t)min((size_t)a.N * a.G * 2, (size_t)0xfffffff0u));
    // activations: piece (half h, row tile rt) = 16 rows x 64 B; lane L fetches chunk (L % 4) ^ swz of row L / 4
    uint32_t x_voff[PPW];
    uint32_t x_lds[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int p = wave * PPW + i;
        const int h = p / RT, rt = p % RT;
        const int row = m0 + rt * 16 + (lane >> 2);
        const int chunk = (lane & 3) ^ blk_swz(lane >> 2);
        x_voff[i] = (row < a.M) ? (uint32_t)(((size_t)row * a.K + h * 32 + chunk * 8) * 2) : 0x80000000u;
        x_lds[i] = (uint32_t)LUT_BYTES + (uint32_t)p * 1024u;
    }
    // weights: lane (r16, q4) reads words 4 q4 .. 4 q4 + 3 of unit r16 (+ 16 words for the second half step)
    const uint32_t w_voff = (uint32_t)r16 * row_bytes + (uint32_t)q4 * 16u;
    // scales: lane (r16, q4) reads 8 groups of column j = q4 (+4 for 2-bit layers) of unit r16
    uint32_t s_voff[SLD];
#pragma unroll
    for (int i = 0; i < SLD; ++i)
        s_voff[i] = (uint32_t)(((size_t)(unit_col0<BITS, TILEP>(unit0 + r16) + (q4 + 4 * i) * TILEP) * a.G) * 2);
    // per wave: [2 block buffers][SLD][1 KB] + a 1-KB sink for the padding requests
    const uint32_t sc_base = (uint32_t)LUT_BYTES + BLK_STAGES * STAGE_BYTES + (uint32_t)wave * (3072u * SLD);
    const uint32_t sc_sink = sc_base + 2048u * SLD;

    u32x4_t w[BLK_STAGES][2];                                      // weight ring: [slot][half step]
    const int gblk0 = (kbeg >> a.lg) >> 3;                         // first 8-group block of this K range
    // batch u = every hidden load of K step u (issued two steps ahead): PPW activation pieces, two weight
    // pieces, SLD scale pieces (real only on the step that starts an 8-group block)
    auto issue_batch = [&](auto slot_tag, int u) {
        constexpr int slot = decltype(slot_tag)::value;
        const bool live = u < nsteps;
        const uint32_t k0 = (uint32_t)(kbeg + u * 64);
        const uint32_t dead = 0x80000000u;
#pragma unroll
        for (int i = 0; i < PPW; ++i)
            dma16_buf(live ? x_voff[i] : dead, x_srd, k0 * 2u, x_lds[i] + (uint32_t)slot * STAGE_BYTES);
#pragma unroll
        for (int h = 0; h < 2; ++h)
            w[slot][h] = buf_load16(live ? w_voff + (uint32_t)h * 64u + k0 * 2u : dead, w_srd, 0);
        const int g = (int)(k0 >> a.lg);
        const bool blk_start = live && ((g & 7) == 0 || u == 0) && ((k0 & ((1u << a.lg) - 1u)) == 0);
        // scale block: LDS-DMA straight into this wave's image (lane-linear = [unit r16][column q4][8 groups]);
        // the padding requests of the other steps land in the sink
#pragma unroll
        for (int i = 0; i < SLD; ++i)
            dma16_buf(blk_start ? s_voff[i] + (uint32_t)((g >> 3) * 16) : dead, s_srd, 0,
                      blk_start ? sc_base + (uint32_t)((g >> 3) & 1) * 1024u * SLD + (uint32_t)i * 1024u : sc_sink);
    };

    // ---- prologue: pair table (32 copies of every entry, 128-B stride), batches 0 and 1 ----
    issue_batch(std::integral_constant<int, 0>{}, 0);
    issue_batch(std::integral_constant<int, 1>{}, 1);
    {
        constexpr int ENT = 1 << (2 * BITS);
        for (int p = tid; p < ENT * 8; p += NW * 64) {
            const uint32_t v = a.QM2[p >> 3];
            *reinterpret_cast<uint4*>(smem + (size_t)(p >> 3) * 128 + (p & 7) * 16) = make_uint4(v, v, v, v);
        }
    }
    const uint32_t lane_off = (uint32_t)(lane & 31) * 4u;
    const uint32_t aread = (uint32_t)(r16 * 4 + (q4 ^ blk_swz(r16))) * 16u;

    f32x4_t acc[TM][J];
#pragma unroll
    for (int t = 0; t < TM; ++t)
#pragma unroll
        for (int j = 0; j < J; ++j) acc[t][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    uint32_t sc[J];                                                // current group's scales of this lane's J columns
#pragma unroll
    for (int j = 0; j < J; ++j) sc[j] = 0;
    int cur_group = -1;

    auto step = [&](auto slot_tag, int t) {
        constexpr int slot = decltype(slot_tag)::value;
        // batch t has landed once at most batch t+1 is outstanding (batches are issued in order)
        asm volatile("s_waitcnt vmcnt(%2)" : "+v"(w[slot][0]), "+v"(w[slot][1]) : "n"(BATCH) : "memory");
        const int k0 = kbeg + t * 64;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                              // every wave's pieces of stage `slot` are in LDS
        issue_batch(std::integral_constant<int, (slot + 2) % BLK_STAGES>{}, t + 2);
        if (t < nsteps) {
            const uint32_t stage = (uint32_t)LUT_BYTES + (uint32_t)slot * STAGE_BYTES;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int grp = (k0 + h * 32) >> a.lg;
                if (grp != cur_group) {                            // wave-uniform
                    // image of a block: [load i][lane (r16, q4) = unit r16, column q4 + 4 i][8 groups]
                    const uint32_t sb = sc_base + (uint32_t)((grp >> 3) & 1) * 1024u * SLD + (uint32_t)(grp & 7) * 2u;
#pragma unroll
                    for (int j = 0; j < J; ++j)
                        sc[j] = lds_ld16(sb + (uint32_t)(j >> 2) * 1024u + (uint32_t)(((j & 3) * 16 + r16) * 16));
                    cur_group = grp;
                }
                u32x4_t af[TM];
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) {
                    const uint4 v = lds_ld128(stage + (uint32_t)(h * RT + wm * TM + tm) * 1024u + aread);
                    af[tm] = u32x4_t{v.x, v.y, v.z, v.w};
                }
                const u32x4_t qw = w[slot][h];
                // Lookups in batches of two column tiles (8 hidden ds_reads each), all issued up front; batch b is
                // released by a COUNTED wait (LDS returns in order; the main loop issues no scalar loads), so the
                // multiplies and MFMAs of batch b run while the later batches are still in flight.
                constexpr int NB = J / 2;
                uint32_t v[NB][8];
#pragma unroll
                for (int b = 0; b < NB; ++b)
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                        for (int ww = 0; ww < 4; ++ww) {
                            const uint32_t idx = __builtin_amdgcn_ubfe(qw[ww], (uint32_t)(2 * BITS * (2 * b + jj)), (uint32_t)(2 * BITS));
                            v[b][jj * 4 + ww] = lds_lookup32((idx << 7) | lane_off);
                        }
                [&]<int... B>(std::integer_sequence<int, B...>) {
                    ([&] {
                        blk_lookup_wait<(NB - 1 - B) * 8>(v[B]);
#pragma unroll
                        for (int jj = 0; jj < 2; ++jj) {
                            u32x4_t bf;
#pragma unroll
                            for (int ww = 0; ww < 4; ++ww) bf[ww] = NT::mul_scale(v[B][jj * 4 + ww], sc[2 * B + jj]);
#pragma unroll
                            for (int tm = 0; tm < TM; ++tm) acc[tm][2 * B + jj] = Mfma<T>::run(bf, af[tm], acc[tm][2 * B + jj]);
                        }
                    }(), ...);
                }(std::make_integer_sequence<int, NB>{});
            }
        }
    };

    // ---- software-pipelined schedule ----
    constexpr int NLK = J * 4;                                     // lookups per half step
    constexpr int LPR = NLK / TM;                                  // ... issued after every row tile
    static_assert(NLK % TM == 0 && NLK == 16 && J == 4 && TM <= 8, "software-pipelined schedule: 4-bit layers");
    uint32_t v[NLK];                                               // hidden lookups of the NEXT half step
    u32x4_t af[TM];                                                // fragments: current, replaced row by row (hidden)
    uint32_t scn[J];                                               // scales of the next half step (hidden)
    // every LDS read of the loop is an inline-asm instruction with an immediate offset: one address register per
    // 64 KB of stages instead of one per (stage, half, row tile), and no compiler-placed lgkmcnt in the MFMA stream
    const uint32_t frag_lo = (uint32_t)LUT_BYTES + (uint32_t)(wm * TM) * 1024u + aread;
    const uint32_t frag_hi = frag_lo + 65536u;
    const uint32_t sc_lane = sc_base + (uint32_t)r16 * 16u;
    auto sp_scales = [&](int t, int h) {                           // -> scn (of this lane's J columns)
        const int grp = (kbeg + t * 64 + h * 32) >> a.lg;
        const uint32_t sb = sc_lane + (uint32_t)((grp >> 3) & 1) * 1024u * SLD + (uint32_t)(grp & 7) * 2u;
        auto one = [&](auto j_tag) {
            constexpr int jc = decltype(j_tag)::value;
            uint32_t& dst = scn[jc];                               // (named first: clang does not capture through asm operands)
            const uint32_t addr = sb;
            asm volatile("ds_read_u16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"((jc >> 2) * 1024 + (jc & 3) * 256) : "memory");
        };
        [&]<int... Jc>(std::integer_sequence<int, Jc...>) {
            (one(std::integral_constant<int, Jc>{}), ...);
        }(std::make_integer_sequence<int, J>{});
    };
    auto sp_lookup = [&](const u32x4_t& qw, auto n_tag) {
        constexpr int n = decltype(n_tag)::value;                  // lookup n: field n / 4, word n % 4
        const uint32_t idx = __builtin_amdgcn_ubfe(qw[n & 3], (uint32_t)(2 * BITS * (n >> 2)), (uint32_t)(2 * BITS));
        v[n] = lds_lookup32((idx << 7) | lane_off);
    };
    auto sp_frag = [&](auto slot_tag, auto h_tag, auto tm_tag) {
        constexpr int off = decltype(slot_tag)::value * STAGE_BYTES + (decltype(h_tag)::value * RT + decltype(tm_tag)::value) * 1024;
        constexpr int tm = decltype(tm_tag)::value;
        u32x4_t& dst = af[tm];
        const uint32_t addr = off < 65536 ? frag_lo : frag_hi;
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off < 65536 ? off : off - 65536) : "memory");
    };
    auto sp_wait_lds = [&]() {                                     // releases every hidden LDS read in flight
        if constexpr (TM == 8) {
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]),
                           "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]),
                           "+v"(af[0]), "+v"(af[1]), "+v"(af[2]), "+v"(af[3]), "+v"(af[4 % TM]), "+v"(af[5 % TM]),
                           "+v"(af[6 % TM]), "+v"(af[7 % TM]), "+v"(scn[0]), "+v"(scn[1]), "+v"(scn[2]), "+v"(scn[3])
                         : : "memory");
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]),
                           "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]),
                           "+v"(af[0]), "+v"(af[1]), "+v"(af[2 % TM]), "+v"(af[3 % TM]),
                           "+v"(scn[0]), "+v"(scn[1]), "+v"(scn[2]), "+v"(scn[3])
                         : : "memory");
        }
    };
    // half step p = (t, h): multiply what half step p-1 looked up, then MFMAs with the LDS work of p+1 between them
    auto sp_half = [&](auto slot_tag, auto h_tag, int t) {
        constexpr int slot = decltype(slot_tag)::value;
        constexpr int h = decltype(h_tag)::value;
        constexpr int nslot = h ? (slot + 1) % BLK_STAGES : slot;  // next half step's stage / ring slot
        constexpr int nh = h ^ 1;
        sp_wait_lds();                                             // lookups, fragments and scales of p are here
        if constexpr (h == 1) {
            // mid step: this wave has read stage t completely; batch t+1 (next ring slot) has landed once at most
            // batch t+2 is outstanding; after the barrier stage t is free for batch t+3
            asm volatile("s_waitcnt vmcnt(%2)" : "+v"(w[nslot][0]), "+v"(w[nslot][1]) : "n"(BATCH) : "memory");
            __builtin_amdgcn_s_barrier();
            issue_batch(slot_tag, t + 3);
        }
        // a step past the end (the ring is padded to whole triples) multiplies by zero scales: its activations are
        // the zeros of an out-of-range LDS-DMA, its weights index 0 of the table
        const bool live = t < nsteps;
        u32x4_t bf[J];
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const uint32_t sj = live ? scn[j] : 0u;
#pragma unroll
            for (int ww = 0; ww < 4; ++ww) bf[j][ww] = NT::mul_scale(v[j * 4 + ww], sj);
        }
        sp_scales(t + h, nh);
        const u32x4_t qw = w[nslot][nh];
        auto row = [&](auto r_tag) {
            constexpr int R = decltype(r_tag)::value;
#pragma unroll
            for (int j = 0; j < J; ++j) acc[R][j] = Mfma<T>::run(bf[j], af[R], acc[R][j]);
            sp_frag(std::integral_constant<int, nslot>{}, std::integral_constant<int, nh>{}, r_tag);
            [&]<int... L>(std::integer_sequence<int, L...>) {
                (sp_lookup(qw, std::integral_constant<int, R * LPR + L>{}), ...);
            }(std::make_integer_sequence<int, LPR>{});
        };
        [&]<int... R>(std::integer_sequence<int, R...>) {
            (row(std::integral_constant<int, R>{}), ...);
        }(std::make_integer_sequence<int, TM>{});
    };

    if constexpr (SP) {
        issue_batch(std::integral_constant<int, 2>{}, 2);
        // batch 0 and the pair table before anyone reads them
        asm volatile("s_waitcnt vmcnt(%2)" : "+v"(w[0][0]), "+v"(w[0][1]) : "n"(2 * BATCH) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        sp_scales(0, 0);
        {
            const u32x4_t qw = w[0][0];
            [&]<int... R>(std::integer_sequence<int, R...>) {
                (sp_frag(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, R>{}), ...);
            }(std::make_integer_sequence<int, TM>{});
            [&]<int... L>(std::integer_sequence<int, L...>) {
                (sp_lookup(qw, std::integral_constant<int, L>{}), ...);
            }(std::make_integer_sequence<int, NLK>{});
        }
        for (int t0 = 0; t0 < npad; t0 += BLK_STAGES) {
            [&]<int... I>(std::integer_sequence<int, I...>) {
                ((sp_half(std::integral_constant<int, I>{}, std::integral_constant<int, 0>{}, t0 + I),
                  sp_half(std::integral_constant<int, I>{}, std::integral_constant<int, 1>{}, t0 + I)), ...);
            }(std::make_integer_sequence<int, BLK_STAGES>{});
        }
        sp_wait_lds();                                             // the prefetch past the end
    } else
    for (int t0 = 0; t0 < npad; t0 += BLK_STAGES) {
        [&]<int... I>(std::integer_sequence<int, I...>) {
            (step(std::integral_constant<int, I>{}, t0 + I), ...);
        }(std::make_integer_sequence<int, BLK_STAGES>{});
    }
    // the last two batches are out-of-range reads still in flight: drain before the registers die
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(w[0][0]), "+v"(w[0][1]), "+v"(w[1][0]), "+v"(w[1][1]), "+v"(w[2][0]), "+v"(w[2][1]) : : "memory");

    // ---- epilogue: lane (r16, q4) of tile (tm, j) holds output row r16 and columns 4 q4 .. 4 q4 + 3 of
    // column tile j = columns of units 4 q4 .. 4 q4 + 3 at field j ----
    const int c_unit = unit0 + q4 * 4;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        const int row = m0 + (wm * TM + tm) * 16 + r16;
        if (row < a.M) {
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const int col = unit_col0<BITS, TILEP>(c_unit) + j * TILEP;
                const f32x4_t v = acc[tm][j];
                if (a.splitk == 1) {
                    uint2 o;
                    o.x = (uint32_t)NT::from_float(v[0]) | ((uint32_t)NT::from_float(v[1]) << 16);
                    o.y = (uint32_t)NT::from_float(v[2]) | ((uint32_t)NT::from_float(v[3]) << 16);
                    *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(a.D) + (size_t)row * a.N + col) = o;
                } else {
                    *reinterpret_cast<f32x4_t*>(a.partial + ((size_t)split * a.M + row) * a.N + col) = v;
                }
            }
        }
    }
}

}  // namespace flute_amd
