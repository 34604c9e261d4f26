// MFMA kernel for every M above the decode kernel's range, operands staged by LDS-DMA.
//
// gfx950 replacement for qgemm_device (flute/csrc/qgemm_kernel.hpp:617-712).  One
// v_mfma_f32_16x16x32 consumes 16 weight columns x 32 k x 16 activation rows.  A wave owns a SLAB
// of 16/R units (R lanes share one unit's words and each takes J/R of its fields => J/R column
// tiles per k-step; R > 1 gives a narrow layer 2-4x more slabs, so that it fills 256 CUs without a
// grid-level K split and its second launch), MT 16-row tiles of activations, and one K range of
// the workgroup's in-LDS K split.
//
// The design point is how the operands reach the registers.  An MFMA operand wants lane (r, q) to hold
// 16 B of ROW r (r = lane % 16): 64 lanes, 64 different cache lines.  The texture addresser
// serves such a load at one lane per clock (measured, tools/ubench/ta_patterns.hip: 61.8 cycles
// per wave-instruction, 16.6 B/clk/CU, against 17.4 cycles when 4..8 neighbouring lanes share a
// line) and that, not MFMA, LDS or L2 bandwidth, bounded the first kernel of this round, which
// loaded operands in that order (10 % of the MFMA peak at M = 256, 6.6 us of load issue at
// M = 16).  Here every global access is line-coalesced:
//   * global_load_lds_dwordx4 (LDS-DMA): lane L fetches 16 B of row L / (4R) (weights; a piece is
//     16/R rows x R k-steps x 64 B) or row L / 4 (activations; 16 rows x 64 B), written to LDS
//     lane-linearly; no VGPR round trip, no ds_write;
//   * the MFMA layout is produced by the ds_read_b128 that follows; the 16-B chunk index inside a
//     row is XOR-swizzled on the SOURCE address (and identically on the read) so that the reads
//     are bank-conflict free;
//   * the ring of in-flight pieces lives in wave-private LDS: no barrier in the main loop, the
//     wave waits for its own DMA with counted s_waitcnt vmcnt(N).
// The weights are the MFMA's A operand and the activations its B operand, so the accumulator
// of lane (r, q) holds output row r and FOUR CONSECUTIVE columns 4q..4q+3: the epilogue stores
// 8 B per lane (16 B for split-K partials) instead of four 2-B scatters, and all waves of the
// workgroup take part in the cross-wave K reduction.
//
// Arithmetic contract as before: fp16 w^ = round_T(lut * s) with v_pk_mul_f16 (the reference's
// packbits_utils.hpp:139); bf16 applies the group scale to the fp32 MFMA result of each group run.
#pragma once
#include "common.h"
#include "mfma.h"
#include "xwg.h"

namespace flute_amd {

constexpr int TILE_GB = 8;           // scale groups per block (one 16-B DMA granule per column)
#ifndef FLUTE_TILE_LUT_SHIFT
#define FLUTE_TILE_LUT_SHIFT 7        // log2 bytes per pair-table entry: 7 = 32 copies (conflict-free), 6 = 16 copies
#endif
constexpr int TILE_LUT_SHIFT = FLUTE_TILE_LUT_SHIFT;

struct TileGeom {
    int scale_bytes;   // wave-private scale blocks: 2 buffers x [column tile][16 columns][8 groups]
    int slot_bytes;    // one ring slot = one macro-step (R k-steps): NP*SW weight + MT*R activation pieces of 1 KB
    int depth;         // ring slots per wave
    int wave_bytes;
    int total;         // dynamic LDS of the launch (main loop carve or epilogue partial tiles)
};

__host__ __device__ inline TileGeom tile_geom(int bits, int R, int mt, int sw, int waves, int budget) {
    const int J = (bits == 3) ? 16 : 16 / bits;
    const int NP = (bits == 3) ? 3 : 1;
    const int nmf = J / R * sw;
    const int lut = (1 << (2 * bits)) << TILE_LUT_SHIFT;
    const int lps = NP * sw + mt * R;
    TileGeom g;
    g.scale_bytes = 2 * (nmf > 4 ? nmf : 4) * 256;
    g.slot_bytes = lps * 1024;
    int d = ((budget - lut) / waves - g.scale_bytes) / g.slot_bytes;
    if (d > 6) d = 6;
    while (d > 1 && (d - 1) * lps > 56) --d;          // vmcnt is a 6-bit counter
    g.depth = d;
    g.wave_bytes = g.scale_bytes + d * g.slot_bytes;
    const int main_bytes = lut + waves * g.wave_bytes;
    const int tiles = mt * nmf;
    const int epi_bytes = waves * (tiles > 16 ? 16 : tiles) * 1024;   // partial tiles of one epilogue pass
    g.total = main_bytes > epi_bytes ? main_bytes : epi_bytes;
    return g;
}

// LDS-DMA: 16 B per lane from `g` to LDS byte `lds_addr + 16 * lane` (wave-uniform base in M0).
// Invisible to hipcc's s_waitcnt bookkeeping: completion is waited for by hand (dma_wait).
__device__ __forceinline__ void dma16(const void* g, uint32_t lds_addr) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(lds_addr) : "memory");
}
template <int N> __device__ __forceinline__ void vm_wait() {
    asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory");
}
// wait until at most `steps` macro-steps (LPS loads each) issued later are still in flight
template <int LPS> __device__ __forceinline__ void dma_wait(int steps) {
    switch (steps) {
        case 0: vm_wait<0>(); break;
        case 1: vm_wait<LPS>(); break;
        case 2: vm_wait<2 * LPS>(); break;
        case 3: vm_wait<(3 * LPS > 63 ? 63 : 3 * LPS)>(); break;
        case 4: vm_wait<(4 * LPS > 63 ? 63 : 4 * LPS)>(); break;
        default: vm_wait<(5 * LPS > 63 ? 63 : 5 * LPS)>(); break;
    }
}

// chunk swizzles (an involution applied to the source address and to the read address): with
// them the 16 lanes of every ds_read_b128 lane group hit 16 different 16-B bank slots
__device__ __forceinline__ int swz_a(int row) { return (4 - (row >> 2)) & 3; }       // 16 rows x 4 chunks
// activation pieces of 8 rows x 8 chunks (two pieces = the two row halves of a 16-row tile): the 16 lanes of a
// ds_read_b128 lane group (rows 0..15, one chunk) then hit 16 different 16-B slots of the 256-B bank row
__device__ __forceinline__ int swz_x(int row8, int h) { return (row8 >> 1) | (h << 2); }
template <int R> __device__ __forceinline__ int swz_q(int row) {                      // 16/R rows x 4R chunks
    if constexpr (R == 1) return (4 - (row >> 2)) & 3;
    else if constexpr (R == 2) return (row >> 1) << 1;
    else return row << 2;
}

template <typename T, int BITS, int TILEP, int R, int MT, int SW = 1>
__global__ __launch_bounds__(512) void qgemm_tile_kernel(const QGemmArgs a) {
    using L = Layout<BITS>;
    using NT = Num<T>;
    constexpr int J = L::J;
    constexpr int NP = L::NPLANES;
    constexpr int NMFS = J / R;                // column tiles of one slab per k-step and row tile
    constexpr int NMF = NMFS * SW;             // SW slabs per wave: every activation fragment serves SW x NMFS tiles
    constexpr int SU = 16 / R;                 // units per slab
    constexpr int QP = NP * SW;                // weight pieces per macro-step
    constexpr int LPS = QP + MT * R;           // DMA pieces per macro-step
    constexpr int GB = TILE_GB;
    constexpr int LUT_BYTES = (1 << (2 * BITS)) << TILE_LUT_SHIFT;   // 32 copies: one per bank of a ds_read_b32 lane group
    constexpr bool PRE = __is_same(T, F16);
    static_assert(BITS != 3 || R == 1, "3-bit fields are not byte aligned: R = 1 only");
    static_assert(SW == 1 || R == 1, "several slabs per wave only without lane sharing");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (lds_base_of(smem) != 0) __builtin_trap();
#ifdef FLUTE_STAMPS   // development build: 100 MHz wall-clock stamps per wave into the (unused) workspace
    uint64_t stamp[8];
    stamp[0] = wall_clock64();
#define FLUTE_STAMP(i) stamp[i] = wall_clock64()
#else
#define FLUTE_STAMP(i)
#endif

#ifdef FLUTE_ABLATE   // development builds only: 1 no table lookups, 2 no activation reads, 4 no refills, 8 no MFMA
    constexpr int dbg = FLUTE_ABLATE;
#else
    constexpr int dbg = 0;
#endif
    const int tid = threadIdx.x;
    const int nthr = blockDim.x;
    const int lane = tid & 63;
    const int r16 = lane & 15;
    const int q4 = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nw = nthr >> 6;
    const int kw = a.kw;
    const int lkw = a.lkw;                     // kw and the waves per workgroup are powers of two
    const int ns = nw >> lkw;
    const int sl = wave >> lkw;
    const int kpart = wave & (kw - 1);
    const int lg = a.lg;

    // block -> (slab group, row tile, K split); the common single-tile / unsplit cases divide nothing
    int bid = blockIdx.x, split = 0, mtile = 0;
    if (a.splitk > 1) { split = bid % a.splitk; bid /= a.splitk; }
    const int mtiles = a.geo[4];
    int sg = bid;
    if (mtiles > 1) {
        if (a.geo[5]) {
            // XCD-aware: consecutive blocks go to consecutive XCDs (8, one L2 each), so the row tiles
            // that share a weight slab are given block ids 8 apart and stream it through ONE L2
            const int j = bid >> 3;
            mtile = j % mtiles;
            sg = (j / mtiles) * 8 + (bid & 7);
        } else {
            mtile = bid % mtiles;
            sg = bid / mtiles;
        }
    }
    const int m0 = mtile * (MT * 16);
    const int slab = (sg * ns + sl) * SW;      // first of this wave's SW consecutive slabs
    // MFMA role of this lane on the weight side: column r16 of every column tile =
    // unit (r16 % SU) of the slab, field i*R + (r16 / SU) of that unit in tile i
    const int ul = r16 % SU;
    const int f = r16 / SU;
    const int kbeg = split * a.k_per_split;
    const int kend = min(a.K, kbeg + a.k_per_split);
    const int kpw = (((kend - kbeg + kw - 1) >> lkw) + 32 * R - 1) / (32 * R) * (32 * R);
    const int kb = min(kend, kbeg + kpart * kpw);
    const int ke = min(kend, kb + kpw);
    const int nsteps = (ke - kb) >> 5;
    const int nmacro = (nsteps + R - 1) / R;

    // LDS carve of tile_geom(), evaluated by the host planner
    const int D = a.geo[0];
    const uint32_t sc_base = LUT_BYTES + (uint32_t)wave * (uint32_t)a.geo[3];
    const uint32_t ring0 = sc_base + (uint32_t)a.geo[1];
    const uint32_t slot_bytes = (uint32_t)a.geo[2];

    const uint16_t* A = reinterpret_cast<const uint16_t*>(a.A);
    const uint16_t* S = reinterpret_cast<const uint16_t*>(a.S);
    const size_t row_words = (size_t)(a.K >> 1);

    // ---- DMA roles: which 16 B of which row this lane fetches for a weight / activation piece ----
    const int lrow = lane / (4 * R);                                  // weight piece row (unit of the slab)
    const int qchunk = (lane % (4 * R)) ^ swz_q<R>(lrow);             // source chunk (8 k each) inside the macro-step
    const uint32_t* qrow[QP];                                         // [slab w][plane pl] -> piece w * NP + pl
#pragma unroll
    for (int w = 0; w < SW; ++w)
#pragma unroll
        for (int pl = 0; pl < NP; ++pl)
            qrow[w * NP + pl] = a.Q + (size_t)unit_row<BITS, TILEP>((slab + w) * SU + lrow, pl, a.N) * row_words;
    // Activations.  R = 1 (one k-step per macro-step): a piece is 16 rows x 64 B (lane L: row L / 4).  R > 1: a piece is
    // 8 rows x 128 B - WHOLE cache lines (lane L: row L / 8, chunk L % 8 of the 64 k of k-steps 2 kh, 2 kh + 1): a request
    // costs the CU's texture addresser time per line it touches (tools/stamps_skinny.py: ~270 cycles of a SIMD's memory
    // issue for 16 half lines), and the activation requests are MT R of the MT R + NP SW requests of a macro-step.  Piece
    // e = h (R / 2) + kh of a row tile: rows 8 h .. 8 h + 7.
    constexpr bool XLINES = R > 1;
    const int arow = XLINES ? (lane >> 3) : (lane >> 2);
    const int achunk = XLINES ? (lane & 7) : ((lane & 3) ^ swz_a(arow));
    const uint16_t* xrow[MT][XLINES ? 2 : 1];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)      // row >= M: clamped; that accumulator column is never stored
#pragma unroll
        for (int h = 0; h < (XLINES ? 2 : 1); ++h)
            xrow[mt][h] = A + (size_t)min(m0 + mt * 16 + h * 8 + arow, a.M - 1) * a.K;
    const int klim = a.K - 8;            // a lane never reads past its row (ragged last macro-step)

    // piece P of a macro-step starting at k0: P < QP weight piece (slab, plane), else activations (k-step s, row tile mt)
    auto issue_piece = [&](int P, int k0, uint32_t slot_addr) {
        if (P < QP) {
            const int kq = min(k0 + qchunk * 8, klim);
            dma16(qrow[P] + (kq >> 1), slot_addr + P * 1024);
        } else {
            const int e = (P - QP) / MT, mt = (P - QP) % MT;
            if constexpr (XLINES) {
                const int h = e / (R / 2), kh = e % (R / 2);
                const int ka = min(k0 + kh * 64 + (achunk ^ swz_x(arow, h)) * 8, klim);
                dma16(xrow[mt][h] + ka, slot_addr + P * 1024);
            } else {
                const int ka = min(k0 + e * 32 + achunk * 8, klim);
                dma16(xrow[mt][0] + ka, slot_addr + P * 1024);
            }
        }
    };
    auto issue = [&](int t, uint32_t slot_addr) {
#pragma unroll
        for (int P = 0; P < LPS; ++P) issue_piece(P, kb + t * (32 * R), slot_addr);
    };
    // ---- scale blocks: 8 groups x (NMF x 16 weight-side columns), double buffered, fetched by DMA one
    // block ahead.  Lane L of a piece fetches the 16 B (8 groups) of column (L % 16) of tile
    // (L / 16); LDS image [tile][column][8 groups].  Needs 16-B aligned rows of S (G % 8 == 0);
    // other shapes stage synchronously with plain loads (sync_scales). ----
    constexpr int SPIECES = (NMF + 3) / 4;                            // 1 KB pieces per block
    constexpr int SBUF = (NMF > 4 ? NMF : 4) * 256;                   // bytes per buffer
    const bool sdma = (a.G % 8) == 0;
    const uint16_t* srow[SPIECES];
#pragma unroll
    for (int sp = 0; sp < SPIECES; ++sp) {
        const int i = min(sp * 4 + q4, NMF - 1);                      // NMF < 4: the surplus lanes refetch the last tile
        const int col = unit_col0<BITS, TILEP>((slab + i / NMFS) * SU + r16 % SU) + ((i % NMFS) * R + r16 / SU) * TILEP;
        srow[sp] = S + (size_t)col * a.G;
    }
    const int blk_last = (nsteps > 0) ? ((ke - 1) >> lg) >> 3 : -1;     // last block this wave touches
    auto issue_scales = [&](int blk) {
        const uint32_t dst = sc_base + (uint32_t)(blk & 1) * SBUF;
#pragma unroll
        for (int sp = 0; sp < SPIECES; ++sp) dma16(srow[sp] + blk * 8, dst + sp * 1024);
    };
    auto sync_scales = [&](int blk) {                                 // G % 8 != 0: bounds-checked, hipcc-visible loads
        uint16_t* dst = reinterpret_cast<uint16_t*>(smem + sc_base + (uint32_t)(blk & 1) * SBUF);
#pragma unroll
        for (int sp = 0; sp < SPIECES; ++sp) {
            uint16_t hv[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) hv[r] = (blk * 8 + r < a.G) ? srow[sp][blk * 8 + r] : (uint16_t)0;
#pragma unroll
            for (int r = 0; r < 8; ++r) dst[(sp * 64 + lane) * 8 + r] = hv[r];
        }
    };

    // ---- prologue: everything the first k-step needs travels together (one memory latency):
    // pair-table words (registers), first scale block and ring slot 0 (DMA) ----
    constexpr int ENT = 1 << (2 * BITS);
    constexpr int LPC = (1 << TILE_LUT_SHIFT) / 64;                   // 64-B pieces per table entry
    constexpr int LUT_R = (ENT * LPC + 511) / 512;                    // table pieces per thread at 512 threads
    uint32_t lutv[LUT_R];
#pragma unroll
    for (int r = 0; r < LUT_R; ++r) {
        const int p = tid + r * nthr;
        lutv[r] = (p < ENT * LPC) ? a.QM2[p / LPC] : 0u;
    }
    if (nsteps > 0 && sdma) issue_scales((kb >> lg) >> 3);
    if (nmacro > 0) issue(0, ring0);
    FLUTE_STAMP(4);

    // ---- MFMA-side addresses inside a slot ----
    const uint32_t qread = (uint32_t)(ul * (4 * R)) * 16;            // + ((s*4 + q4) ^ swz) * 16
    const int qswz = swz_q<R>(ul);
    const uint32_t aread = (uint32_t)(r16 * 4 + (q4 ^ swz_a(r16))) * 16;     // R = 1
    uint32_t aread_s[R];                                                      // R > 1: k-step s of the macro-step
#pragma unroll
    for (int s = 0; s < R; ++s)
        aread_s[s] = (uint32_t)(((r16 >> 3) * (R / 2 > 0 ? R / 2 : 1) + (s >> 1)) * MT) * 1024u +
                     (uint32_t)((r16 & 7) * 8 + ((((s & 1) * 4 + q4)) ^ swz_x(r16 & 7, r16 >> 3))) * 16u;
    const uint32_t lane_off = (uint32_t)(lane & ((1 << (TILE_LUT_SHIFT - 2)) - 1)) * 4;   // table copy of this lane

    // ---- pair table: entry e at [e * 128, +128): 32 copies of its 4 bytes ----
#pragma unroll
    for (int r = 0; r < LUT_R; ++r) {
        const int p = tid + r * nthr;
        if (p < ENT * LPC) {
            uint4* d = reinterpret_cast<uint4*>(smem + (size_t)(p / LPC) * (LPC * 64) + (p % LPC) * 64);
            const uint4 vv = make_uint4(lutv[r], lutv[r], lutv[r], lutv[r]);
            d[0] = vv; d[1] = vv; d[2] = vv; d[3] = vv;
        }
    }
    for (int p = tid + LUT_R * nthr; p < ENT * LPC; p += nthr) {      // small workgroups only
        const uint32_t v = a.QM2[p / LPC];
        uint4* d = reinterpret_cast<uint4*>(smem + (size_t)(p / LPC) * (LPC * 64) + (p % LPC) * 64);
        const uint4 vv = make_uint4(v, v, v, v);
        d[0] = vv; d[1] = vv; d[2] = vv; d[3] = vv;
    }
    FLUTE_STAMP(5);
    if (nsteps > 0 && !sdma) sync_scales((kb >> lg) >> 3);
    FLUTE_STAMP(6);
    // the rest of the ring (the table words above travelled with slot 0 only)
    for (int t = 1; t < D && t < nmacro; ++t) issue(t, ring0 + (uint32_t)t * slot_bytes);
    __syncthreads();                             // table visible (only barrier before the epilogue)
    FLUTE_STAMP(1);
    int t_sc = 0;                                // macro-step during which the next scale block was requested

    f32x4_t acc[MT][NMF], run[MT][NMF];
    uint32_t sreg[NMF];                          // fp16: current group's scale of weight column r16 (raw T, low half)
#pragma unroll
    for (int i = 0; i < NMF; ++i) {
        sreg[i] = 0;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            acc[mt][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            run[mt][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        }
    }
    int cur_group = -1;
    // bf16: fold the finished group's fp32 run into the accumulator with that group's scales (read
    // back from the block buffer: the accumulator's 4 columns are weight-side columns 4q..4q+3)
    auto fold_run = [&](int grp) {
        const uint32_t sb = sc_base + (uint32_t)((grp >> 3) & 1) * SBUF + (uint32_t)(grp & 7) * 2;
#pragma unroll
        for (int i = 0; i < NMF; ++i) {
            float sf[4];
#pragma unroll
            for (int e = 0; e < 4; ++e)
                sf[e] = NT::to_float(*reinterpret_cast<const uint16_t*>(smem + sb + (uint32_t)(i * 16 + q4 * 4 + e) * 16));
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[mt][i][e] = __builtin_fmaf(run[mt][i][e], sf[e], acc[mt][i][e]);
                run[mt][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            }
        }
    };

    uint32_t slot = ring0;                       // slot of macro-step t
    int slot_idx = 0;
#ifdef FLUTE_STAMPS
    uint64_t cyc_wait = 0, cyc_loop0 = __builtin_readcyclecounter();
#endif
    for (int t = 0; t < nmacro; ++t) {
#ifdef FLUTE_STAMPS
        const uint64_t cw0 = __builtin_readcyclecounter();
#endif
        dma_wait<LPS>((dbg & 4) ? 0 : min(D - 1, nmacro - 1 - t));
#ifdef FLUTE_STAMPS
        cyc_wait += __builtin_readcyclecounter() - cw0;
#endif
#ifndef FLUTE_TILE_NO_BATCH
        bool batched = false;
        if constexpr (R > 1) {
            // ---- lane-sharing variants (R k-steps per macro-step, J/R column tiles each): a full macro-step
            // runs as ONE dependent chain - every word/activation read, then every pair lookup, then the
            // multiplies and MFMAs - instead of R chains of (read, wait, lookup, wait, multiply, MFMA); no
            // branch between the reads.  A ragged last macro-step takes the k-step loop below. ----
            if ((t + 1) * R <= nsteps) {
                batched = true;
                uint32_t sc[R][NMF];
                if constexpr (PRE) {
#pragma unroll
                for (int s = 0; s < R; ++s) {
                    const int grp = (kb + (t * R + s) * 32) >> lg;
                    if (grp != cur_group) {
                        const int blk = grp >> 3;
                        if (cur_group < 0 || blk != (cur_group >> 3)) {   // entering a scale block (see below)
                            if (sdma) {
                                if (cur_group >= 0 && t - t_sc < D) vm_wait<0>();
                                if (blk < blk_last) { issue_scales(blk + 1); t_sc = t; }
                            } else if (cur_group >= 0) {
                                sync_scales(blk);
                            }
                        }
                        const uint32_t sb = sc_base + (uint32_t)(blk & 1) * SBUF + (uint32_t)(grp & 7) * 2;
#pragma unroll
                        for (int i = 0; i < NMF; ++i)
                            sreg[i] = *reinterpret_cast<const uint16_t*>(smem + sb + (uint32_t)(i * 16 + r16) * 16);
                        cur_group = grp;
                    }
#pragma unroll
                    for (int i = 0; i < NMF; ++i) sc[s][i] = sreg[i];     // in registers: the block buffer may be refilled
                }
                }
                uint32_t qw[R][4];
                u32x4_t af[R][MT];
#pragma unroll
                for (int s = 0; s < R; ++s) {
                    const uint4 v = lds_ld128(slot + qread + (uint32_t)(((s * 4 + q4) ^ qswz) * 16));
                    qw[s][0] = v.x; qw[s][1] = v.y; qw[s][2] = v.z; qw[s][3] = v.w;
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        const uint4 x = lds_ld128(slot + (QP + mt) * 1024 + aread_s[s]);
                        af[s][mt] = u32x4_t{x.x, x.y, x.z, x.w};
                    }
                }
                const bool refill = (t + D < nmacro);
                if (refill) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // every read of the slot has returned
#pragma unroll
                    for (int P = 0; P < LPS; ++P) issue_piece(P, kb + (t + D) * (32 * R), slot);
                }
                uint32_t lut[R][NMF][4];
#pragma unroll
                for (int s = 0; s < R; ++s)
#pragma unroll
                    for (int i = 0; i < NMF; ++i)
#pragma unroll
                        for (int ww = 0; ww < 4; ++ww) {
                            const uint32_t idx = __builtin_amdgcn_ubfe(qw[s][ww], (uint32_t)(2 * BITS) * (uint32_t)(i * R + f),
                                                                       (uint32_t)(2 * BITS));
                            lut[s][i][ww] = lds_ld32((idx << TILE_LUT_SHIFT) | lane_off);
                        }
#pragma unroll
                for (int s = 0; s < R; ++s) {
                    if constexpr (!PRE) {
                        // bf16: group bookkeeping between the MFMAs, in the k-step loop's order (fold the finished
                        // group's run, then request the next scale block); the reads and lookups are already out
                        const int grp = (kb + (t * R + s) * 32) >> lg;
                        if (grp != cur_group) {
                            if (cur_group >= 0) fold_run(cur_group);
                            const int blk = grp >> 3;
                            if (cur_group < 0 || blk != (cur_group >> 3)) {
                                if (sdma) {
                                    if (cur_group >= 0 && t - t_sc < D) vm_wait<0>();
                                    if (blk < blk_last) { issue_scales(blk + 1); t_sc = t; }
                                } else if (cur_group >= 0) {
                                    sync_scales(blk);
                                }
                            }
                            cur_group = grp;
                        }
                    }
#pragma unroll
                    for (int i = 0; i < NMF; ++i) {
                        u32x4_t bf;
#pragma unroll
                        for (int ww = 0; ww < 4; ++ww) bf[ww] = PRE ? NT::mul_scale(lut[s][i][ww], sc[s][i]) : lut[s][i][ww];
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) {
                            if constexpr (PRE) acc[mt][i] = Mfma<T>::run(bf, af[s][mt], acc[mt][i]);
                            else run[mt][i] = Mfma<T>::run(bf, af[s][mt], run[mt][i]);
                        }
                    }
                }
            }
        }
        if (!batched)
#endif
#pragma unroll
        for (int s = 0; s < R; ++s) {
            const int ks = t * R + s;
            if (ks < nsteps) {                                            // wave-uniform
                const int k0 = kb + ks * 32;
                const int grp = k0 >> lg;
                if (grp != cur_group) {
                    if constexpr (!PRE) { if (cur_group >= 0) fold_run(cur_group); }
                    const int blk = grp >> 3;
                    if (cur_group < 0 || blk != (cur_group >> 3)) {       // entering a scale block
                        if (sdma) {
                            // the block was requested during macro-step t_sc, ahead of that step's refill: the
                            // slot wait of step t covers it once t >= t_sc + D (loads retire in order); a
                            // block entered sooner (short first block, g = 32 with R = 4) drains the queue
                            if (cur_group >= 0 && t - t_sc < D) vm_wait<0>();
                            if (blk < blk_last) { issue_scales(blk + 1); t_sc = t; }   // into the buffer the previous block left
                        } else if (cur_group >= 0) {
                            sync_scales(blk);
                        }
                    }
                    if constexpr (PRE) {
                        const uint32_t sb = sc_base + (uint32_t)(blk & 1) * SBUF + (uint32_t)(grp & 7) * 2;
#pragma unroll
                        for (int i = 0; i < NMF; ++i)
                            sreg[i] = *reinterpret_cast<const uint16_t*>(smem + sb + (uint32_t)(i * 16 + r16) * 16);
                    }
                    cur_group = grp;
                }
                uint32_t qw[QP][4];
#pragma unroll
                for (int pl = 0; pl < QP; ++pl) {
                    const uint4 v = lds_ld128(slot + pl * 1024 + qread + (uint32_t)(((s * 4 + q4) ^ qswz) * 16));
                    qw[pl][0] = v.x; qw[pl][1] = v.y; qw[pl][2] = v.z; qw[pl][3] = v.w;
                }
                u32x4_t af[MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    if ((dbg & 2) && mt > 0) { af[mt] = af[0]; continue; }
                    const uint4 v = XLINES ? lds_ld128(slot + (QP + mt) * 1024 + aread_s[s]) : lds_ld128(slot + (QP + s * MT + mt) * 1024 + aread);
                    af[mt] = u32x4_t{v.x, v.y, v.z, v.w};
                }
                // Refill of this slot for macro-step t + D, spread over the k-step: a piece may be overwritten
                // as soon as its reads have returned (the activations of k-step s now, the weights after the
                // last k-step); one piece goes out after each column tile's MFMAs so that the wave is not
                // parked in the texture addresser's queue for five pieces in a row
                const bool refill = (t + D < nmacro) && !(dbg & 4);
                // (whole-line activation pieces serve both k-steps of a pair: everything is refilled in the last k-step)
                const int NREF = XLINES ? ((s == R - 1) ? LPS : 0) : MT + ((s == R - 1) ? QP : 0);       // constant after unrolling
                if (refill) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                // the pair lookups of (up to) four column tiles are issued before their first multiply:
                // hipcc otherwise funnels them through one register (lookup, wait, multiply, 16 times)
                constexpr int IB = NMF < 4 ? NMF : 4;
#pragma unroll
                for (int i0 = 0; i0 < NMF; i0 += IB) {
                    uint32_t lut[IB][4];
#pragma unroll
                    for (int ii = 0; ii < IB; ++ii) {
                        const int i = i0 + ii;
#pragma unroll
                        for (int ww = 0; ww < 4; ++ww) {
                            uint32_t addr;
                            const int sw_ = i / NMFS, it_ = i % NMFS;         // slab and tile inside the slab (constants after unrolling)
                            if constexpr (BITS == 4 || BITS == 2) {
                                // v_bfe_u32 + v_lshl_or_b32 (per-lane field when R lanes share the word)
                                const uint32_t wq = qw[sw_ * NP][ww];
                                const uint32_t idx = (R == 1) ? ((wq >> (2 * BITS * it_)) & ((1u << (2 * BITS)) - 1u))
                                                              : __builtin_amdgcn_ubfe(wq, (uint32_t)(2 * BITS) * (uint32_t)(it_ * R + f),
                                                                                      (uint32_t)(2 * BITS));
                                addr = (idx << TILE_LUT_SHIFT) | lane_off;
                            } else {
                                uint32_t wv[NP];
#pragma unroll
                                for (int pl = 0; pl < NP; ++pl) wv[pl] = qw[sw_ * NP + pl][ww];
                                addr = (field<BITS>(wv, it_) << TILE_LUT_SHIFT) | lane_off;
                            }
                            lut[ii][ww] = (dbg & 1) ? addr : lds_ld32(addr);
                        }
                    }
#pragma unroll
                    for (int ii = 0; ii < IB; ++ii) {
                        const int i = i0 + ii;
                        u32x4_t bf;
#pragma unroll
                        for (int ww = 0; ww < 4; ++ww) bf[ww] = PRE ? NT::mul_scale(lut[ii][ww], sreg[i]) : lut[ii][ww];
                        // weights are the A operand: lane (r, q) of the result = output row r, columns 4q..4q+3
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) {
                            if (dbg & 8) { acc[mt][i][0] += __builtin_bit_cast(float, bf[0] ^ bf[1] ^ bf[2] ^ bf[3] ^ af[mt][0]); continue; }
                            if constexpr (PRE) acc[mt][i] = Mfma<T>::run(bf, af[mt], acc[mt][i]);
                            else run[mt][i] = Mfma<T>::run(bf, af[mt], run[mt][i]);
                        }
                        if (refill) {
#pragma unroll
                            for (int q = i; q < NREF; q += NMF)
                                issue_piece(XLINES ? q : (q < MT ? QP + s * MT + q : q - MT), kb + (t + D) * (32 * R), slot);
                        }
                    }
                }
            }
        }
        if (++slot_idx == D) { slot_idx = 0; slot = ring0; } else slot += slot_bytes;
    }
    if constexpr (!PRE) { if (cur_group >= 0) fold_run(cur_group); }
    FLUTE_STAMP(2);
#ifdef FLUTE_STAMPS
    stamp[7] = cyc_wait;                                   // shader cycles spent in dma_wait
    stamp[4] = __builtin_readcyclecounter() - cyc_loop0;   // shader cycles of the whole main loop (replaces stamp 4)
#endif

    // ---- epilogue: the kw partial tiles of a slab are summed through LDS by ALL its waves
    // (tile tt of the slab by wave tt % kw), then stored 8 B (16 B for split-K partials) per lane ----
    const int c0 = q4 * 4;                                           // first of this lane's 4 columns inside a tile
    const bool inl = a.splitk > 1 && a.state != nullptr;
    const __amdgpu_buffer_rsrc_t slabs = xwg_rsrc(a.partial, inl ? (uint32_t)((size_t)a.splitk * a.M * a.N * 4) : 0u);
    auto store_tile = [&](int mt, int i, const f32x4_t v) {
        const int row = m0 + mt * 16 + r16;
        if (row >= a.M) return;
        const int col = unit_col0<BITS, TILEP>((slab + i / NMFS) * SU + c0 % SU) + ((i % NMFS) * R + c0 / SU) * TILEP;
        if (a.splitk == 1) {
            uint2 o;
            o.x = (uint32_t)NT::from_float(v[0]) | ((uint32_t)NT::from_float(v[1]) << 16);
            o.y = (uint32_t)NT::from_float(v[2]) | ((uint32_t)NT::from_float(v[3]) << 16);
            *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(a.D) + (size_t)row * a.N + col) = o;
        } else if (inl) {
            xwg_store(v, slabs, (uint32_t)((((size_t)split * a.M + row) * a.N + col) * 4));     // write-through: combined below, in this launch
        } else {
            *reinterpret_cast<f32x4_t*>(a.partial + ((size_t)split * a.M + row) * a.N + col) = v;
        }
    };
    // the same tile of every K slice, summed in ascending slice order (round 4: the last workgroup to arrive at a
    // (slab group, row tile) does this for the tiles its waves stored - the reference's fix-up, tile_scheduler_utils.hpp:58-211,
    // without a second launch)
    auto combine_tile = [&](int mt, int i) {
        const int row = m0 + mt * 16 + r16;
        if (row >= a.M) return;
        const int col = unit_col0<BITS, TILEP>((slab + i / NMFS) * SU + c0 % SU) + ((i % NMFS) * R + c0 / SU) * TILEP;
        const uint32_t off = (uint32_t)(((size_t)row * a.N + col) * 4), slice = (uint32_t)((size_t)a.M * a.N * 4);
        f32x4_t v = f32x4_t{0.f, 0.f, 0.f, 0.f};
        for (int s0 = 0; s0 < a.splitk; s0 += 4) {
            f32x4_t ld[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) ld[j] = xwg_load(slabs, s0 + j < a.splitk ? (uint32_t)(s0 + j) * slice + off : 0xfffffff0u);
#pragma unroll
            for (int j = 0; j < 4; ++j) v += ld[j];                // past the last slice: out of range reads as zero
        }
        uint2 o;
        o.x = (uint32_t)NT::from_float(v[0]) | ((uint32_t)NT::from_float(v[1]) << 16);
        o.y = (uint32_t)NT::from_float(v[2]) | ((uint32_t)NT::from_float(v[3]) << 16);
        *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(a.D) + (size_t)row * a.N + col) = o;
    };
    if (kw == 1) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int i = 0; i < NMF; ++i) store_tile(mt, i, acc[mt][i]);
    } else {
        // at most 16 accumulator tiles (16 KB) per wave and pass: 8 waves x 16 KB is what LDS holds
        constexpr int NT_TILES = MT * NMF;
        constexpr int PASS = NT_TILES > 16 ? 16 : NT_TILES;
        __syncthreads();                          // every wave is done with the table and its ring
#pragma unroll
        for (int p0 = 0; p0 < NT_TILES; p0 += PASS) {
            if (p0 > 0) __syncthreads();          // the previous pass has been read
#pragma unroll
            for (int tl = 0; tl < PASS; ++tl) {
                const int tt = p0 + tl;           // tile (mt, i) = (tt / NMF, tt % NMF)
                *reinterpret_cast<f32x4_t*>(smem + ((size_t)(wave * PASS + tl) * 64 + lane) * 16) = acc[tt / NMF][tt % NMF];
            }
            __syncthreads();
            for (int tl = kpart; tl < PASS; tl += kw) {
                f32x4_t v = *reinterpret_cast<const f32x4_t*>(smem + ((size_t)((sl * kw) * PASS + tl) * 64 + lane) * 16);
                for (int kp = 1; kp < kw; ++kp)
                    v += *reinterpret_cast<const f32x4_t*>(smem + ((size_t)((sl * kw + kp) * PASS + tl) * 64 + lane) * 16);
                store_tile((p0 + tl) / NMF, (p0 + tl) % NMF, v);
            }
        }
    }
    if (inl) {
        // bid = this workgroup's (slab group, row tile) index: the K slices of it are neighbouring blocks
        xwg_word* st = xwg_state(a.state + 2 * (size_t)((int)blockIdx.x / a.splitk));
        const uint32_t before = xwg_arrive(st, 0u, tid);           // (LDS dword 0: the table is dead, the barrier inside retires the epilogue's reads)
        if (before == (uint32_t)(a.splitk - 1)) {
            if (kw == 1) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int i = 0; i < NMF; ++i) combine_tile(mt, i);
            } else {
                constexpr int NT_TILES = MT * NMF;
                constexpr int PASS = NT_TILES > 16 ? 16 : NT_TILES;
#pragma unroll
                for (int p0 = 0; p0 < NT_TILES; p0 += PASS)
                    for (int tl = kpart; tl < PASS; tl += kw) combine_tile((p0 + tl) / NMF, (p0 + tl) % NMF);
            }
            xwg_reset(st, tid);
        }
    }
#ifdef FLUTE_STAMPS
    __builtin_amdgcn_s_waitcnt(0);               // stores retired
    FLUTE_STAMP(3);
    if (lane == 0 && a.splitk == 1 && a.partial != nullptr) {
        uint64_t* o = reinterpret_cast<uint64_t*>(a.partial) + ((size_t)blockIdx.x * nw + wave) * 8;
        for (int i = 0; i < 8; ++i) o[i] = stamp[i];
    }
#endif
#undef FLUTE_STAMP
}

}  // namespace flute_amd
