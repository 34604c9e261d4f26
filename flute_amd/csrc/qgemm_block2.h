// Prefill kernel, second geometry: a wave owns ALL 256 rows of the workgroup's block and 32 of its 256 columns.
//
// qgemm_block.h splits the 256 x 256 block 2 x 4 over the waves: each weight is dequantised by two waves, and a
// half step carries 16 lookups + 16 multiplies for 32 MFMAs per wave; that kernel is bound by a wave's in-order
// issue stream (~145 instructions per half step), not by a pipe.  Here the block is split 1 x 8:
// a wave multiplies 16 row tiles by TWO column tiles = 8 units x 4 fields, so every weight of the block is
// dequantised exactly once (8 lookups + 8 multiplies per 32 MFMAs).  Lane (r16, q4) holds the words 4 q4 .. 4 q4 + 3
// of unit r16 % 8 and feeds MFMA weight row r16 of column tile t with field r16 / 8 + 2 t, i.e. lanes r16 and
// r16 + 8 load the same 16 B (one cache line less per request, not one more).
// Activation fragments: 16 row tiles per half step through EIGHT register slots - the fragment of row tile R + 8
// replaces row tile R's as soon as its two MFMAs are issued (hidden ds_read_b128 with immediate offsets, 8 row
// tiles = 16 MFMAs of lookahead, released by a counted lgkmcnt per row tile); the next half step's lookups and scales ride between the MFMAs of row tiles
// 8..15.  Two barriers per 64-k step: (A) at its start - every wave has retired its reads of stage t-1, batch t+2
// may overwrite it; (B) at mid step - every wave's batch t+1 has landed, stage t+1 may be read.
// Same arithmetic contract as qgemm_block.h: w^ = round_T(lut * s), fp32 accumulation, one output rounding.
// Round 4 (what qgemm_splitk.h showed about requests - the CU's addresser is paid per cache line a request touches):
// activation pieces are 8 rows x 128 B, whole lines (were 16 rows x 64 B), and a wave issues ONE whole-line weight request
// per step (U unit rows x 128 B; the half step a lane lacks comes from lane r16 ^ 8 by DPP row_ror:8) instead of two that
// touched 16 half lines each: 128-row blocks 72 -> 62.5 us per block at K = 4096, 256-row blocks 126.9 -> 123.0 us at
// M = 4096 on 4096^2 (profiles/r04/splitk_lab_run7_block2_whole_line_requests.jsonl).
#pragma once
#include "qgemm_block.h"

namespace flute_amd {

// RT = 16: 256-row blocks as described; RT = 8 / 4: 128- / 64-row blocks (every fragment lives in a slot, each replaced
// by the NEXT half step's fragment of the same row tile; smaller stages, for outputs with too few 256-row blocks).
// BITS = 2: a word holds 8 four-bit pair fields, so the wave's 32 columns are 4 units x 8 fields (lane r16: unit r16 % 4,
// field r16 / 4, + 4 for the second column tile) and the pair table has 16 entries; everything else is the same code.
template <typename T, int TILEP, int RT = 16, int BITS = 4>
__global__ __launch_bounds__(512) void qgemm_block2_kernel(const BlockArgs args) {
    using NT = Num<T>;
    static_assert(BITS == 4 || BITS == 2, "3-bit layers use the per-wave MFMA kernel (qgemm_tile.h)");
    constexpr int J = 16 / BITS;                                   // fields per word
    constexpr int U = 32 / J;                                      // units per wave (32 columns)
    constexpr int FPT = 16 / U;                                    // fields per column tile and unit: tile t uses field fsel + FPT t
    constexpr int FB = 2 * BITS;                                   // bits of a pair index
    // (RT = 4, 64-row blocks, works too but a block then takes 87 % of a 128-row block's time for half its work:
    // the per-step costs - 8 lookups per wave, the DMA queue, two barriers - do not shrink with the rows; not instantiated)
    static_assert(RT == 16 || RT == 8 || RT == 4, "row tiles per block");
    constexpr bool SPLIT = __is_same(T, BF16) && RT == 16;         // batch issue spread over both half steps (below)
    static_assert(RT * 2 / 8 + 2 <= RT, "the batch (PPW + 2 requests) is issued over the row tiles of a half step");
    constexpr int NS = RT < 8 ? RT : 8;                            // fragment slots
    constexpr int NW = 8, BM = RT * 16, NT2 = 2;                   // waves, rows, column tiles per wave
    constexpr int PIECES = RT * 2, PPW = PIECES / NW;
    constexpr int BATCH = PPW + 1 + 1;                             // X pieces, one weight piece, one scale block
    constexpr int LUT_BYTES = (1 << (2 * BITS)) * 128;
    constexpr int STAGE_BYTES = PIECES * 1024;
#ifdef FLUTE_B2_ABLATE   // development builds (tools/build_variant.sh): 1 no activation requests in the loop, 2 no weight / scale requests,
    constexpr int dbg = FLUTE_B2_ABLATE;                           // 4 no MFMA, 8 no table lookups, 16 no fragment reads, 32 no barriers
#else
    constexpr int dbg = 0;
#endif

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
    const int u8 = r16 % U;                                        // this lane's unit within the wave's U
    const int fsel = r16 / U;                                      // ... and field of column tile 0 (tile 1: + FPT)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    int bid = blockIdx.x, split = 0;
    if (a.splitk > 1) { split = bid % a.splitk; bid /= a.splitk; }
    int tm_idx, tn_idx;
    if (a.order == 1) {
        const int per = a.tiles_m >> 3, x = bid & 7, i = bid >> 3;
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
    const int unit0 = (tn_idx * NW + wave) * U;                    // this wave's U units
    const int kbeg = split * a.k_per_split;
    const int kend = min(a.K, kbeg + a.k_per_split);
    const int nsteps = (kend - kbeg) >> 6;
    const uint32_t row_bytes = (uint32_t)a.K * 2u;

    const srd_t x_srd = make_srd(a.A, (uint32_t)min((size_t)a.M * a.K * 2, (size_t)0xfffffff0u));
    const srd_t w_srd = make_srd(reinterpret_cast<const char*>(a.Q) + (size_t)unit0 * row_bytes, (uint32_t)U * row_bytes);
    const srd_t s_srd = make_srd(a.S, (uint32_t)min((size_t)a.N * a.G * 2, (size_t)0xfffffff0u));
    // Activations (round 4: WHOLE cache lines - a request is priced per line it touches; before: 16 rows x 64 B): piece
    // p = 2 rt + rh of a stage = rows 16 rt + 8 rh .. + 7, 128 B (the 64 k of the step) each; lane L fetches the 16-B chunk
    // (L & 7) ^ blk_swz8(L >> 3, rh) of row L >> 3, written lane-linearly.  This wave's pieces: wave * PPW + i.  Rows past M
    // need no flag: their byte offset is past the descriptor's range (voffset is what the range check covers) and reads
    // as zero; the K offset of a step travels in the scalar offset.
    constexpr int PH = PPW;
    const int p0 = wave * PPW;
    uint32_t x_vo[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int rt = (p0 + i) >> 1, rh = (p0 + i) & 1, row8 = lane >> 3;
        x_vo[i] = (uint32_t)(((size_t)(m0 + rt * 16 + rh * 8 + row8) * a.K + (((lane & 7) ^ blk_swz8(row8, rh)) * 8)) * 2);
    }
    const uint32_t x_lds0 = (uint32_t)LUT_BYTES + (uint32_t)p0 * 1024u;
    // Weights (round 4): ONE request per wave and step - U unit rows x 128 B, whole lines: lane (r16, q4) fetches chunk
    // 4 (r16 >> 3) + q4 of unit r16 % U; the half step a lane lacks comes from lane r16 ^ 8 (same unit) by DPP row_ror:8
    const uint32_t w_voff = (uint32_t)u8 * row_bytes + (uint32_t)((r16 >> 3) * 4 + q4) * 16u;
    // scale block: lane L < 32 fetches 8 groups of column (unit L % U, field L / U); the image is lane-linear
    const uint32_t s_voff = (lane < 32)
        ? (uint32_t)(((size_t)(unit_col0<BITS, TILEP>(unit0 + (lane % U)) + (lane / U) * TILEP) * a.G) * 2) : 0x80000000u;
    const uint32_t sc_base = (uint32_t)LUT_BYTES + BLK_STAGES * STAGE_BYTES + (uint32_t)wave * 3072u;
    const uint32_t sc_sink = sc_base + 2048u;

    u32x4_t w[BLK_STAGES];
    auto half_words = [&](const u32x4_t& own, auto h_tag) {         // the words of half step h out of a step's piece
        constexpr int h = decltype(h_tag)::value;
        u32x4_t r;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            r[j] = (uint32_t)__builtin_amdgcn_update_dpp((int)own[j], (int)own[j], 0x128 /* row_ror:8 */, 0xf, h == 0 ? 0xc : 0x3, false);
        return r;
    };
    // batch u = every hidden load of K step u.  Batches past the end (issued two steps ahead, never consumed)
    // re-read the last step.  No per-lane arithmetic: every K offset is wave-uniform and travels in the scalar offset.
    auto issue_batch = [&](auto slot_tag, int u) {
        constexpr int slot = decltype(slot_tag)::value;
        const uint32_t k0 = (uint32_t)(kbeg + min(u, nsteps - 1) * 64);
#pragma unroll
        for (int i = 0; i < PH; ++i)
            dma16_buf(x_vo[i], x_srd, k0 * 2u, x_lds0 + (uint32_t)i * 1024u + (uint32_t)slot * STAGE_BYTES);
        w[slot] = buf_load16(w_voff, w_srd, k0 * 2u);
        const int g = (int)(k0 >> a.lg);
        const bool blk_start = (u < nsteps) && ((g & 7) == 0 || u == 0) && ((k0 & ((1u << a.lg) - 1u)) == 0);
        // the block lands in this wave's image when the step starts one, else (same request) in the sink
        dma16_buf(s_voff, s_srd, (uint32_t)((g >> 3) * 16),
                  blk_start ? sc_base + (uint32_t)((g >> 3) & 1) * 1024u : sc_sink);
    };

    // the same batch, one request at a time (in the loop the BATCH requests ride between the MFMAs of the first row
    // tiles of half step 0 instead of queueing behind barrier (A) in all eight waves at once); same order as above
    auto issue_one = [&](auto slot_tag, auto i_tag, int u) {
        constexpr int slot = decltype(slot_tag)::value;
        constexpr int i = decltype(i_tag)::value;
        const uint32_t k0 = (uint32_t)(kbeg + min(u, nsteps - 1) * 64);
        if constexpr (i < PH) {
            if constexpr (!(dbg & 1)) dma16_buf(x_vo[i], x_srd, k0 * 2u, x_lds0 + (uint32_t)i * 1024u + (uint32_t)slot * STAGE_BYTES);
        } else if constexpr (dbg & 2) {
        } else if constexpr (i < PH + 1) {
            w[slot] = buf_load16(w_voff, w_srd, k0 * 2u);
        } else {
            const int g = (int)(k0 >> a.lg);
            const bool blk_start = (u < nsteps) && ((g & 7) == 0 || u == 0) && ((k0 & ((1u << a.lg) - 1u)) == 0);
            dma16_buf(s_voff, s_srd, (uint32_t)((g >> 3) * 16),
                      blk_start ? sc_base + (uint32_t)((g >> 3) & 1) * 1024u : sc_sink);
        }
    };

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
    // fragment of row tile R, half step h, stage slot: LUT + slot * STAGE_BYTES + R * 2048 + piece (r16 >> 3) * 1024 + row
    // (r16 & 7) * 128 + position ((4 h + q4) ^ swz) * 16: the lane part in a base register per half step (+ 64 KB for the
    // offsets past the 16-bit immediate), the rest immediate
    const uint32_t frag_l0 = (uint32_t)LUT_BYTES + (uint32_t)((r16 >> 3) * 1024 + (r16 & 7) * 128 + ((q4 ^ blk_swz8(r16 & 7, r16 >> 3)) * 16));
    const uint32_t frag_l1 = (uint32_t)LUT_BYTES + (uint32_t)((r16 >> 3) * 1024 + (r16 & 7) * 128 + (((4 + q4) ^ blk_swz8(r16 & 7, r16 >> 3)) * 16));
    const uint32_t frag_h0 = frag_l0 + 65536u, frag_h1 = frag_l1 + 65536u;
    const uint32_t sc_lane = sc_base + (uint32_t)(fsel * U + u8) * 16u;
    const uint32_t shift0 = (uint32_t)(fsel * FB);                 // bit offset of this lane's field, column tile 0

    f32x4_t acc[RT][NT2];
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int t = 0; t < NT2; ++t) acc[r][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    uint32_t v[8];                                                 // hidden lookups of the NEXT half step: [tile][word]
    u32x4_t af[NS];                                                // fragment slots (row tile R lives in slot R % 8)
    uint32_t scn[NT2];                                             // scales of the next half step

    auto scales = [&](int t, int h) {
        const int grp = (kbeg + t * 64 + h * 32) >> a.lg;
        const uint32_t sb = sc_lane + (uint32_t)((grp >> 3) & 1) * 1024u + (uint32_t)(grp & 7) * 2u;
        uint32_t& d0 = scn[0];
        uint32_t& d1 = scn[1];
        asm volatile("ds_read_u16 %0, %1" : "=v"(d0) : "v"(sb) : "memory");
        asm volatile("ds_read_u16 %0, %1 offset:256" : "=v"(d1) : "v"(sb) : "memory");     // field + FPT = 16 image lanes on
    };
    auto lookup = [&](const u32x4_t& qw, auto n_tag) {
        constexpr int n = decltype(n_tag)::value;                  // tile n / 4, word n % 4
        const uint32_t idx = __builtin_amdgcn_ubfe(qw[n & 3], shift0 + (uint32_t)(FB * FPT * (n >> 2)), (uint32_t)FB);
        if constexpr (dbg & 8) v[n] = idx; else v[n] = lds_lookup32((idx << 7) | lane_off);
    };
    auto frag = [&](auto slot_tag, auto h_tag, auto r_tag) {
        if constexpr (dbg & 16) return;
        constexpr int R = decltype(r_tag)::value;
        constexpr int off = decltype(slot_tag)::value * STAGE_BYTES + R * 2048;
        constexpr int hh = decltype(h_tag)::value;
        u32x4_t& dst = af[R & 7];
        const uint32_t addr = off < 65536 ? (hh == 0 ? frag_l0 : frag_l1) : (hh == 0 ? frag_h0 : frag_h1);
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off < 65536 ? off : off - 65536) : "memory");
    };
    auto wait_lds = [&]() {
        if constexpr (NS == 8) {
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]),
                           "+v"(af[0]), "+v"(af[1]), "+v"(af[2]), "+v"(af[3]), "+v"(af[4 % NS]), "+v"(af[5 % NS]),
                           "+v"(af[6 % NS]), "+v"(af[7 % NS]), "+v"(scn[0]), "+v"(scn[1])
                         : : "memory");
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]),
                           "+v"(af[0]), "+v"(af[1]), "+v"(af[2]), "+v"(af[3]), "+v"(scn[0]), "+v"(scn[1])
                         : : "memory");
        }
    };

    auto half = [&](auto slot_tag, auto h_tag, int t) {
        constexpr int slot = decltype(slot_tag)::value;
        constexpr int h = decltype(h_tag)::value;
        constexpr int nslot = h ? (slot + 1) % BLK_STAGES : slot;
        constexpr int nh = h ^ 1;
        wait_lds();
        if constexpr (h == 0) {
            if constexpr (!(dbg & 32)) __builtin_amdgcn_s_barrier();   // (A) stage t-1 is free: batch t+2 follows, spread over the rows
        } else {
            // (B) batch t+1 has landed once at most batch t+2 is outstanding (SPLIT: only its PH activation pieces have
            // been issued by now; its weight / scale requests follow during this half step)
            constexpr int NB = ((dbg & 1) ? 0 : PH) + ((dbg & 2) ? 0 : 2);
            asm volatile("s_waitcnt vmcnt(%1)" : "+v"(w[nslot]) : "n"(SPLIT ? ((dbg & 1) ? 0 : PH) : NB) : "memory");
            if constexpr (!(dbg & 32)) __builtin_amdgcn_s_barrier();
        }
        u32x4_t bf[NT2];
#pragma unroll
        for (int c = 0; c < NT2; ++c) {
            const uint32_t sj = scn[c];
            const uint32_t vin[4] = {v[c * 4], v[c * 4 + 1], v[c * 4 + 2], v[c * 4 + 3]};
            uint32_t o[4];
            NT::mul_scale4(vin, sj, o);
            bf[c] = u32x4_t{o[0], o[1], o[2], o[3]};
        }
        scales(t + h, nh);
        const u32x4_t qw = half_words(w[nslot], std::integral_constant<int, nh>{});
        auto row = [&](auto r_tag) {
            constexpr int R = decltype(r_tag)::value;
            if constexpr (R >= 8) {
                // row tile R's fragment was requested after row tile R-8's MFMAs of THIS half step.  LDS returns in
                // order; younger than it: the fragments of row tiles R+1..15 and the two requests (fragment, lookup)
                // each of row tiles 8..R-1 = 7 + (R - 8) operations
                u32x4_t& slot_reg = af[R & 7];
                asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(slot_reg) : "n"(7 + (R - 8)) : "memory");
            }
#pragma unroll
            for (int c = 0; c < NT2; ++c) {
                if constexpr (dbg & 4) acc[R][c][0] += __builtin_bit_cast(float, bf[c][0] ^ af[R & 7][0]);
                else acc[R][c] = Mfma<T>::run(bf[c], af[R & 7], acc[R][c]);
            }
            // batch t+2 rides between the MFMAs, one request per row tile (eight waves issuing a whole batch at once
            // behind barrier (A) queue on the texture path and stall in order: 108.7 -> 99.9 us per 256-row block).
            // SPLIT (bf16 256-row blocks, whose multiply stage is 3x longer): the PH activation pieces on every
            // second row tile of half step 0, the weight pieces and the scale block in half step 1 (116 -> 104 us;
            // fp16 and the 128-row blocks measured slower that way)
            if constexpr (SPLIT) {
                if constexpr (h == 0 && (R & 1) == 0 && R / 2 < PH)
                    issue_one(std::integral_constant<int, (slot + 2) % BLK_STAGES>{}, std::integral_constant<int, R / 2>{}, t + 2);
                if constexpr (h == 1 && (R & 1) == 0 && R / 2 < 2)
                    issue_one(std::integral_constant<int, (slot + 2) % BLK_STAGES>{}, std::integral_constant<int, PH + R / 2>{}, t + 2);
            } else {
                if constexpr (h == 0 && R < BATCH)
                    issue_one(std::integral_constant<int, (slot + 2) % BLK_STAGES>{}, r_tag, t + 2);
            }
            if constexpr (RT == 16 && R < 8) {
                frag(slot_tag, h_tag, std::integral_constant<int, R + 8>{});
            } else {
                frag(std::integral_constant<int, nslot>{}, std::integral_constant<int, nh>{}, std::integral_constant<int, R & 7>{});
                if constexpr (RT == 4) {                           // 8 lookups over 4 row tiles
                    lookup(qw, std::integral_constant<int, 2 * R>{});
                    lookup(qw, std::integral_constant<int, 2 * R + 1>{});
                } else {
                    lookup(qw, std::integral_constant<int, R & 7>{});
                }
            }
        };
        [&]<int... R>(std::integer_sequence<int, R...>) {
            (row(std::integral_constant<int, R>{}), ...);
        }(std::make_integer_sequence<int, RT>{});
    };

    // batch 0 and the pair table before anyone reads them
    asm volatile("s_waitcnt vmcnt(%1)" : "+v"(w[0]) : "n"(BATCH) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    scales(0, 0);
    {
        const u32x4_t qw = half_words(w[0], std::integral_constant<int, 0>{});
        [&]<int... R>(std::integer_sequence<int, R...>) {
            (frag(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, R>{}), ...);
        }(std::make_integer_sequence<int, NS>{});
        [&]<int... L>(std::integer_sequence<int, L...>) {
            (lookup(qw, std::integral_constant<int, L>{}), ...);
        }(std::make_integer_sequence<int, 8>{});
    }
    // The ring slots are compile-time, so the step loop is unrolled by three - and left after the LAST step, not
    // after a whole triple (K = 4096 = 64 steps would otherwise run 66).  What is in flight at the exit (the
    // clamped batches nsteps, nsteps + 1 and the prefetch for step nsteps) is dead and drained below.
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
    wait_lds();                                                    // the prefetch past the end
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]) : : "memory");

    // ---- epilogue: accumulator register i of lane (r16, q4) = weight row 4 q4 + i of the column tile = unit
    // (4 q4 + i) % 8, field q4 / 2 + 2 t: four consecutive columns; the lane's output row is r16 ----
    const int c_unit = unit0 + (4 * q4) % U;
#pragma unroll
    for (int r = 0; r < RT; ++r) {
        const int row = m0 + r * 16 + r16;
        if (row < a.M) {
#pragma unroll
            for (int t = 0; t < NT2; ++t) {
                const int col = unit_col0<BITS, TILEP>(c_unit) + ((4 * q4) / U + FPT * t) * TILEP;
                const f32x4_t o4 = acc[r][t];
                if (a.splitk == 1) {
                    uint2 o;
                    o.x = (uint32_t)NT::from_float(o4[0]) | ((uint32_t)NT::from_float(o4[1]) << 16);
                    o.y = (uint32_t)NT::from_float(o4[2]) | ((uint32_t)NT::from_float(o4[3]) << 16);
                    *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(a.D) + (size_t)row * a.N + col) = o;
                } else {
                    *reinterpret_cast<f32x4_t*>(a.partial + ((size_t)split * a.M + row) * a.N + col) = o4;
                }
            }
        }
    }
}

}  // namespace flute_amd
