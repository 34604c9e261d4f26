// Prefill kernel for 3-bit layers: the 1 x 8 block geometry of qgemm_block2.h on the three-plane layout.
//
// A 3-bit unit is 16 columns (column = unit_col0 + 32 j, j = 0..15) whose pair indices live in three bit planes:
// field j < 15 sits in plane j % 3 at bit 6 (j / 3); field 15 takes the top two bits of all three planes
// (common.h `field<3>`).  Giving a wave "its" columns as units x fields the 4-bit way would make the plane and the
// shift lane-dependent (and field 15 a 5-instruction special case in every lane).  Instead the MFMA weight row is
// the UNIT: a workgroup owns 16 units x 16 fields = 256 columns, lane (r16, q4) holds words 4 q4 .. 4 q4 + 3 of
// unit r16, and every wave multiplies two column tiles = two FIELDS of those 16 units.  Plane and shift are then
// wave-uniform scalars (one v_bfe_u32 per lookup, as for 4 bits) and the accumulator of lane (r16, q4) holds four
// consecutive units = four consecutive columns (8-B stores).  The fields are dealt to the waves so that a wave's
// two fields share a plane wherever possible - (0,3) (6,9) (1,4) (7,10) (2,5) (8,11): ONE 16-B piece per lane and
// half step feeds both tiles, as for 4 bits - wave 6 takes (12,13) (two planes) and wave 7 takes (14,15): field 15
// needs all three planes (5 VALU per lookup), one of which is field 14's.  Three instantiations of the body, chosen
// by a wave-uniform branch; all execute the same barriers.
// Everything else is qgemm_block2.h (round 4: activations as whole-line 8-row x 128-B pieces too: 3-bit M = 4096 on
// 4096^2 130.0 -> 127.9 us, M = 1024 on 28672 x 8192 478 -> 466; the PLANE pieces stay 16 rows x 64 B - whole-line
// plane pieces were measured in round 5 and dropped, below): RT = 8 (128-row blocks) or, round 3, RT = 16 (256-row blocks).  Wave 7's ring of
// three plane pieces per half step x three stages is 72 registers - with 16 row tiles of accumulators (128) and every
// wave of a kernel getting the same allocation, too many - so at RT = 16 the SECOND and THIRD plane pieces of waves 6
// and 7 go to wave-private LDS by LDS-DMA (18 KB) and are read back, one half step ahead, with the fragments (246
// registers, no scratch; 3-bit M = 4096 on 4096^2: 161.6 -> 127.1 us fp16, 172.5 -> 130.7 bf16):
// stages, fragment slots, two barriers per step, requests riding between the MFMAs, exact step count.
// Arithmetic: w^ = round_T(lut * s) (packbits_utils.hpp:139), fp32 accumulation, one rounding of the output.
// The per-wave MFMA kernel (qgemm_tile.h) ran these layers at 360-380 TFLOP/s (M = 4096).
#pragma once
#include "qgemm_block2.h"
#include "xwg.h"

// Measured and dropped (round 5, profiles/r05/call24_w3_inlaunch_and_line_planes.log): the bit-plane pieces as WHOLE cache lines
// (two requests per step of 8 unit rows x 128 B, the words of a half step put together by one DPP row_ror:8 move per dword - the
// development variant FLUTE_B3_LINE_PLANES written at the end of round 4): M = 4096 on 4096^2 129.0 / 128.6 us against 127.3,
// M = 1024 on 28672 x 8192 473 against 467 - the extra VALU (16 .. 44 per step) cost what the halved line count saved.
namespace flute_amd {

template <typename T, int RT>
__global__ __launch_bounds__(512) void qgemm_block3_kernel(const BlockArgs args) {
    using NT = Num<T>;
    constexpr int BITS = 3, TILEP = 32;
    // RT = 8: 128-row blocks (the weight ring leaves no registers for 16 row tiles).  RT = 1, 2, 4: "skinny" blocks of
    // 16 / 32 / 64 rows for small batches, launched with a grid-level K split (fp32 slabs + splitk_reduce_kernel): a
    // 3-bit wave of the per-wave kernel is 256 columns wide, so a 4096-wide layer gives it 16 column slabs; here the
    // same layer gives 16 column blocks x the K split, every weight looked up once.
    static_assert(RT == 16 || RT == 8 || RT == 4 || RT == 2 || RT == 1, "row tiles per block");
    constexpr int NW = 8, BM = RT * 16, NT2 = 2;
    constexpr int PIECES = RT * 2;                                 // 1-KB activation pieces per stage
    constexpr int PH = (PIECES + NW - 1) / NW;                     // ... requested by a wave (skinny blocks: one, some of them idle)
    constexpr int NS = RT < 8 ? RT : 8;                            // fragment slots (RT = 16: row tile R lives in slot R % 8, as qgemm_block2.h)
    constexpr int LPR = RT >= 8 ? 1 : 8 / RT;                      // lookups issued after a row tile (RT = 16: after row tiles 8 .. 15)
    // stages = ring slots: batch t + NST - 1 is requested during step t.  (Six stages for the skinny blocks measured
    // SLOWER than three - 4096^2 M = 16: 18.5 vs 15.8 us: they are not latency-bound; what they pay is the fixed part,
    // prologue + fp32 slabs + the reduce launch, ~8 us of a 16-us call.)
    constexpr int NST = BLK_STAGES;
    constexpr int LUT_BYTES = 64 * 128;
    constexpr int STAGE_BYTES = PIECES * 1024;

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
    const int unit0 = tn_idx * 16;                                 // the workgroup's 16 units (lane r16 <-> unit)
    const int kbeg = split * a.k_per_split;
    const int kend = min(a.K, kbeg + a.k_per_split);
    const int nsteps = (kend - kbeg) >> 6;
    const uint32_t row_bytes = (uint32_t)a.K * 2u;

    const srd_t x_srd = make_srd(a.A, (uint32_t)min((size_t)a.M * a.K * 2, (size_t)0xfffffff0u));
    const srd_t w_srd = make_srd(a.Q, (uint32_t)min((size_t)(3 * (a.N >> 4)) * row_bytes, (size_t)0xfffffff0u));
    const srd_t s_srd = make_srd(a.S, (uint32_t)min((size_t)a.N * a.G * 2, (size_t)0xfffffff0u));
    // Activations (round 4, as qgemm_block2.h: WHOLE cache lines - a request is priced per line it touches; before: 16 rows
    // x 64 B): piece p = 2 rt + rh of a stage = rows 16 rt + 8 rh .. + 7, 128 B (the 64 k of the step) each; lane L fetches
    // the 16-B chunk (L & 7) ^ blk_swz8(L >> 3, rh) of row L >> 3, written lane-linearly.  Rows past M read as zero (their
    // byte offset is past the descriptor's range).
    const int p0 = wave * PH;
    const bool x_mine = p0 < PIECES;                               // (wave-uniform) skinny blocks have fewer pieces than waves
    uint32_t x_vo[PH];
#pragma unroll
    for (int i = 0; i < PH; ++i) {
        const int rt = (p0 + i) >> 1, rh = (p0 + i) & 1, row8 = lane >> 3;
        x_vo[i] = x_mine ? (uint32_t)(((size_t)(m0 + rt * 16 + rh * 8 + row8) * a.K + (((lane & 7) ^ blk_swz8(row8, rh)) * 8)) * 2) : 0x80000000u;
    }
    // plane rows of this lane's unit (common.h unit_row<3>): plane 0 = row u, planes 1 / 2 = 32 rows apart
    const int u = unit0 + r16;
    const uint32_t wv_p0 = (uint32_t)u * row_bytes + (uint32_t)q4 * 16u;
    const uint32_t wv_p1 = (uint32_t)((a.N >> 4) + (u >> 5) * 64 + (u & 31)) * row_bytes + (uint32_t)q4 * 16u;
    const uint32_t wv_dp = 32u * row_bytes;
    const uint32_t sc_base = (uint32_t)LUT_BYTES + NST * STAGE_BYTES + (uint32_t)wave * 3072u;
    const uint32_t sc_sink = sc_base + 2048u;
    // RT = 16 (256-row blocks): the SECOND and THIRD plane pieces of waves 6 / 7 go to wave-private LDS (LDS-DMA) instead of
    // the register ring - 16 row tiles of accumulators leave no registers for three planes x three stages x two half steps
    const uint32_t pl_base = (uint32_t)LUT_BYTES + NST * STAGE_BYTES + NW * 3072u + (wave == 7 ? 6u * 1024u : 0u);   // wave-uniform (M0 of the DMA)
    const uint32_t pl_lane = pl_base + (uint32_t)lane * 16u;
    const uint32_t x_lds0 = x_mine ? (uint32_t)LUT_BYTES + (uint32_t)p0 * 1024u : sc_sink;     // idle request: zeros into the sink

    {   // pair table: 64 entries, 32 copies each (128-B stride)
        for (int p = tid; p < 64 * 8; p += NW * 64) {
            const uint32_t v = a.QM2[p >> 3];
            *reinterpret_cast<uint4*>(smem + (size_t)(p >> 3) * 128 + (p & 7) * 16) = make_uint4(v, v, v, v);
        }
    }
    const uint32_t lane_off = (uint32_t)(lane & 31) * 4u;
    // fragment of row tile R, half step h, stage slot: LUT + slot * STAGE_BYTES + R * 2048 + piece (r16 >> 3) * 1024 + row
    // (r16 & 7) * 128 + position ((4 h + q4) ^ swz) * 16: half step 1 is 64 B away from half step 0, in the direction swz says
    const uint32_t frag_l0 = (uint32_t)LUT_BYTES + (uint32_t)((r16 >> 3) * 1024 + (r16 & 7) * 128 + ((q4 ^ blk_swz8(r16 & 7, r16 >> 3)) * 16));
    const uint32_t frag_l1 = frag_l0 ^ 64u;                        // (LUT_BYTES and the rest are multiples of 128)
    const uint32_t sc_lane = sc_base + (uint32_t)r16 * 16u;

    // ---- the body: three instantiations (see the header), the same barriers in each ----
    auto body = [&](auto kind_tag) {
        // KIND 0: both fields in one plane (one piece per half step); 1: two planes; 2: fields 14 and 15 (three planes)
        constexpr int KIND = decltype(kind_tag)::value;
        constexpr bool LAST = KIND == 2;
        constexpr int NPL = KIND + 1;                              // weight pieces per half step
        constexpr int BATCH = PH + 2 * NPL + 1;
        constexpr int RPR = (BATCH + RT - 1) / RT;                 // requests issued after every row tile of half step 0
        // tile t = field f_t: plane, bit offset (wave-uniform), this lane's row offset of that plane
        const int f0 = (wave < 6) ? (wave >> 1) + 6 * (wave & 1) : 2 * wave;          // 0 6 1 7 2 8 | 12 | 14
        const int f1 = (wave < 6) ? f0 + 3 : f0 + 1;                                   // 3 9 4 10 5 11 | 13 | 15
        const uint32_t sh0 = (uint32_t)(6 * (f0 / 3)), sh1 = (uint32_t)(6 * (f1 / 3));
        const int pl0 = f0 % 3, pl1 = f1 % 3;
        const uint32_t wv0 = (pl0 == 0) ? wv_p0 : wv_p1 + (uint32_t)(pl0 - 1) * wv_dp;
        const uint32_t wv1 = (pl1 == 0) ? wv_p0 : wv_p1 + (uint32_t)(pl1 - 1) * wv_dp;
        // scale block: lane L < 32 fetches 8 groups of column (unit L % 16, field f_(L / 16)); lane-linear image
        const uint32_t s_voff = (lane < 32)
            ? (uint32_t)(((size_t)(unit_col0<BITS, TILEP>(unit0 + (lane & 15)) + ((lane >> 4) ? f1 : f0) * TILEP) * a.G) * 2) : 0x80000000u;

        static_assert((NST - 2) * BATCH <= 63, "vmcnt is six bits");
        constexpr bool XLDS = RT == 16 && NPL > 1;                 // planes 1 .. NPL-1 through LDS
        constexpr int NREG = XLDS ? 1 : NPL;                       // planes in the register ring
        constexpr int NX = NPL - NREG;
        u32x4_t w[NST][2][NREG];
        u32x4_t pw[NX > 0 ? NX : 1];                               // the LDS planes of the NEXT half step
        const uint32_t pl_lane_k = pl_lane;
        auto issue_one = [&](auto slot_tag, auto i_tag, int ustep) {
            constexpr int slot = decltype(slot_tag)::value;
            constexpr int i = decltype(i_tag)::value;
            const uint32_t k0 = (uint32_t)(kbeg + min(ustep, nsteps - 1) * 64);
            if constexpr (i < PH) {
                dma16_buf(x_vo[i], x_srd, k0 * 2u,
                          x_lds0 + (x_mine ? (uint32_t)i * 1024u + (uint32_t)slot * STAGE_BYTES : 0u));
            } else if constexpr (i < PH + 2 * NPL) {
                constexpr int h = (i - PH) / NPL, c = (i - PH) % NPL;
                uint32_t vo;
                if constexpr (LAST) vo = (c == 0) ? wv_p0 : wv_p1 + (uint32_t)(c - 1) * wv_dp;     // planes 0, 1, 2
                else vo = (c == 0) ? wv0 : wv1;
                const uint32_t so = k0 * 2u + (uint32_t)h * 64u;
                if constexpr (XLDS && c >= 1)
                    dma16_buf(vo, w_srd, so, pl_base + (uint32_t)(((slot * 2 + h) * NX + (c - 1)) * 1024));
                else
                    w[slot][h][c] = buf_load16(vo, w_srd, so);
            } else {
                const int g = (int)(k0 >> a.lg);
                const bool blk_start = (ustep < nsteps) && ((g & 7) == 0 || ustep == 0) && ((k0 & ((1u << a.lg) - 1u)) == 0);
                dma16_buf(s_voff, s_srd, (uint32_t)((g >> 3) * 16),
                          blk_start ? sc_base + (uint32_t)((g >> 3) & 1) * 1024u : sc_sink);
            }
        };
        auto issue_batch = [&](auto slot_tag, int ustep) {
            [&]<int... I>(std::integer_sequence<int, I...>) {
                (issue_one(slot_tag, std::integral_constant<int, I>{}, ustep), ...);
            }(std::make_integer_sequence<int, BATCH>{});
        };
        auto wait_batch = [&](auto slot_tag, auto n_tag) {         // releases ring slot `slot` once <= n requests are outstanding
            constexpr int n = decltype(n_tag)::value;
            auto& ws = w[decltype(slot_tag)::value];               // (named first: clang does not capture through asm operands)
            if constexpr (NREG == 3)
                asm volatile("s_waitcnt vmcnt(%6)"
                             : "+v"(ws[0][0]), "+v"(ws[0][1]), "+v"(ws[0][2]), "+v"(ws[1][0]), "+v"(ws[1][1]), "+v"(ws[1][2])
                             : "n"(n) : "memory");
            else if constexpr (NREG == 2)
                asm volatile("s_waitcnt vmcnt(%4)"
                             : "+v"(ws[0][0]), "+v"(ws[0][1]), "+v"(ws[1][0]), "+v"(ws[1][1])
                             : "n"(n) : "memory");
            else
                asm volatile("s_waitcnt vmcnt(%2)" : "+v"(ws[0][0]), "+v"(ws[1][0]) : "n"(n) : "memory");
        };

        [&]<int... I>(std::integer_sequence<int, I...>) {            // batches 0 .. NST-2
            (issue_batch(std::integral_constant<int, I>{}, I), ...);
        }(std::make_integer_sequence<int, NST - 1>{});

        f32x4_t acc[RT][NT2];
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
            for (int t = 0; t < NT2; ++t) acc[r][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        uint32_t v[8];                                             // hidden lookups of the NEXT half step: [tile][word]
        u32x4_t af[NS];                                            // fragment slots (row tile R in slot R % 8)
        uint32_t scn[NT2];

        auto scales = [&](int t, int h) {
            const int grp = (kbeg + t * 64 + h * 32) >> a.lg;
            const uint32_t sb = sc_lane + (uint32_t)((grp >> 3) & 1) * 1024u + (uint32_t)(grp & 7) * 2u;
            uint32_t& d0 = scn[0];
            uint32_t& d1 = scn[1];
            asm volatile("ds_read_u16 %0, %1" : "=v"(d0) : "v"(sb) : "memory");
            asm volatile("ds_read_u16 %0, %1 offset:256" : "=v"(d1) : "v"(sb) : "memory");
        };
        // plane c of (slot, half): a ring register, or (XLDS, c >= 1) the register its LDS copy was read into
        auto lookup = [&](auto slot_tag, auto h_tag, auto n_tag) {
            constexpr int n = decltype(n_tag)::value;              // tile n / 4, word n % 4
            constexpr int ww = n & 3;
            constexpr int S_ = decltype(slot_tag)::value, H_ = decltype(h_tag)::value;
            // word ww of register plane c for this lane's (unit, half step)
            auto ring_word = [&](auto c_tag) -> uint32_t { return w[S_][H_][decltype(c_tag)::value][ww]; };
            const uint32_t q0w = ring_word(std::integral_constant<int, 0>{});
            const uint32_t q1w = XLDS ? pw[0][ww] : ring_word(std::integral_constant<int, (NREG > 1 ? 1 : 0)>{});
            const uint32_t q2w = XLDS ? pw[NX > 1 ? 1 : 0][ww] : ring_word(std::integral_constant<int, (NREG > 2 ? 2 : 0)>{});
            uint32_t idx;
            if constexpr (LAST) {
                if constexpr (n < 4) idx = __builtin_amdgcn_ubfe(q2w, 24u, 6u);                    // field 14: plane 2, bit 24
                else idx = (q0w >> 30) | ((q1w >> 28) & 0xcu) | ((q2w >> 26) & 0x30u);              // field 15 (common.h field<3>)
            } else if constexpr (n < 4) {
                idx = __builtin_amdgcn_ubfe(q0w, sh0, 6u);
            } else {
                idx = __builtin_amdgcn_ubfe(KIND == 0 ? q0w : q1w, sh1, 6u);
            }
            v[n] = lds_lookup32((idx << 7) | lane_off);
        };
        auto planes = [&](auto slot_tag, auto h_tag) {            // hidden reads of the LDS planes of (slot, half)
            if constexpr (XLDS) {
                [&]<int... C>(std::integer_sequence<int, C...>) {
                    (([&] {
                        u32x4_t& dst = pw[C];
                        const uint32_t addr = pl_lane_k;
                        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr),
                                     "n"(((decltype(slot_tag)::value * 2 + decltype(h_tag)::value) * NX + C) * 1024) : "memory");
                    }()), ...);
                }(std::make_integer_sequence<int, NX>{});
            }
        };
        auto frag = [&](auto slot_tag, auto h_tag, auto r_tag) {
            constexpr int R = decltype(r_tag)::value;
            constexpr int off = decltype(slot_tag)::value * STAGE_BYTES + R * 2048;
            constexpr int hh = decltype(h_tag)::value;
            u32x4_t& dst = af[R & 7];
            const uint32_t fl = hh == 0 ? frag_l0 : frag_l1;
            const uint32_t addr = off < 65536 ? fl : fl + 65536u;     // (named first: clang does not capture through asm operands)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off < 65536 ? off : off - 65536) : "memory");
        };
        auto wait_lds = [&]() {
            if constexpr (RT >= 8)
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]),
                               "+v"(af[0]), "+v"(af[1 % NS]), "+v"(af[2 % NS]), "+v"(af[3 % NS]), "+v"(af[4 % NS]), "+v"(af[5 % NS]),
                               "+v"(af[6 % NS]), "+v"(af[7 % NS]), "+v"(scn[0]), "+v"(scn[1])
                             : : "memory");
            else if constexpr (RT == 4)
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]),
                               "+v"(af[0]), "+v"(af[1 % RT]), "+v"(af[2 % RT]), "+v"(af[3 % RT]), "+v"(scn[0]), "+v"(scn[1])
                             : : "memory");
            else if constexpr (RT == 2)
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]),
                               "+v"(af[0]), "+v"(af[1 % RT]), "+v"(scn[0]), "+v"(scn[1])
                             : : "memory");
            else
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]),
                               "+v"(af[0]), "+v"(scn[0]), "+v"(scn[1])
                             : : "memory");
        };

        auto half = [&](auto slot_tag, auto h_tag, int t) {
            constexpr int slot = decltype(slot_tag)::value;
            constexpr int h = decltype(h_tag)::value;
            constexpr int nslot = h ? (slot + 1) % NST : slot;
            constexpr int nh = h ^ 1;
            wait_lds();
            // (B) batch t+1 has landed once at most batches t+2 .. t+NST-1 are outstanding
            if constexpr (h == 1) wait_batch(std::integral_constant<int, nslot>{}, std::integral_constant<int, (NST - 2) * BATCH>{});
            // (round 5, as qgemm_splitk.h) what touches only the wave's own registers, scale image and LDS planes runs BEFORE the workgroup meets:
            // 128-row blocks x 2 K slices, M = 1024 on 4096^2 bf16: 53.4 -> 50.9 us; 256-row blocks: equal (same-box A/B, profiles/r05/call35_*.log;
            // the 2- / 4-bit blocks of qgemm_block2.h measured 1 % SLOWER that way in fp16 and keep their order)
            u32x4_t bf[NT2];
#pragma unroll
            for (int c = 0; c < NT2; ++c) {
                const uint32_t vin[4] = {v[c * 4], v[c * 4 + 1], v[c * 4 + 2], v[c * 4 + 3]};
                uint32_t o[4];
                NT::mul_scale4(vin, scn[c], o);
                bf[c] = u32x4_t{o[0], o[1], o[2], o[3]};
            }
            scales(t + h, nh);
            planes(std::integral_constant<int, nslot>{}, std::integral_constant<int, nh>{});
            asm volatile("" : "+v"(bf[0]), "+v"(bf[1]) : : "memory");     // (keeps hipcc from sinking the multiplies below the barrier)
            // (A) [h = 0] stage t-1 is free: batch t+NST-1 follows, spread over the rows; (B) [h = 1] stage t+1 is complete
            __builtin_amdgcn_s_barrier();
            auto row = [&](auto r_tag) {
                constexpr int R = decltype(r_tag)::value;
                if constexpr (R >= 8) {
                    // (RT = 16, as qgemm_block2.h) row tile R's fragment was requested after row tile R-8's MFMAs of THIS half
                    // step; younger than it: the fragments of row tiles R+1..15 and (fragment, lookup) of row tiles 8..R-1.
                    // The LDS planes were read before row tile 0: they have returned with it
                    u32x4_t& slot_reg = af[R & 7];
                    u32x4_t& p0 = pw[0];
                    u32x4_t& p1 = pw[NX > 1 ? 1 : 0];                // (an operand list must not name one register twice)
                    if constexpr (NX == 2) asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(slot_reg), "+v"(p0), "+v"(p1) : "n"(7 + (R - 8)) : "memory");
                    else if constexpr (NX == 1) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(slot_reg), "+v"(p0) : "n"(7 + (R - 8)) : "memory");
                    else asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(slot_reg) : "n"(7 + (R - 8)) : "memory");
                }
#pragma unroll
                for (int c = 0; c < NT2; ++c) acc[R][c] = Mfma<T>::run(bf[c], af[R & 7], acc[R][c]);
                if constexpr (h == 0)                              // batch t+NST-1: RPR requests after every row tile
                    [&]<int... Q>(std::integer_sequence<int, Q...>) {
                        (([&] {
                            if constexpr (R * RPR + Q < BATCH)
                                issue_one(std::integral_constant<int, (slot + NST - 1) % NST>{},
                                          std::integral_constant<int, R * RPR + Q>{}, t + NST - 1);
                        }()), ...);
                    }(std::make_integer_sequence<int, RPR>{});
                if constexpr (RT == 16 && R < 8) {
                    frag(slot_tag, h_tag, std::integral_constant<int, R + 8>{});
                } else {
                    frag(std::integral_constant<int, nslot>{}, std::integral_constant<int, nh>{}, std::integral_constant<int, R & 7>{});
                    [&]<int... L>(std::integer_sequence<int, L...>) {
                        (lookup(std::integral_constant<int, nslot>{}, std::integral_constant<int, nh>{}, std::integral_constant<int, (R & 7) * LPR + L>{}), ...);
                    }(std::make_integer_sequence<int, LPR>{});
                }
            };
            [&]<int... R>(std::integer_sequence<int, R...>) {
                (row(std::integral_constant<int, R>{}), ...);
            }(std::make_integer_sequence<int, RT>{});
        };

        // batch 0 and the pair table before anyone reads them
        wait_batch(std::integral_constant<int, 0>{}, std::integral_constant<int, (NST - 2) * BATCH>{});
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        scales(0, 0);
        planes(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
        if constexpr (XLDS) {
            u32x4_t& p0 = pw[0];
            u32x4_t& p1 = pw[NX > 1 ? 1 : 0];
            if constexpr (NX == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(p0), "+v"(p1), "+v"(scn[0]), "+v"(scn[1]) : : "memory");
            else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(p0), "+v"(scn[0]), "+v"(scn[1]) : : "memory");
        }
        {
            [&]<int... R>(std::integer_sequence<int, R...>) {
                (frag(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, R>{}), ...);
            }(std::make_integer_sequence<int, NS>{});
            [&]<int... L>(std::integer_sequence<int, L...>) {
                (lookup(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, L>{}), ...);
            }(std::make_integer_sequence<int, 8>{});
        }
        auto step = [&](auto slot_tag, int t) {
            half(slot_tag, std::integral_constant<int, 0>{}, t);
            half(slot_tag, std::integral_constant<int, 1>{}, t);
        };
        // unrolled by the ring (the slots are compile-time), left after the LAST step
        auto round = [&]<int... I>(std::integer_sequence<int, I...>, int t0) -> bool {
            bool done = false;
            ((done = done || (step(std::integral_constant<int, I>{}, t0 + I), t0 + I + 1 >= nsteps)), ...);
            return done;
        };
        for (int t0 = 0; !round(std::make_integer_sequence<int, NST>{}, t0); t0 += NST) {}
        wait_lds();                                                // the prefetch past the end
        [&]<int... I>(std::integer_sequence<int, I...>) {
            (wait_batch(std::integral_constant<int, I>{}, std::integral_constant<int, 0>{}), ...);
        }(std::make_integer_sequence<int, NST>{});

        // ---- epilogue: accumulator register i of lane (r16, q4) = unit 4 q4 + i of the workgroup, i.e. four
        // consecutive columns of field f_t; the lane's output row is r16 ----
        if constexpr (RT == 8) {
            // 128-row blocks x 2 / 4 K slices combined INSIDE the launch (round 5, xwg.h, E form): the reference's Stream-K fix-up
            // for any bit width (tile_scheduler_utils.hpp:58-211, :460-481).  Until round 4: [M][N] fp32 slabs and a reduce launch
            // - kept (a.state == nullptr) for the 256-row blocks and for the skinny blocks, whose 4 - 8 slices of 16 .. 64 rows
            // the reduce pass sums faster than their last arrivers would (measured, api.hip).
            if (a.splitk > 1 && a.state != nullptr) {
                const uint32_t col0 = (uint32_t)unit_col0<BITS, TILEP>(unit0 + 4 * q4);
                auto store_d = [&](int i, int t, const f32x4_t o4) {
                    const int orow = m0 + i * 16 + r16;
                    if (orow < a.M) {
                        uint2 o;
                        o.x = (uint32_t)NT::from_float(o4[0]) | ((uint32_t)NT::from_float(o4[1]) << 16);
                        o.y = (uint32_t)NT::from_float(o4[2]) | ((uint32_t)NT::from_float(o4[3]) << 16);
                        *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(a.D) + (size_t)orow * a.N + col0 + (uint32_t)((t ? f1 : f0) * TILEP)) = o;
                    }
                };
                xwg_seam<RT, NT2, NW>(acc, a.splitk, split, (uint32_t)bid, gridDim.x / (uint32_t)a.splitk, wave, lane, tid, a.partial, a.state, store_d);
                return;
            }
        }
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            const int orow = m0 + r * 16 + r16;
            if (orow < a.M) {
#pragma unroll
                for (int t = 0; t < NT2; ++t) {
                    const int col = unit_col0<BITS, TILEP>(unit0 + 4 * q4) + (t ? f1 : f0) * TILEP;
                    const f32x4_t o4 = acc[r][t];
                    if (a.splitk == 1) {
                        uint2 o;
                        o.x = (uint32_t)NT::from_float(o4[0]) | ((uint32_t)NT::from_float(o4[1]) << 16);
                        o.y = (uint32_t)NT::from_float(o4[2]) | ((uint32_t)NT::from_float(o4[3]) << 16);
                        *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(a.D) + (size_t)orow * a.N + col) = o;
                    } else {
                        *reinterpret_cast<f32x4_t*>(a.partial + ((size_t)split * a.M + orow) * a.N + col) = o4;
                    }
                }
            }
        }
    };

    if (wave < 6) body(std::integral_constant<int, 0>{});
    else if (wave == 6) body(std::integral_constant<int, 1>{});
    else body(std::integral_constant<int, 2>{});
}

}  // namespace flute_amd
