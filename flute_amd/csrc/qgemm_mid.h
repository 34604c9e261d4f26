// Mid-M kernel (round 3): 64 x 64 (or 128 x 64) output tiles for batches whose output has too few 128 / 256-row
// blocks to fill the chip - BASELINE's M = 256 on 4096-wide layers: 256 x 4096 outputs = 16 blocks of
// qgemm_block2.h, so round 2 ran it on the per-wave kernel (qgemm_tile.h: 0.16 of the MFMA peak, 2.1 x the
// algorithmic HBM traffic because every wave streams a private activation slice).  The reference fills the
// machine at any M by splitting tiles_M x tiles_N x tiles_K evenly over its CTAs
// (flute/csrc/tile_scheduler_utils.hpp:460-481) and reducing partial tiles through global memory; on this chip
// a cross-workgroup reduction costs more than the kernel (fp32 slabs + a second launch), so the tile is made
// small enough instead - 256 x 4096 -> 256 tiles of 64 x 64, one per CU - and the per-tile costs are cut:
//   * eight waves = 4 column tiles (16 columns each) x 2 k-halves: wave (ct, kh) multiplies ALL row tiles of the
//     block by its 16 columns over k = [32 kh, 32 kh + 32) of every 64-k step: one lookup per MFMA (every weight
//     of the tile is dequantised once), accumulators of the two k-halves are summed through LDS at the end;
//   * activations: one [BM][64 k] stage per step by LDS-DMA (a wave issues RT / 4 of its 1-KiB pieces), an
//     8- / 6-stage ring, ONE s_barrier per step; conflict-free ds_read_b128 fragments (qgemm_block.h's chunk swizzle);
//   * weights: 16 B per lane and step straight into registers (lane (r16, q4): words 4 q4 .. 4 q4 + 3 of unit
//     r16 % 4, field r16 / 4), hidden buffer loads in a register ring of the same depth; the lookups of step t+1 are issued between
//     the multiplies and the MFMAs of step t;
//   * scales: 8-group blocks by LDS-DMA into a wave-private double buffer (as qgemm_block2.h);
//   * blocks of one column range run on ONE XCD back to back (block id % 8 = XCD): its weights are fetched from
//     HBM once and served to the other row blocks by that XCD's L2; the activations (M x K) stay in every L2.
// Arithmetic contract as the block kernels: w^ = round_T(lut * s) (packbits_utils.hpp:139), fp32 accumulation,
// one rounding of the output.  4- and 2-bit layers (3-bit: qgemm_block3.h / qgemm_tile.h).
#pragma once
#include "qgemm_block.h"

namespace flute_amd {

// Ring depth: a 64-k step of this tile is only ~400 cycles long, so a request must be issued many steps ahead of
// its use (three stages, the block kernels' depth, left every step waiting one L2 latency: 35 us per 64-row tile
// at K = 4096 - profiles/r03_mid_lab.jsonl)
__host__ __device__ constexpr int mid_stages(int rt) { return rt == 4 ? 8 : 6; }
__host__ __device__ constexpr int mid_lds_bytes(int bits, int rt) {
    return (1 << (2 * bits)) * 128 + mid_stages(rt) * rt * 2 * 1024 + 8 * 2 * 1024;
}

template <typename T, int BITS, int TILEP, int RT>
__global__ __launch_bounds__(512) void qgemm_mid_kernel(const BlockArgs args) {
    constexpr int NSTG = mid_stages(RT);                           // ring slots (activations in LDS, weights in registers)
    constexpr int PD = NSTG - 1;                                   // a batch is issued PD steps ahead of its use
    using NT = Num<T>;
    static_assert(BITS == 4 || BITS == 2, "3-bit layers: qgemm_block3.h / qgemm_tile.h");
    static_assert(RT == 4 || RT == 8, "64- or 128-row tiles");
    constexpr int J = 16 / BITS;                                   // fields per word
    constexpr int FB = 2 * BITS;                                   // bits of a pair index
    constexpr int UW = (BITS == 4) ? 16 : 8;                       // units per 64-column tile
    constexpr int NW = 8, BM = RT * 16;
    constexpr int PIECES = RT * 2, PPW = PIECES / NW;              // 1-KiB activation pieces per stage / per wave
    // hidden loads of a step: PPW activation pieces + one weight piece, plus a scale block on the steps that start
    // one.  The counted waits assume NO scale block among the younger batches: an extra one only makes a wait release
    // one more (older, long landed) load than it needs to - every step issuing a 1-KiB scale DMA (as the block kernels
    // do, into a sink) doubled the LDS-DMA traffic of this small tile: 32.8 us per tile instead of ...
    constexpr int BATCH = PPW + 1;
    constexpr int LUT_BYTES = (1 << (2 * BITS)) * 128;
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
    const int ct = wave & 3;                                       // column tile of the block
    const int kh = wave >> 2;                                      // 32-k half of every step
    // 4 bits: tile ct = units 4 ct .. 4 ct + 3, all four fields; 2 bits: units 4 (ct % 2) .., fields 4 (ct / 2) ..
    const int ubase = (BITS == 4) ? ct * 4 : (ct & 1) * 4;
    const int fbase = (BITS == 4) ? 0 : (ct >> 1) * 4;

    int bid = blockIdx.x, split = 0;
    if (a.splitk > 1) { split = bid % a.splitk; bid /= a.splitk; }
    int tm_idx, tn_idx;
    if (a.order == 1) {                                            // XCD x = bid % 8 owns column ranges x, x + 8, ...
        const int x = bid & 7, i = bid >> 3;
        tn_idx = (i / a.tiles_m) * 8 + x;
        tm_idx = i % a.tiles_m;
    } else {
        tm_idx = bid % a.tiles_m;
        tn_idx = bid / a.tiles_m;
    }
    const int m0 = tm_idx * BM;
    const int unit0 = tn_idx * UW;
    const int kbeg = split * a.k_per_split;
    const int kend = min(a.K, kbeg + a.k_per_split);
    const int nsteps = (kend - kbeg) >> 6;
    const uint32_t row_bytes = (uint32_t)a.K * 2u;

    const srd_t x_srd = make_srd(a.A, (uint32_t)min((size_t)a.M * a.K * 2, (size_t)0xfffffff0u));
    const srd_t w_srd = make_srd(reinterpret_cast<const char*>(a.Q) + (size_t)(unit0 + ubase) * row_bytes, 4u * row_bytes);
    const srd_t s_srd = make_srd(a.S, (uint32_t)min((size_t)a.N * a.G * 2, (size_t)0xfffffff0u));
    // activations: piece p = (half p / RT, row tile p % RT), this wave's pieces p0 .. p0 + PPW - 1; rows past M lie
    // past the descriptor's range (voffset is what the range check covers) and read as zero
    const int p0 = wave * PPW;
    const uint32_t x_v0 = (uint32_t)(((size_t)(m0 + (p0 % RT) * 16 + (lane >> 2)) * a.K + (p0 / RT) * 32 +
                                      ((lane & 3) ^ blk_swz(lane >> 2)) * 8) * 2);
    const uint32_t x_dv = 16u * row_bytes;
    const uint32_t x_lds0 = (uint32_t)LUT_BYTES + (uint32_t)p0 * 1024u;
    // weights: lane (r16, q4) reads words 4 q4 .. 4 q4 + 3 of unit r16 % 4 for this wave's k-half
    const uint32_t w_voff = (uint32_t)(r16 & 3) * row_bytes + (uint32_t)(kh * 32 + q4 * 8) * 2u;
    const uint32_t shift = (uint32_t)((fbase + (r16 >> 2)) * FB);
    // scale block: lane L < 16 fetches 8 groups of weight row L of the tile (unit L % 4, field L / 4)
    const uint32_t s_voff = (lane < 16)
        ? (uint32_t)(((size_t)(unit_col0<BITS, TILEP>(unit0 + ubase + (lane & 3)) + (fbase + (lane >> 2)) * TILEP) * a.G) * 2)
        : 0x80000000u;
    const uint32_t sc_base = (uint32_t)LUT_BYTES + NSTG * STAGE_BYTES + (uint32_t)wave * 2048u;

    u32x4_t w[NSTG];
    // batch u = the hidden loads of K step u; batches past the end re-read the last step (never consumed)
    auto issue_batch = [&](auto slot_tag, int u) {
        constexpr int slot = decltype(slot_tag)::value;
        const uint32_t k0 = (uint32_t)(kbeg + min(u, nsteps - 1) * 64);
#pragma unroll
        for (int i = 0; i < PPW; ++i)
            dma16_buf(x_v0 + (uint32_t)i * x_dv, x_srd, k0 * 2u, x_lds0 + (uint32_t)i * 1024u + (uint32_t)slot * STAGE_BYTES);
        w[slot] = buf_load16(w_voff, w_srd, k0 * 2u);
        const int g = (int)(k0 >> a.lg);
        const bool blk_start = (u < nsteps) && ((g & 7) == 0 || u == 0) && ((k0 & ((1u << a.lg) - 1u)) == 0);
        if (blk_start) dma16_buf(s_voff, s_srd, (uint32_t)((g >> 3) * 16), sc_base + (uint32_t)((g >> 3) & 1) * 1024u);   // wave-uniform
    };

    [&]<int... U>(std::integer_sequence<int, U...>) {
        (issue_batch(std::integral_constant<int, U>{}, U), ...);
    }(std::make_integer_sequence<int, PD>{});
    {
        constexpr int ENT = 1 << (2 * BITS);
        for (int p = tid; p < ENT * 8; p += NW * 64) {
            const uint32_t v = a.QM2[p >> 3];
            *reinterpret_cast<uint4*>(smem + (size_t)(p >> 3) * 128 + (p & 7) * 16) = make_uint4(v, v, v, v);
        }
    }
    const uint32_t lane_off = (uint32_t)(lane & 31) * 4u;
    const uint32_t frag_base = (uint32_t)LUT_BYTES + (uint32_t)(kh * RT) * 1024u + (uint32_t)(r16 * 4 + (q4 ^ blk_swz(r16))) * 16u;
    const uint32_t sc_lane = sc_base + (uint32_t)r16 * 16u;

    f32x4_t acc[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r) acc[r] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    uint32_t v[4];                                                 // hidden lookups of the NEXT step
    uint32_t scn;                                                  // ... and its scale
    u32x4_t xf[RT];

    auto prefetch = [&](const u32x4_t& qw, int t) {                // lookups + scale of step t (this wave's half)
        const int grp = (kbeg + t * 64 + kh * 32) >> a.lg;
        const uint32_t sb = sc_lane + (uint32_t)((grp >> 3) & 1) * 1024u + (uint32_t)(grp & 7) * 2u;
        asm volatile("ds_read_u16 %0, %1" : "=v"(scn) : "v"(sb) : "memory");
#pragma unroll
        for (int ww = 0; ww < 4; ++ww) {
            const uint32_t idx = __builtin_amdgcn_ubfe(qw[ww], shift, (uint32_t)FB);
            v[ww] = lds_lookup32((idx << 7) | lane_off);
        }
    };

    // batch 0 and the pair table before anyone reads them
    asm volatile("s_waitcnt vmcnt(%1)" : "+v"(w[0]) : "n"((PD - 1) * BATCH) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    prefetch(w[0], 0);

#ifdef FLUTE_STAMPS      // development build: cycles per phase of a step, summed over the steps, per wave -> workspace
    uint64_t ph[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t tprev = __builtin_amdgcn_s_memtime();
#define FLUTE_MSTAMP(i) { const uint64_t tn_ = __builtin_amdgcn_s_memtime(); ph[i] += tn_ - tprev; tprev = tn_; }
#else
#define FLUTE_MSTAMP(i)
#endif
    auto step = [&](auto slot_tag, int t) {
        constexpr int slot = decltype(slot_tag)::value;
        constexpr int nslot = (slot + 1) % NSTG;
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            u32x4_t& dst = xf[r];
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(frag_base + (uint32_t)(slot * STAGE_BYTES)), "n"(r * 1024) : "memory");
        }
        FLUTE_MSTAMP(0);
        // stage t-1 was released by the barrier that ended step t-1: batch t+PD overwrites it (the requests are
        // issued while the fragment reads are in flight)
        issue_batch(std::integral_constant<int, (slot + PD) % NSTG>{}, t + PD);
        FLUTE_MSTAMP(1);
        // lookups / scale of THIS step (issued during the previous one): older than the RT fragment reads
        asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(scn) : "n"(RT) : "memory");
        FLUTE_MSTAMP(2);
        u32x4_t bf;
#pragma unroll
        for (int ww = 0; ww < 4; ++ww) bf[ww] = NT::mul_scale(v[ww], scn);
        // my loads of batch t+1 (issued PD steps ago; batches t+2 .. t+PD are younger): the weights of the next step
        asm volatile("s_waitcnt vmcnt(%1)" : "+v"(w[nslot]) : "n"((PD - 1) * BATCH) : "memory");
        FLUTE_MSTAMP(3);
        prefetch(w[nslot], t + 1);
        FLUTE_MSTAMP(4);
        // the fragments: older than the five reads of the prefetch
        if constexpr (RT == 4)
            asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(xf[0]), "+v"(xf[1]), "+v"(xf[2]), "+v"(xf[3]) : : "memory");
        else
            asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(xf[0]), "+v"(xf[1]), "+v"(xf[2]), "+v"(xf[3]), "+v"(xf[4 % RT]), "+v"(xf[5 % RT]),
                         "+v"(xf[6 % RT]), "+v"(xf[7 % RT]) : : "memory");
        FLUTE_MSTAMP(5);
#pragma unroll
        for (int r = 0; r < RT; ++r) acc[r] = Mfma<T>::run(bf, xf[r], acc[r]);
        FLUTE_MSTAMP(6);
        // every wave's pieces of batch t+1 have landed (each waited for its own above) and stage t is read
        __builtin_amdgcn_s_barrier();
        FLUTE_MSTAMP(7);
    };
    for (int t0 = 0;; t0 += NSTG) {
        bool last = false;
        [&]<int... U>(std::integer_sequence<int, U...>) {
            ((last ? (void)0 : (step(std::integral_constant<int, U>{}, t0 + U), last = (t0 + U + 1 >= nsteps), (void)0)), ...);
        }(std::make_integer_sequence<int, NSTG>{});
        if (last) break;
    }
#ifdef FLUTE_STAMPS
    if (lane == 0 && a.splitk == 1 && a.partial != nullptr) {
        uint64_t* o = reinterpret_cast<uint64_t*>(a.partial) + ((size_t)blockIdx.x * NW + wave) * 16;
        for (int i = 0; i < 8; ++i) o[i] = ph[i];
        o[8] = (uint64_t)nsteps;
    }
#endif
#undef FLUTE_MSTAMP
    // drain what is still in flight (the clamped batches and the prefetch past the end)
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(scn) : : "memory");
#pragma unroll
    for (int i = 0; i < NSTG; ++i) asm volatile("s_waitcnt vmcnt(0)" : "+v"(w[i]) : : "memory");
    __builtin_amdgcn_s_barrier();                                  // every LDS-DMA into the stages has landed

    // ---- the two k-halves of a column tile are summed through LDS (the stage area), then stored: accumulator
    // register e of lane (r16, q4) = weight row 4 q4 + e of the tile = unit ubase + e, field fbase + q4: four
    // consecutive columns; the lane's output row is r16 of its row tile ----
    f32x4_t* xch = reinterpret_cast<f32x4_t*>(smem + LUT_BYTES);
    if (kh == 1) {
#pragma unroll
        for (int r = 0; r < RT; ++r) xch[(ct * RT + r) * 64 + lane] = acc[r];
    }
    __syncthreads();
    if (kh == 0) {
        const int col = unit_col0<BITS, TILEP>(unit0 + ubase) + (fbase + q4) * TILEP;
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            const f32x4_t o4 = acc[r] + xch[(ct * RT + r) * 64 + lane];
            const int row = m0 + r * 16 + r16;
            if (row < a.M) {
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
