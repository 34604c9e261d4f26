// Decode-regime kernel (M <= 4): HBM-bound streaming LUT-dequant GEMV.
//
// Replaces, for small M, the reference's qgemm_device main loop
// (flute/csrc/qgemm_kernel.hpp:617-712) + Stream-K fixup
// (tile_scheduler_utils.hpp:58-211).  CDNA4 design, not a translation.  On
// gfx950 every VALU op of a wave64 costs a full quad-cycle and the r01 profile
// showed the first version VALU/latency bound (7.5 VALU per weight pair, 60-75 %
// of wave time in s_waitcnt), so this version is built around instruction count:
//   * one wave streams ONE unit (one Q32 row / the 3-plane triple for b=3)
//     along K; lane l of load i reads bytes [16 l, 16 l + 16) of the i-th KiB:
//     every global_load_dwordx4 is one fully coalesced 1 KiB burst straight to
//     VGPRs (weights are used once - no LDS round trip);
//   * the pair table lives in LDS with a 256-byte entry stride: the LDS address
//     of a lookup is {0, 0, field byte, lane*4}, built by ONE v_perm_b32, and
//     lane l always reads bank l (conflict-free data-dependent ds_read):
//       b=4: 256 entries x 64 copies x 4 B;  b=2: a BYTE table (two 4-bit
//       fields -> two columns) 256 x 32 copies x 8 B, read with ds_read_b64;
//       b=3 (6-bit fields, not byte aligned): 64 x 32 copies x 4 B, bfe+lshl_or;
//   * the group scale is applied once per 8-k run in fp32
//     (acc += s * sum_k x_k*lut_k) instead of once per pair: same value as the
//     reference's round_T(lut*s) contract on one-hot inputs, within 2^-11
//     relative per term otherwise (PRE = true keeps the per-pair v_pk_mul_f16);
//   * activations and scales are staged in LDS once per K chunk, the chunk as
//     large as LDS allows (normally the whole K range: one barrier);
//   * workgroups are persistent over unit groups (table built once), K is split
//     over `kw` waves inside the workgroup (LDS reduce) and, only when the grid
//     would otherwise be too small, across workgroups (fp32 slabs + reduce pass);
//   * lane reduction with DPP row ops + v_readlane, not ds_bpermute.
#pragma once
#include <type_traits>

#include "common.h"
#include "fwht.h"

namespace flute_amd {

template <int BITS> struct DecCfg {
#ifndef FLUTE_DEC_CK
#define FLUTE_DEC_CK 2
#endif
    static constexpr int CK = (BITS == 3) ? 1 : FLUTE_DEC_CK;   // adjacent 16-B pieces per lane and slot
#ifndef FLUTE_DEC_U
#define FLUTE_DEC_U 2
#endif
    // slots in flight per lane.  Measured on MI355X (profiles/r01_ring_depth.txt): 2 slots x 2
    // pieces (4 KiB per wave) is the optimum for b=4/2 - deeper rings get SLOWER (a wave blocks
    // at its refill load when the memory queues are full and cannot run the compute it
    // already has data for).  b=3: 2 slots of one piece per plane keep the M=1 variant under 128
    // VGPRs, i.e. 16 waves per CU: its 3-op field extraction makes it the most compute-heavy
    // variant and it gains more from waves that compute while others wait than from a deeper ring.
    static constexpr int U = FLUTE_DEC_U;
    static constexpr int LUT_BYTES = (BITS == 3) ? 64 * 128 : 65536;
};

// largest workgroup a variant may be launched with: 1024 threads cap the kernel at 128
// VGPRs, which only the light variants fit without spilling
__host__ __device__ constexpr int dec_max_threads(int bits, int mb) {
    return ((bits == 4 && mb <= 2) || (bits == 3 && mb == 1)) ? 1024 : 512;
}

struct DecodeGeom {
    int kc;            // k per staged chunk (multiple of 512)
    int nbuf;          // 1: whole K range staged once; 2: chunked, double buffered
    int gcap;          // scale groups per chunk buffer
    int upw;           // units per workgroup
    size_t x_off, s_off, red_off, total;
};

// LDS carve shared by host (plan) and device.  `krange` = k handled by one workgroup.
__host__ __device__ inline DecodeGeom decode_geom(int bits, int mb, int lg, int waves, int kw,
                                                  int krange, int lds_budget) {
    DecodeGeom g;
    const int J = (bits == 3) ? 16 : 16 / bits;
    const int lut = (bits == 3) ? 64 * 128 : 65536;
    g.upw = waves / kw;
    const int ncols = g.upw * J;
    const int red = waves * J * mb * 4;
    // bytes per k of a chunk buffer: activations 2*mb, scales 4*ncols per group
    int kc = (krange + 511) & ~511;
    g.nbuf = 1;
    auto need = [&](int kcc, int nb) {
        const long gc = (((kcc >> lg) + 1) + 7) & ~7L;       // staged groups, multiple of 8
        return (long)lut + (long)nb * ((long)mb * kcc * 2 + gc * ncols * 4) + red + 64;
    };
    if (need(kc, 1) > lds_budget) {
        g.nbuf = 2;
        kc = 8192;
        while (kc > 512 && need(kc, 2) > lds_budget) kc >>= 1;
    }
    g.kc = kc;
    g.gcap = (kc >> lg) + 1;
    g.x_off = lut;
    g.s_off = g.x_off + (size_t)g.nbuf * mb * kc * 2;
    g.red_off = g.s_off + (size_t)g.nbuf * (size_t)((g.gcap + 7) & ~7) * ncols * 4;
    g.total = g.red_off + red;
    return g;
}

// sum over the 64 lanes, result valid in every lane of the first row (lane 0 used)
__device__ __forceinline__ float wave_sum_dpp(float v) {
    // within each row of 16 lanes: butterfly via quad_perm / row_half_mirror / row_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));  // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));  // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true)); // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true)); // row_mirror
    const int iv = __builtin_bit_cast(int, v);
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 48));
    return (v + r1) + (r2 + r3);
}

// DBG (ablation builds only): bit 0 = skip the LDS table lookups, bit 1 = skip the weight loads
template <typename T, int BITS, int TILEP, int MB, bool PRE, int DBG = 0>
__global__ __launch_bounds__(dec_max_threads(BITS, MB)) void qgemv_kernel(const QGemmArgs a) {
    using L = Layout<BITS>;
    using NT = Num<T>;
    constexpr int J = L::J;
    constexpr int NP = L::NPLANES;
    constexpr int U = DecCfg<BITS>::U;   // 128-VGPR variants
    constexpr int CK = DecCfg<BITS>::CK;
    constexpr int LPS = 8 * CK;            // 64-k lines covered by one wave-wide slot
    constexpr int LSH = (CK == 2) ? 2 : 3; // lanes per line = 8 / CK

    extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef FLUTE_STAMPS   // development build: 100 MHz wall-clock stamps of the FIRST visit, per wave, into the workspace
    uint64_t stamp[8];
    for (int i = 0; i < 8; ++i) stamp[i] = 0;
    stamp[0] = wall_clock64();
#define FLUTE_DSTAMP(i) if (stamp[i] == 0) stamp[i] = wall_clock64()
#else
#define FLUTE_DSTAMP(i)
#endif

    const int tid = threadIdx.x;
    const int nthr = blockDim.x;
    const int lane = tid & 63;
    const int sub = lane & ((8 / CK) - 1);   // (16*CK)-byte piece inside the 128-B line
    const int oct = lane >> LSH;             // which of the LPS lines of one wave-wide slot
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nw = nthr >> 6;
    const int kw = a.kw;
#ifdef FLUTE_STAMPS
    asm volatile("" :: "s"(kw));             // the kernel arguments have arrived
    stamp[7] = wall_clock64();
#endif
    const int lkw = a.lkw;
    const int ul = wave >> lkw;
    const int kpart = wave & (kw - 1);
    const int lg = a.lg;

    // LDS carve of decode_geom(), evaluated by the host planner (no integer division in this prologue)
    struct { int kc, nbuf, gcap, upw, x_off, s_off; } geo = {a.geo[0], a.geo[1], a.geo[2], 1 << a.geo[3], a.geo[4], a.geo[5]};
    const int KC = geo.kc;
    const int upw = geo.upw;
    const int ncols = upw * J;
    const int lncols = a.geo[3] + ((J == 4) ? 2 : (J == 8 ? 3 : 4));
    uint16_t* xs = reinterpret_cast<uint16_t*>(smem + a.geo[4]);
    uint32_t* ss = reinterpret_cast<uint32_t*>(smem + a.geo[5]);
    float* red = reinterpret_cast<float*>(smem + a.geo[6]);
    const int gstride = (geo.gcap + 7) & ~7;                  // words per column in the scale stage
    const size_t ss_buf_words = (size_t)gstride * ncols;

    const uint16_t* A = reinterpret_cast<const uint16_t*>(a.A);
    const uint16_t* S = reinterpret_cast<const uint16_t*>(a.S);
    // the v_perm-built table addresses are absolute: the table must sit at LDS byte 0
    const uint32_t lds0 = lds_base_of(smem);
    if (lds0 != 0) __builtin_trap();

    int split = 0, wg = blockIdx.x, nwg = gridDim.x;
    if (a.splitk > 1) {                                        // rare (very narrow layers)
        split = blockIdx.x % a.splitk;
        wg = blockIdx.x / a.splitk;
        nwg = gridDim.x / a.splitk;
    }
    const int ngroups = a.units >> a.geo[3];
    const int kbeg = split * a.k_per_split;
    const int kend = min(a.K, kbeg + a.k_per_split);

    // ---- pair table -> LDS, split into "issue the global loads" and "write LDS" so that the
    // loads travel together with the first weight / activation / scale loads ----
    constexpr int LUT_ENT = (BITS == 3) ? 64 : 256;          // entries
    constexpr int LUT_PCS = (BITS == 3) ? 2 : 4;             // 64-B pieces per entry
    constexpr bool BIG_WG = dec_max_threads(BITS, MB) == 1024;   // 128-VGPR variants: keep the prologue lean
    constexpr int LUT_R = BIG_WG ? 2 : 4;                     // table pieces per thread held in registers
    // ASYNC = loads hidden from hipcc (inline asm) and waited for with ONE counted s_waitcnt after the
    // first weight-ring slots have been issued behind them: the prologue's staging data then returns
    // FIRST (loads retire in order) and the first ring slot is consumed while the rest of the weights
    // are still streaming.  Every staging load is unconditional (clamped address): its register must
    // not be read, copied or merged before the wait.
    uint32_t lut_v0[LUT_R], lut_v1[LUT_R];
    auto lut_issue = [&](auto async_tag) {
        constexpr bool ASYNC = decltype(async_tag)::value;
#pragma unroll
        for (int r = 0; r < LUT_R; ++r) {
            const int e = min(tid + r * nthr, LUT_ENT * LUT_PCS - 1) / LUT_PCS;
            const uint32_t* s0 = a.QM2 + ((BITS == 2) ? (e & 15) : e);
            const uint32_t* s1 = a.QM2 + ((BITS == 2) ? (e >> 4) : e);
            lut_v1[r] = 0;
            if constexpr (ASYNC) {
                lut_v0[r] = ring_load4(s0);
                if constexpr (BITS == 2) lut_v1[r] = ring_load4(s1);
            } else {
                lut_v0[r] = *s0;
                if constexpr (BITS == 2) lut_v1[r] = *s1;
            }
        }
    };
    auto lut_commit = [&]() {
#pragma unroll
        for (int r = 0; r < LUT_R; ++r) {
            const int p = tid + r * nthr;
            if (p < LUT_ENT * LUT_PCS) {
                const int e = p / LUT_PCS;
                uint4* d = reinterpret_cast<uint4*>(smem + (size_t)e * (LUT_PCS * 64) + (p % LUT_PCS) * 64);
                const uint32_t v1 = (BITS == 2) ? lut_v1[r] : lut_v0[r];
                const uint4 vv = make_uint4(lut_v0[r], v1, lut_v0[r], v1);
                d[0] = vv; d[1] = vv; d[2] = vv; d[3] = vv;
            }
        }
        for (int p = tid + LUT_R * nthr; p < LUT_ENT * LUT_PCS; p += nthr) {     // tiny workgroups only
            const int e = p / LUT_PCS;
            uint32_t v0, v1;
            if constexpr (BITS == 2) { v0 = a.QM2[e & 15]; v1 = a.QM2[e >> 4]; }
            else { v0 = a.QM2[e]; v1 = v0; }
            uint4* d = reinterpret_cast<uint4*>(smem + (size_t)e * (LUT_PCS * 64) + (p % LUT_PCS) * 64);
            const uint4 vv = make_uint4(v0, v1, v0, v1);
            d[0] = vv; d[1] = vv; d[2] = vv; d[3] = vv;
        }
    };
    // per-lane low address byte(s) of a lookup
    const uint32_t lane_off = (BITS == 2) ? (uint32_t)(lane & 31) * 8 : (BITS == 4 ? (uint32_t)lane * 4
                                                                                  : (uint32_t)(lane & 31) * 4);

    const size_t row_words = (size_t)(a.K >> 1);
    // ---- the wave's work is a flat list of VISITS (unit group, K chunk); its weights form ONE
    // continuous stream through a U-slot register ring: a refill that runs past the end of a
    // visit fetches the head of the next one, so the ring only drains when the kernel ends ----
    // chunked staging (nbuf == 2) uses power-of-two chunks
    const int nchunks = (geo.nbuf == 1) ? 1 : (kend - kbeg + KC - 1) >> a.geo[7];
    const int nug = a.geo[8] + (wg < a.geo[9] ? 1 : 0);        // unit groups wg, wg + nwg, ... < ngroups
    const int nvisits = nug * nchunks;
    struct Visit { int ug, c, kc0, kc_len, l0, myL, nIp, kbase, lnmax; };
    auto visit_of = [&](int ugi, int c_) -> Visit {            // visit = (ugi-th unit group of this workgroup, chunk c_)
        Visit t;
        t.ug = wg + ugi * nwg;
        t.c = c_;
        t.kc0 = kbeg + t.c * KC;
        t.kc_len = min(KC, kend - t.kc0);
        const int Lc = t.kc_len >> 6;                                        // 64-k lines in the chunk
        const int Lw = (((Lc + kw - 1) >> lkw) + LPS - 1) & ~(LPS - 1);      // lines per wave
        t.l0 = kpart * Lw;
        t.myL = max(0, min(Lw, Lc - t.l0));
        const int nI = (t.myL + LPS - 1) / LPS;
        t.nIp = max(U, ((nI + U - 1) / U) * U);                              // slots, padded to the ring size
        t.kbase = t.kc0 + ((t.myL > 0) ? t.l0 : 0) * 64;                      // wave without lines: stay in the row
        t.lnmax = max(t.myL, 1) - 1;
        return t;
    };
    auto rows_of = [&](int ug_, const uint32_t* (&rows)[NP]) {
        const int u_ = ug_ * upw + ul;
#pragma unroll
        for (int pl = 0; pl < NP; ++pl)
            rows[pl] = a.Q + (size_t)unit_row<BITS, TILEP>(u_, pl, a.N) * row_words + sub * (4 * CK);
    };

    ring16_t q[U][NP][CK];
    constexpr int RING_SLOT = NP * CK;               // loads per slot
    Visit cur, nxt;
    int vis_ugi = 0;                                 // unit-group ordinal of the current visit
    const uint32_t* crow[NP];
    const uint32_t* nrow[NP];
    auto load_next_desc = [&](int v) {               // descriptor of visit v+1 (or a clamp onto cur's last line)
        if (v + 1 < nvisits) {
            int nc = cur.c + 1, nu = vis_ugi;
            if (nc == nchunks) { nc = 0; ++nu; }
            nxt = visit_of(nu, nc);
            rows_of(nxt.ug, nrow);
        } else {
            nxt = cur;
            nxt.kbase = cur.kbase + cur.lnmax * 64;
            nxt.lnmax = 0;
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) nrow[pl] = crow[pl];
        }
    };
    // Ring loads are UNCONDITIONAL (clamped, never branched around): a load inside a branch makes
    // hipcc lose count of the outstanding loads, and the counted waits below rely on exactly one
    // refill per consumed slot.
    auto ring_load = [&](int i, int s) {             // s: slot index relative to the current visit
        const bool in_cur = s < cur.nIp;
        const int sl = in_cur ? s : s - cur.nIp;
        const int lc = min(sl * LPS + oct, in_cur ? cur.lnmax : nxt.lnmax);
        const int kk = (in_cur ? cur.kbase : nxt.kbase) + lc * 64;
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) {
            const uint32_t* base = in_cur ? crow[pl] : nrow[pl];
#pragma unroll
            for (int ck = 0; ck < CK; ++ck) q[i][pl][ck] = ring_load16_nt(base + (kk >> 1) + ck * 4);
        }
    };
    // Staging of one visit's activations and scales, split into "issue the loads" (first XP 16-B pieces
    // of the activations and one 8-group piece of scales per thread, kept in registers) and "write LDS"
    // (+ plain load-and-store passes for whatever exceeds those registers).
    constexpr int XP = BIG_WG ? 1 : 4;                    // 16-B activation pieces per thread (registers)
    int staged_chunk = -1;               // chunk whose activations sit in LDS (nbuf == 1 case)
    ring16_t xv[XP], sv;
    struct StageCtx { int ug, c, kc0, kc_len, g0c, gcnt, xpieces, spieces, s_cidx, s_gp; bool s_fast; };
    StageCtx sc;
    auto scale_src = [&](int pidx, int& cidx, int& gp) -> const uint16_t* {
        // consecutive threads take consecutive columns: conflict-free LDS writes below
        const int gpi = pidx >> lncols;                 // ncols = upw * J is a power of two
        cidx = pidx - gpi * ncols;
        gp = gpi * 8;
        const int ulc = cidx / J;
        const int j = cidx - ulc * J;
        const int n = unit_col0<BITS, TILEP>(sc.ug * upw + ulc) + j * TILEP;
        return S + (size_t)n * a.G + sc.g0c + gp;
    };
    auto stage_issue = [&](const Visit& vis, auto async_tag) {
        constexpr bool ASYNC = decltype(async_tag)::value;
        sc.ug = vis.ug; sc.c = vis.c; sc.kc0 = vis.kc0; sc.kc_len = vis.kc_len;
        sc.g0c = vis.kc0 >> lg;
        sc.gcnt = ((vis.kc0 + vis.kc_len - 1) >> lg) - sc.g0c + 1;
        const bool stage_x = (geo.nbuf == 2) || (staged_chunk != vis.c);
        sc.xpieces = stage_x ? MB * (KC / 8) : 0;
        sc.spieces = ncols * ((sc.gcnt + 7) >> 3);
        FLUTE_DSTAMP(4);
        if (sc.xpieces > 0) {                                     // workgroup-uniform
#pragma unroll
            for (int r = 0; r < XP; ++r) {
                const int pidx = min(r * nthr + tid, sc.xpieces - 1);
                const int m = (MB == 1) ? 0 : pidx / (KC / 8);
                const int kk = min((pidx - m * (KC / 8)) * 8, sc.kc_len - 8);
                const uint16_t* src = A + (size_t)min(a.m0 + m, a.M - 1) * a.K + sc.kc0 + kk;
                if constexpr (ASYNC) xv[r] = ring_load16(src);
                else xv[r] = *reinterpret_cast<const ring16_t*>(src);
            }
        }
        {
            const uint16_t* sp0 = scale_src(min(tid, sc.spieces - 1), sc.s_cidx, sc.s_gp);
            // full, 16-B aligned run of 8 groups: one vector load; anything else is read element-wise at commit
            sc.s_fast = (sc.s_gp + 8 <= sc.gcnt) && ((reinterpret_cast<uintptr_t>(sp0) & 15) == 0);
            const uint16_t* src = sc.s_fast ? sp0 : S;
            if constexpr (ASYNC) sv = ring_load16(src);
            else sv = *reinterpret_cast<const ring16_t*>(src);
        }
    };
    auto stage_commit = [&]() {
        const int buf = (geo.nbuf == 2) ? (sc.c & 1) : 0;
        uint16_t* xsb = xs + (size_t)buf * MB * KC;
        uint32_t* ssb = ss + (size_t)buf * ss_buf_words;
        const int kc0 = sc.kc0, kc_len = sc.kc_len, gcnt = sc.gcnt;
        auto scale_load_slow = [&](const uint16_t* sp, int gp) -> uint4 {
            uint16_t h[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) h[r] = (gp + r < gcnt) ? sp[r] : (uint16_t)0;
            return make_uint4(h[0] | ((uint32_t)h[1] << 16), h[2] | ((uint32_t)h[3] << 16),
                              h[4] | ((uint32_t)h[5] << 16), h[6] | ((uint32_t)h[7] << 16));
        };
        auto scale_load = [&](const uint16_t* sp, int gp) -> uint4 {
            if (gp + 8 <= gcnt && ((reinterpret_cast<uintptr_t>(sp) & 15) == 0))
                return *reinterpret_cast<const uint4*>(sp);
            return scale_load_slow(sp, gp);
        };
        // scales live in LDS as [column][gstride] 32-bit words (fp32, or raw T for PRE)
        auto scale_store = [&](uint4 t, int cidx, int gp) {
            const uint32_t w[4] = {t.x, t.y, t.z, t.w};
            uint32_t o[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const uint16_t h = (uint16_t)((r & 1) ? (w[r >> 1] >> 16) : (w[r >> 1] & 0xffff));
                o[r] = PRE ? (uint32_t)h : __builtin_bit_cast(uint32_t, NT::to_float(h));
            }
#pragma unroll
            for (int r = 0; r < 8; ++r)
                if (gp + r < gstride) ssb[(size_t)(gp + r) * ncols + cidx] = o[r];
        };
        FLUTE_DSTAMP(5);
        FLUTE_DSTAMP(6);
#pragma unroll
        for (int r = 0; r < XP; ++r) {
            const int pidx = r * nthr + tid;
            if (pidx < sc.xpieces) {
                const int m = (MB == 1) ? 0 : pidx / (KC / 8);
                const int kk = (pidx - m * (KC / 8)) * 8;
                uint32_t w[4] = {xv[r].x, xv[r].y, xv[r].z, xv[r].w};
                // fused pre-rotation (flute.qgemm_hadamard, qgemm.cpp:201-244): the 64 pieces of a wave
                // are 512 consecutive k of one row = whole Hadamard blocks (K % had == 0, had <= 512)
                if (a.had_log > 0) fwht_piece<T>(w, lane, a.had_log, a.had_scale);
                const uint4 v = (kk < kc_len) ? make_uint4(w[0], w[1], w[2], w[3]) : make_uint4(0, 0, 0, 0);
                *reinterpret_cast<uint4*>(xsb + (size_t)m * KC + kk) = v;
            }
        }
        if (tid < sc.spieces) {
            int cidx, gp;
            const uint16_t* sp0 = scale_src(tid, cidx, gp);
            const uint4 t = sc.s_fast ? make_uint4(sv.x, sv.y, sv.z, sv.w) : scale_load_slow(sp0, gp);
            scale_store(t, cidx, gp);
        }
        // leftovers (large MB*K or many columns): plain load-then-store passes
        for (int pidx = XP * nthr + tid; pidx < sc.xpieces; pidx += nthr) {
            const int m = (MB == 1) ? 0 : pidx / (KC / 8);
            const int kk = (pidx - m * (KC / 8)) * 8;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (kk < kc_len)
                v = *reinterpret_cast<const uint4*>(A + (size_t)min(a.m0 + m, a.M - 1) * a.K + kc0 + kk);
            if (a.had_log > 0) {
                uint32_t w[4] = {v.x, v.y, v.z, v.w};
                fwht_piece<T>(w, lane, a.had_log, a.had_scale);
                v = make_uint4(w[0], w[1], w[2], w[3]);
            }
            *reinterpret_cast<uint4*>(xsb + (size_t)m * KC + kk) = v;
        }
        for (int pidx = nthr + tid; pidx < sc.spieces; pidx += nthr) {
            int cidx, gp;
            const uint16_t* sp = scale_src(pidx, cidx, gp);
            scale_store(scale_load(sp, gp), cidx, gp);
        }
        staged_chunk = sc.c;
    };
    if (nvisits > 0) {
        // Issue order = return order: table words, visit 0's activations and scales, then the first U
        // ring slots.  One counted wait (U ring slots may still be in flight) releases the staging
        // data; the wave then writes LDS while its weights are still arriving.
        cur = visit_of(0, 0);
        rows_of(cur.ug, crow);
        lut_issue(std::true_type{});
        stage_issue(cur, std::true_type{});
#pragma unroll
        for (int i = 0; i < U; ++i) ring_load(i, i);     // slots < cur.nIp: needs no successor descriptor
        load_next_desc(0);
        {
            constexpr int NY = U * RING_SLOT;            // loads issued after the staging loads
#pragma unroll
            for (int r = 0; r < LUT_R; ++r) ring_wait4<NY>(lut_v0[r], lut_v1[r]);
#pragma unroll
            for (int r = 0; r < XP; ++r) ring_wait<NY>(xv[r]);
            ring_wait<NY>(sv);
        }
        lut_commit();
        stage_commit();
    }

    float acc[J][MB];
    for (int v = 0; v < nvisits; ++v) {
        const int ug = cur.ug;
        const int c = cur.c;
        if (c == 0) {
#pragma unroll
            for (int j = 0; j < J; ++j)
#pragma unroll
                for (int m = 0; m < MB; ++m) acc[j][m] = 0.f;
        }
        {
            const int buf = (geo.nbuf == 2) ? (c & 1) : 0;
            const int kc0 = cur.kc0;
            const int kc_len = cur.kc_len;
            const int l0 = cur.l0;
            const int myL = cur.myL;
            const int g0c = kc0 >> lg;
            const int gcnt = ((kc0 + kc_len - 1) >> lg) - g0c + 1;
            uint16_t* xsb = xs + (size_t)buf * MB * KC;
            uint32_t* ssb = ss + (size_t)buf * ss_buf_words;

            if (v > 0) { stage_issue(cur, std::false_type{}); stage_commit(); }   // visit 0: staged by the prologue
            __syncthreads();
            // Every load issued so far has landed (in-order retirement: the staging loads above
            // were consumed).  Tell hipcc so: otherwise it guards registers last written by a
            // staging load with s_waitcnt vmcnt(0) at the loop header and drains the ring.
            __builtin_amdgcn_s_waitcnt(0x0F70);     // vmcnt(0), lgkmcnt/expcnt untouched
            FLUTE_DSTAMP(1);

            // ---- stream the lines: consume slot i, then refill it U slots ahead ----
            for (int it = 0; it < cur.nIp; it += U) {
#pragma unroll
                for (int i = 0; i < U; ++i) {
                    const int ln = (it + i) * LPS + oct;
                    // slot i is valid once at most the (U-1) younger slots are outstanding
                    if constexpr (NP == 1 && CK == 2) ring_wait<(U - 1) * RING_SLOT>(q[i][0][0], q[i][0][1]);
                    else if constexpr (NP == 3) ring_wait<(U - 1) * RING_SLOT>(q[i][0][0], q[i][1][0], q[i][2][0]);
                    else ring_wait<(U - 1) * RING_SLOT>(q[i][0][0]);
                    if (ln < myL) {
                        const int kl = (l0 + ln) * 64 + sub * (8 * CK);    // k inside the chunk
                        const int gl = ((kc0 + kl) >> lg) - g0c;
                        uint32_t sw[J];
                        {
                            const uint32_t sa = (uint32_t)geo.s_off + (uint32_t)buf * (uint32_t)(ss_buf_words * 4) +
                                                (uint32_t)(gl * ncols + ul * J) * 4;
#pragma unroll
                            for (int h = 0; h < J / 4; ++h) {
                                const uint4 t = lds_ld128(sa + 16 * h);
                                sw[4 * h] = t.x; sw[4 * h + 1] = t.y; sw[4 * h + 2] = t.z; sw[4 * h + 3] = t.w;
                            }
                        }
                        uint32_t xw[MB][4 * CK];
                        {
                            const uint32_t xa = (uint32_t)geo.x_off + (uint32_t)buf * (uint32_t)(MB * KC * 2) + (uint32_t)kl * 2;
#pragma unroll
                            for (int m = 0; m < MB; ++m)
#pragma unroll
                                for (int ck = 0; ck < CK; ++ck) {
                                    const uint4 t = lds_ld128(xa + (uint32_t)(m * KC * 2) + 16 * ck);
                                    xw[m][4 * ck] = t.x; xw[m][4 * ck + 1] = t.y;
                                    xw[m][4 * ck + 2] = t.z; xw[m][4 * ck + 3] = t.w;
                                }
                        }
                        // column outer, k-pair inner: the 8-k partial sums of a column live in
                        // MB temporaries; the group scale is applied once per column and line
                        if constexpr (BITS == 2) {
#pragma unroll
                            for (int jp = 0; jp < 4; ++jp) {
                                float a0[MB], a1[MB];
#pragma unroll
                                for (int m = 0; m < MB; ++m) {
                                    a0[m] = PRE ? acc[2 * jp][m] : 0.f;
                                    a1[m] = PRE ? acc[2 * jp + 1][m] : 0.f;
                                }
#pragma unroll
                                for (int ww = 0; ww < 4 * CK; ++ww) {
                                    const uint32_t w0 = q[i][0][ww >> 2][ww & 3];
                                    const uint32_t addr = __builtin_amdgcn_perm(w0, lane_off, 0x0c0c0400u | ((4u + jp) << 8));
                                    const uint2 v2 = lds_ld64(addr);
                                    uint32_t v0 = v2.x, v1 = v2.y;
                                    if constexpr (PRE) {
                                        v0 = NT::mul_scale(v0, sw[2 * jp]);
                                        v1 = NT::mul_scale(v1, sw[2 * jp + 1]);
                                    }
#pragma unroll
                                    for (int m = 0; m < MB; ++m) {
                                        a0[m] = NT::dot2(v0, xw[m][ww], a0[m]);
                                        a1[m] = NT::dot2(v1, xw[m][ww], a1[m]);
                                    }
                                }
#pragma unroll
                                for (int m = 0; m < MB; ++m) {
                                    acc[2 * jp][m] = PRE ? a0[m] : __builtin_fmaf(a0[m], __builtin_bit_cast(float, sw[2 * jp]), acc[2 * jp][m]);
                                    acc[2 * jp + 1][m] = PRE ? a1[m] : __builtin_fmaf(a1[m], __builtin_bit_cast(float, sw[2 * jp + 1]), acc[2 * jp + 1][m]);
                                }
                            }
                        } else {
#pragma unroll
                            for (int j = 0; j < J; ++j) {
                                float al[MB];
#pragma unroll
                                for (int m = 0; m < MB; ++m) al[m] = PRE ? acc[j][m] : 0.f;
#pragma unroll
                                for (int ww = 0; ww < 4 * CK; ++ww) {
                                    uint32_t w[NP];
#pragma unroll
                                    for (int pl = 0; pl < NP; ++pl)
                                        w[pl] = q[i][pl][ww >> 2][ww & 3];
                                    uint32_t addr;
                                    if constexpr (BITS == 4)
                                        addr = __builtin_amdgcn_perm(w[0], lane_off, 0x0c0c0400u | ((4u + j) << 8));
                                    else
                                        addr = (field<BITS>(w, j) << 7) | lane_off;
                                    uint32_t v = (DBG & 1) ? addr : lds_ld32(addr);
                                    if constexpr (PRE) v = NT::mul_scale(v, sw[j]);
#pragma unroll
                                    for (int m = 0; m < MB; ++m) al[m] = NT::dot2(v, xw[m][ww], al[m]);
                                }
#pragma unroll
                                for (int m = 0; m < MB; ++m)
                                    acc[j][m] = PRE ? al[m] : __builtin_fmaf(al[m], __builtin_bit_cast(float, sw[j]), acc[j][m]);
                            }
                        }

                    }
                    // refill this slot U slots ahead: same visit, or the head of the next one
                    ring_load(i, it + i + U);
                }
            }
        }
        // advance the stream descriptors (the ring already holds the next visit's first U slots)
        if (cur.c == nchunks - 1) ++vis_ugi;
        cur = nxt;
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) crow[pl] = nrow[pl];
        load_next_desc(v + 1);
        if (c != nchunks - 1) continue;
        FLUTE_DSTAMP(2);

        // ---- lanes -> wave (DPP) -> kw waves (LDS) -> output ----
#pragma unroll
        for (int j = 0; j < J; ++j)
#pragma unroll
            for (int m = 0; m < MB; ++m) {
                const float v = wave_sum_dpp(acc[j][m]);
                if (lane == 0) red[wave * (J * MB) + j * MB + m] = v;
            }
        __syncthreads();
        for (int t = tid; t < upw * J * MB; t += nthr) {
            const int ulc = t / (J * MB);
            const int r = t - ulc * (J * MB);
            const int j = r / MB;
            const int m = r - j * MB;
            float sum = 0.f;
            for (int kp = 0; kp < kw; ++kp) sum += red[(ulc * kw + kp) * (J * MB) + r];
            const int row = a.m0 + m;
            if (row < a.M) {
                const int n = unit_col0<BITS, TILEP>(ug * upw + ulc) + j * TILEP;
                if (a.splitk == 1)
                    reinterpret_cast<uint16_t*>(a.D)[(size_t)row * a.N + n] = NT::from_float(sum);
                else
                    a.partial[((size_t)split * a.M + row) * a.N + n] = sum;
            }
        }
    }
    // the last refills are clamped re-reads still landing in q[]: drain before the wave ends
    if (nvisits > 0) {
#pragma unroll
        for (int i = 0; i < U; ++i) {
            if constexpr (NP == 1 && CK == 2) ring_wait<0>(q[i][0][0], q[i][0][1]);
            else if constexpr (NP == 3) ring_wait<0>(q[i][0][0], q[i][1][0], q[i][2][0]);
            else ring_wait<0>(q[i][0][0]);
        }
    }
#ifdef FLUTE_STAMPS
    __builtin_amdgcn_s_waitcnt(0);
    stamp[3] = wall_clock64();
    if (lane == 0 && a.splitk == 1 && a.partial != nullptr) {
        uint64_t* o = reinterpret_cast<uint64_t*>(a.partial) + ((size_t)blockIdx.x * nw + wave) * 8;
        for (int i = 0; i < 8; ++i) o[i] = stamp[i];
    }
#endif
#undef FLUTE_DSTAMP
}

}  // namespace flute_amd
