// Host side of the C ABI (include/flute_amd.h): template table, launch planning
// and dispatch.  Mirrors the role of flute/csrc/qgemm.cpp:39-83 (qgemm_raw) +
// qgemm_kernel_raw_generated.cu:15-768 (_qgemm_raw's template switch) +
// qgemm_kernel.hpp:824-939 (qgemm_host), re-thought for gfx950: instead of one
// Stream-K kernel with 36/144 tile variants there are two kernel families
// (streaming decode for M <= 2 - on small layers M <= 4 -, MFMA above, block-tiled MFMA for prefill) whose launch geometry is derived
// from the template's knobs and the problem shape.  Nothing here is process-global mutable state
// except the per-device "large LDS granted" cache (mutex-guarded): plan overrides travel with the call.
#include <hip/hip_runtime.h>
#include <math.h>
#include <algorithm>
#include <mutex>
#include <stdint.h>
#include <string.h>
#include <vector>

#include "../../include/flute_amd.h"
#include "kernels.h"
#include "qgemm_stream.h"
#include "qgemm_persist.h"
#include "qgemm_fast.h"
#include "qgemm_fastm.h"
#include "qgemm_persistm.h"
#include "qgemm_skinny.h"
#include "mfma.h"
#include "qgemm_tile.h"
#include "qgemm_block.h"
#include "qgemm_splitk.h"

using namespace flute_amd;

namespace {

// had8: set by flute_qgemm_hadamard for M <= 4 - the DECODE planners then prefer 8-wave workgroups (the fused rotation is
// done by the workgroup's waves, 512 k each); it never reaches the MFMA planners, whose wave count is the template's
struct Ovr { int family, m_block, waves, kw, splitk, m_tiles, slabs, depth, one_shot, had8; };
Ovr ovr_of(const flute_overrides* o) {
    if (!o) return Ovr{-1, -1, -1, -1, -1, -1, -1, -1, -1, 0};
    return Ovr{o->family, o->m_block, o->waves, o->kw, o->splitk, o->m_tiles, o->slabs_per_wave, o->ring_depth, o->one_shot, 0};
}

constexpr int kMaxLds = 160 * 1024;
constexpr int kFamilyBlock = 3;                 // block-tiled prefill kernel (qgemm_block.h)
constexpr int kFamilySkinny = 5;                // registers-only MFMA kernel for 3 <= M <= 32 (qgemm_skinny.h)
constexpr int kFamilySplitK = 6;                // 128 / 64 x 128 / 64 tiles, K split over workgroups, combined in the launch (qgemm_splitk.h)
constexpr int kFamilyFastM = 7;                 // lean MFMA decode kernel: 4 unit rows x all of K per workgroup, M <= 16 (qgemm_fastm.h)
constexpr int kFamilyPersistM = 8;              // persistent MFMA decode kernel: workgroups stream column-group sets x all of K, M <= 16 (qgemm_persistm.h)
// Workspace layout (every kernel): [0, kXwgFlagBytes) tile state words of the in-launch reductions (xwg.h; zero between
// calls), fp32 slabs behind them.  A planner sees the room behind the state words only.
size_t slab_room(size_t workspace_bytes) { return workspace_bytes > kXwgFlagBytes ? workspace_bytes - kXwgFlagBytes : 0; }
// development A/B (tools/time_cases.py): FLUTE_AMD_B3_TWO_LAUNCH=1 keeps the 3-bit blocks' K slices on fp32 slabs + the reduce launch
bool block3_two_launch() {
    static const bool v = [] { const char* e = getenv("FLUTE_AMD_B3_TWO_LAUNCH"); return e && e[0] == '1'; }();
    return v;
}

int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }
int floor_pow2(int v) { int p = 1; while (p * 2 <= v) p *= 2; return p; }
int ceil_div(int a, int b) { return (a + b - 1) / b; }
// 3-bit skinny blocks (64 rows x the K split the launcher will pick) on at least 80 % of the CUs?  Up to M = 48 the per-wave kernel
// (three row tiles per wave) wins where they do not: 10240 x 8192 M = 48, 40 blocks x 4 slices = 160 workgroups, 44.5 us against 32.4;
// with 224 - 256 workgroups the blocks win by 3 - 24 % (8192^2, 28672 x 8192, 8192 x 28672, 14336 x 4096, 4096 x 14336:
// profiles/r05_planner_regret_between_*.json)
bool skinny3_fills(int M, int col_blocks, int K, int lg, int num_sms) {
    const long tiles = (long)ceil_div(M, 64) * col_blocks;
    const int align_k = std::max(64, 8 << lg);
    long sk = 1;
    while (tiles * sk * 2 <= (long)num_sms && K / (sk * 2) >= std::max(256, align_k)) sk *= 2;
    return tiles * sk * 5 >= (long)num_sms * 4;
}
int round_up(int a, int b) { return ceil_div(a, b) * b; }
// rows of a block-kernel configuration (flute_plan::m_block of family 3): 4 / 5 = 256 / 128 rows,
// 8 + RT = the skinny 3-bit blocks of RT row tiles
int block_rows(int cfg) { return cfg >= 8 ? (cfg - 8) * 16 : ((cfg & 1) ? 128 : 256); }

// Template id -> knobs.  Enumeration order is the reference's
// (flute/codegen_utils.py:110-152): SMs_Multiple {1,2,4} x tile {0,1,2} x
// stages {2,3,4,5} x (b=4 only) QuantMapMode {Vectorized,_32,_16,_8}; tile 0
// has TileP 64, tiles 1,2 TileP 32, so id -> TileP is identical to the
// reference table and reference-packed weights keep their id.
bool decode_template(int bits, int id, flute_template_info* t) {
    if (bits != 2 && bits != 3 && bits != 4) return false;
    const int nq = (bits == 4) ? 4 : 1;
    const int total = 3 * 3 * 4 * nq;
    if (id < 0 || id >= total) return false;
    const int q = id % nq;
    const int st = (id / nq) % 4;
    const int tile = (id / (nq * 4)) % 3;
    const int mult = id / (nq * 12);
    static const int kMult[3] = {1, 2, 4};
    static const int kThreads[3] = {1024, 1024, 512};
    static const int kTileM[3] = {64, 64, 16};
    static const int kTileP[3] = {64, 32, 32};
    static const int kCopies[4] = {1, 32, 16, 8};
    t->num_bits = bits;
    t->template_id = id;
    t->sms_multiple = kMult[mult];
    t->threads = kThreads[tile];
    t->tile_m = kTileM[tile];
    t->tile_k = 64;
    t->tile_p = kTileP[tile];
    t->stages = 2 + st;
    // the reference's QuantMapMode slot; b=2/3 templates have none
    t->lut_copies = (bits == 4) ? kCopies[q] : 32;
    return true;
}

// ---- streaming decode kernel: launch shape ------------------------------------------------------------
// Every (waves per workgroup, in-workgroup K split) that fits LDS is priced by the piece-slots its busiest
// CU executes (1 piece = 1 KiB of one unit's packed row) plus a per-visit overhead, and the candidates are
// ranked; template knob Stages (2..5) picks the best / 2nd / 3rd / 4th, so the tuner's search over ids
// covers the shapes that matter.
struct StreamShape {
    int W, kw, nchunks, kc, kx, upw, ngroups, nwg, visits, s_wave_bytes, s_fast;
    size_t x_off, s_off, red_off, total;
    double cost;
};

bool stream_shape(int bits, int mb, int lg, int units, int krange, int G, bool split_aligned, int num_sms, int W,
                  int kw, int threads_cap, StreamShape* o) {
    const int J = (bits == 3) ? 16 : 16 / bits;
    const int g = 1 << lg;
    const int gpp = 512 >> lg;
    const int lut = stream_lut_bytes(bits);
    const int align_k = std::max(512, 8 * g);                 // K boundaries on 16-B scale granules
    StreamShape s;
    s.W = W; s.kw = kw;
    s.upw = W / kw;
    s.ngroups = ceil_div(units, s.upw);
    bool fits = false;
    for (int nch = 1; nch <= 64; ++nch) {
        int kc = (nch == 1) ? round_up(krange, 512) : round_up(ceil_div(krange, nch), align_k);
        const int real_chunks = ceil_div(krange, kc);
        if (nch > 1 && real_chunks != nch) continue;
        const int pc = ceil_div(std::min(kc, krange), 512);
        const int pk = ceil_div(pc, kw);
        const int ngran = ceil_div(pk * gpp, 8);
        s.s_wave_bytes = ngran * J * 16;
        s.kx = round_up(std::min(kc, krange), 512);
        s.x_off = (size_t)lut;
        s.s_off = s.x_off + (size_t)mb * s.kx * 2;
        s.red_off = s.s_off + (size_t)W * s.s_wave_bytes;
        s.total = s.red_off + 64 + (size_t)2 * W * J * mb * 4;     // arrival counters + two partial-sum buffers
        if (s.total <= (size_t)kMaxLds) {
            s.nchunks = nch; s.kc = kc;
            const bool chunk_ok = nch == 1 || kc % (8 * g) == 0;
            const bool part_ok = kw == 1 || (pk * 512) % (8 * g) == 0;
            s.s_fast = (G % 8 == 0 && split_aligned && chunk_ok && part_ok) ? 1 : 0;
            const int occ = (threads_cap >= 1024 && W <= 8 && s.total <= (size_t)kMaxLds / 2) ? 2 : 1;
            s.nwg = std::min(s.ngroups, num_sms * occ);
            s.visits = ceil_div(s.ngroups, s.nwg);
            const int wgs_cu = ceil_div(s.nwg, num_sms);
            const double per_visit = pk + 1.5 + (kw > 1 ? 2.0 : 0.0) + (s.s_fast ? 0.0 : 4.0);
            double cost = (double)wgs_cu * W * s.visits * nch * per_visit;
            const int waves_cu = wgs_cu * W;
            if (waves_cu < 12) cost *= 1.0 + 0.05 * (12 - waves_cu);   // fewer bytes in flight per CU
            if (nch > 1) cost *= 1.25;                        // activations restaged per visit, barriers
            s.cost = cost;
            fits = true;
            break;
        }
    }
    if (!fits) return false;
    *o = s;
    return true;
}

// ---- one-shot decode kernel (qgemm_oneshot.h): launch shape --------------------------------------------
// A workgroup of W = upw x kw waves owns upw units for the whole of K (no persistence, no grid K split): wave
// (ul, kpart) decodes pk <= D pieces of its unit.  Candidates: D pieces per wave x W in {4, 8, 16}, kw the
// smallest power of two that fits K into kw x D pieces.  Ranked by what tools/ubench/oneshot_lab measured on
// 4096^2 / 4096x11008 (profiles/r03_oneshot_lab.jsonl): launches that give every CU a workgroup first, then
// the smallest in-workgroup K split (kw = 1 needs no cross-wave reduction), then the smallest workgroup that
// keeps the launch at <= 3 workgroups per CU (every workgroup builds its own table image).
// Template knobs: Stages 2..5 -> rank 0..3; QuantMapMode digit (4-bit ids) 1 / 2 -> D = 4 / 8 only.
struct OneShape { int W, kw, upw, pk, depth, pipe, ipw, grid; size_t lds; };
struct OneArgs { int lg, lkw, upw, pk, ipw, depth, pipe, nvis, nch, nwg, nsets; };

int plan_oneshot(int bits, int lg, int M, int N, int K, int num_sms, const flute_template_info& t, int template_id,
                 const Ovr& ov, flute_plan* p, OneArgs* oa) {
    if (lg < 6) return FLUTE_ERR_SHAPE;                       // 32-wide groups: twice the scale words per wave
    if ((K >> lg) & 1) return FLUTE_ERR_SHAPE;                // scale rows are read as aligned dwords (two groups each)
    const int J = (bits == 3) ? 16 : 16 / bits;
    const int units = N / J;
    const int npieces = ceil_div(K, 512);
    int mb = 1; while (mb < M) mb <<= 1;
    if (mb > 4 || (bits == 3 && mb > 2)) return FLUTE_ERR_SHAPE;
    const int dlo = (bits == 3) ? 2 : 4, dhi = 2 * dlo;
    int dsel = 0;                                             // 0: both depths
    if (bits == 4 && (template_id % 4) == 1) dsel = dlo;
    if (bits == 4 && (template_id % 4) == 2) dsel = dhi;
    if (ov.one_shot == 1 && (ov.depth == dlo || ov.depth == dhi)) dsel = ov.depth;
    const int wcap = std::min(t.threads, oneshot_max_threads(bits, mb)) / 64;
    const int runs = oneshot_lut_runs(bits);
    const int xpr = (mb == 4) ? 1 : 2;
    std::vector<OneShape> cands;
    for (int D = dhi; D >= dlo; D /= 2) {
        if (dsel && D != dsel) continue;
        for (int W = 4; W <= 16; W *= 2) {
            int w = W;
            if (ov.waves > 0) { if (W != 4) continue; w = ov.waves; }
            if (w > wcap || w < 1) continue;
            int kw = 1;
            while (ceil_div(npieces, kw) > D) kw *= 2;
            if (ov.kw > 0) kw = floor_pow2(ov.kw);
            if (kw > w || w % kw || ceil_div(npieces, kw) > D) continue;
            OneShape s;
            s.W = w; s.kw = kw; s.upw = w / kw; s.pk = ceil_div(npieces, kw); s.depth = D;
            s.ipw = ceil_div(runs, w);
            if (s.ipw > (bits == 2 ? 16 : 8)) continue;       // table runs a wave writes: at most 8 (2 bits: 16, four per lane group) - more would leave the image incomplete
            if (npieces * 64 > xpr * w * 64) continue;        // activations staged from registers only
            s.grid = ceil_div(units, s.upw);
            s.lds = oneshot_lds_bytes(bits, mb, D, lg, K, w);
            if (s.lds > (size_t)kMaxLds) continue;
            // pipelined loop: one row, every wave holds D whole pieces
            s.pipe = (mb == 1 && s.pk == D && npieces == kw * s.pk && units % s.upw == 0 && K % 512 == 0) ? 1 : 0;
            cands.push_back(s);
        }
    }
    if (cands.empty()) return FLUTE_ERR_SHAPE;
    auto fills = [&](const OneShape& s) { return (long)s.grid * 10 >= (long)num_sms * 9; };
    std::stable_sort(cands.begin(), cands.end(), [&](const OneShape& a, const OneShape& b) {
        if (fills(a) != fills(b)) return fills(a);
        if (!fills(a)) return a.grid > b.grid;
        if (a.kw != b.kw) return a.kw < b.kw;
        const bool many_a = a.grid > 3 * num_sms, many_b = b.grid > 3 * num_sms;      // > 3 workgroups per CU: take the larger workgroup
        if (many_a != many_b) return many_b;
        if (a.W != b.W) return a.W < b.W;
        return a.depth > b.depth;
    });
    size_t pick = std::min((size_t)std::max(0, t.stages - 2), cands.size() - 1);
    if (ov.waves > 0 || ov.kw > 0) pick = 0;
    const OneShape& s = cands[pick];
    p->family = 0;
    p->m_block = mb; p->waves = s.W; p->kw = s.kw; p->splitk = 1; p->k_per_split = K;
    p->grid = (unsigned)s.grid; p->block = (unsigned)(s.W * 64);
    p->lds_bytes = s.lds; p->lut_copies = 32;
    p->ring_depth = s.depth; p->visits = 1; p->k_chunks = 1; p->one_shot = 1 + s.pipe;
    if (oa) { oa->lg = lg; oa->lkw = ilog2(s.kw); oa->upw = s.upw; oa->pk = s.pk; oa->ipw = s.ipw; oa->depth = s.depth; oa->pipe = s.pipe; }
    return FLUTE_OK;
}

// Lean one-row kernel (qgemm_fast.h, round 5): 4 bits, M = 1, K = 512 * D * KW a power of two.  Shapes (waves per workgroup, waves
// per unit row, pieces per wave) by K, first choice first - measured in tools/ubench/oneshot_lab (profiles/r05/fast_lab_run*.jsonl;
// us per launch next to the round-4 one-shot kernel in the same harness): K = 4096: (4, 1, 8) 4.24 / (8, 2, 4) 4.35 against 4.43
// on 4096 x 4096, 7.49 / 7.50 against 8.63 on 4096 x 11008; K = 8192: (8, 2, 8) 6.50 on 8192 x 4096 (g = 128).  rank = the
// template's Stages - 2.
int plan_fast(int bits, int lg, int M, int N, int K, int num_sms, int rank, int want_waves, flute_plan* p, OneArgs* oa) {
    if (bits != 4 || M < 1 || M > 4 || lg < 6 || lg > 8) return FLUTE_ERR_SHAPE;
    int mb = 1; while (mb < M) mb <<= 1;                       // rows per pass: 1, 2, 4 (their activations beside the table image: mb * K * 2 <= 32 KB)
    if ((size_t)mb * K * 2 > 32768) return FLUTE_ERR_SHAPE;
    struct Shape { int W, KW, D; };
    std::vector<Shape> c;
    if (K == 2048) c = {{4, 1, 4}};
    else if (K == 3584) c = {{4, 1, 7}};                       // (Gemma-2-9B: 7 pieces per wave)
    else if (K == 4096) {
        // one round of workgroups (<= one per CU): a wave per unit row, no cross-wave sum (4096^2: 4.05 us against 4.16 with
        // 8 waves); more rounds: two waves per SIMD hide each other's stalls (11008: 6.86 against 6.97, 5120: 4.90 / 5.15,
        // 8192: 5.46 / 5.54 - profiles/r05/time_cases_lean_shapes.jsonl)
        if (N / 16 <= num_sms) c = {{4, 1, 8}, {8, 2, 4}};
        else c = {{8, 2, 4}, {4, 1, 8}};
    } else if (K == 8192) c = {{8, 2, 8}};
    else return FLUTE_ERR_SHAPE;
    Shape sh = c[std::min((size_t)std::max(0, rank), c.size() - 1)];
    if (want_waves > 0) {                                     // override `waves` picks the shape with that many waves
        bool found = false;
        for (const Shape& x : c) if (x.W == want_waves) { sh = x; found = true; break; }
        if (!found) return FLUTE_ERR_SHAPE;
    }
    const int units = N / 4, upw = sh.W / sh.KW;
    if (units % upw) return FLUTE_ERR_SHAPE;
    if (((size_t)N * (size_t)(K >> lg)) * 2 >= (size_t)0xfffffff0u || (size_t)units * K * 2 >= ((size_t)1 << 40)) return FLUTE_ERR_SHAPE;
    (void)num_sms;
    memset(p, 0, sizeof(*p));
    p->family = 0;
    p->m_block = mb; p->waves = sh.W; p->kw = sh.KW; p->splitk = 1; p->k_per_split = K;
    p->grid = (unsigned)(units / upw); p->block = (unsigned)(sh.W * 64);
    p->lds_bytes = fast_lds_bytes(sh.W, sh.KW, sh.D, lg, mb); p->lut_copies = 32;
    p->ring_depth = sh.D; p->visits = 1; p->k_chunks = 1; p->one_shot = 4;
    if (oa) { memset(oa, 0, sizeof(*oa)); oa->lg = lg; oa->lkw = ilog2(sh.KW); oa->upw = upw; oa->pk = sh.D; oa->depth = sh.D; oa->pipe = 1; }
    return FLUTE_OK;
}

// Lean MFMA decode kernel (qgemm_fastm.h, round 5): 4 bits, M <= 16, K = 4096 (8 waves x 512 k) or 2048 (8 x 256 k), a workgroup =
// 4 unit rows (16 columns) x all of K; every wave's K range must hold >= 2 groups (its scale words are read as whole dwords).
// Round 6: ng column groups (4 unit rows each) per workgroup share one staged activation set: the smallest of 1, 2, 3 that covers the
// layer in one round of workgroups (override slabs_per_wave = 1 / 2 / 3 fixes it).
int plan_fastm(int bits, int lg, int M, int N, int K, int num_sms, int ng_ovr, flute_plan* p, OneArgs* oa) {
    if (bits != 4 || M < 1 || M > 16 || lg < 6 || lg > 8 || (K != 4096 && K != 2048)) return FLUTE_ERR_SHAPE;
    const int W = 8, nm = K / (128 * W);
    if (((128 * nm) >> lg) < 2) return FLUTE_ERR_SHAPE;
    const int units = N / 4;
    if (units % 4) return FLUTE_ERR_SHAPE;
    int ng = 1;
    while (ng < 3 && ceil_div(units / 4, ng) > num_sms) ++ng;
    if (ng_ovr >= 1 && ng_ovr <= 3) ng = ng_ovr;
    if (((size_t)N * (size_t)(K >> lg)) * 2 >= (size_t)0xfffffff0u || (size_t)units * K * 2 >= ((size_t)1 << 40)) return FLUTE_ERR_SHAPE;
    memset(p, 0, sizeof(*p));
    p->family = kFamilyFastM;
    p->m_block = 16; p->m_tiles = 1; p->slabs_per_wave = ng; p->waves = W; p->kw = W; p->splitk = 1; p->k_per_split = K;
    p->grid = (unsigned)ceil_div(units / 4, ng); p->block = (unsigned)(W * 64);
    p->lds_bytes = fastm_lds_bytes(K); p->lut_copies = 32;
    p->ring_depth = nm; p->visits = 1; p->k_chunks = 1; p->one_shot = 0;
    if (oa) { memset(oa, 0, sizeof(*oa)); oa->lg = lg; oa->depth = nm; }
    return FLUTE_OK;
}

// Persistent MFMA decode kernel (qgemm_persistm.h, round 6): 4 bits, M <= 16, K a multiple of 128 (>= 1024), group size 64 / 128.  A set =
// ng column groups (16 columns each); the sets are dealt round-robin to `grid` workgroups, `visits` sets each (the last round may be short);
// grid = the fewest workgroups that keep `visits` whole rounds.  ng (override slabs_per_wave) by the model below (ties: the smaller); visits by override m_tiles.
// Model (us; fitted to profiles/r06/call31_persistm_xr.log): the kernel is bound by a wave's in-order issue, not by HBM - a macro-step (128 k
// of ng groups) costs a wave 0.45 / 0.9 / 1.3 us for ng = 1 / 2 / 3 plus 0.04 / 0.135 / 0.35 us for its 1 / 2 / 4 activation requests, the
// busiest workgroup runs `visits` sets of ceil(K / 1024) macro-steps; never below the weights at 5.3 TB/s; 3.5 us fixed.
double persistm_model_us(int M, int N, int K, int num_sms, int ng, int* grid_out, int* visits_out, int visits_ovr) {
    const int groups = N / 16;
    const int nsets = ceil_div(groups, ng);
    int visits = ceil_div(nsets, num_sms);
    if (visits_ovr >= 1) visits = visits_ovr;
    int grid = ceil_div(nsets, visits);
    if (grid > num_sms) { grid = num_sms; visits = ceil_div(nsets, grid); }
    static const double a_ng[4] = {0, 0.45, 0.9, 1.3};
    const double b_x = M <= 4 ? 0.04 : (M <= 8 ? 0.135 : 0.35);
    const double issue = (double)visits * ceil_div(K, 1024) * (a_ng[ng] + b_x);
    const double hbm = ((double)N * K / 2) / 5.3e6;
    if (grid_out) *grid_out = grid;
    if (visits_out) *visits_out = visits;
    return 3.5 + (hbm > issue ? hbm : issue);
}
int plan_persistm(int bits, int lg, int M, int N, int K, int num_sms, int ng_ovr, int visits_ovr, int xres_ovr, flute_plan* p, OneArgs* oa) {
    // (sixteen waves per workgroup - four per SIMD, rings three deep - measured slower than eight on every layer: 28672 x 8192 M = 4 32.1 against
    // 31.1 us, 8192^2 13.3 against 11.1, profiles/r06/call33_persistm_16_waves_dropped.log; the kernel keeps the template parameter)
    // (group size 128: a column's scale row must be a whole number of dwords - the macro-step's 4-B scale request)
    if ((bits != 4 && bits != 2) || M < 1 || M > 16 || lg < 6 || lg > 7 || K % 128 || (lg == 7 && K % 256) || K < 1024 || N % 16) return FLUTE_ERR_SHAPE;
    if ((size_t)N * K * bits / 8 >= (size_t)0xfffffff0u || (size_t)N * (size_t)(K >> lg) * 2 >= (size_t)0xfffffff0u || (size_t)M * K * 2 >= (size_t)0xfffffff0u)
        return FLUTE_ERR_SHAPE;
    // Column groups per set (profiles/r06/planner_regret_persistm*.json).  M <= 8: one, unless a wave's stream is long - visits x macro-steps
    // >= 40 at one group per set (8192 x 28672, 28672 x 8192: two groups 28.4 against 31.1 us, 30.3 against 30.9; 8192^2, 16 macro-steps: one group
    // 10.9 against 12.5, 10240 x 8192 14.1 against 16.7).  M > 8, where every set pulls 16 rows of activations: by the model.
    int ng = 1, grid = 0, visits = 0;
    if (M <= 8) {
        const int groups = N / 16;
        if (groups > num_sms && (long)ceil_div(groups, num_sms) * ceil_div(K, 1024) >= 40) ng = 2;
    } else {
        double best = 0;
        for (int c = 1; c <= 3; ++c) {
            const double us = persistm_model_us(M, N, K, num_sms, c, nullptr, nullptr, visits_ovr);
            if (c == 1 || us < best * 0.98) { best = us; ng = c; }
        }
    }
    if (ng_ovr >= 1 && ng_ovr <= 3) ng = ng_ovr;
    (void)persistm_model_us(M, N, K, num_sms, ng, &grid, &visits, visits_ovr);
    memset(p, 0, sizeof(*p));
    p->family = kFamilyPersistM;
    p->m_block = 16; p->m_tiles = 1; p->slabs_per_wave = ng; p->waves = 8; p->kw = 8; p->splitk = 1; p->k_per_split = K;
    p->grid = (unsigned)grid; p->block = 512u;
    const int xr = M <= 4 ? 1 : (M <= 8 ? 2 : 4);                  // activation requests per macro-step (4 rows each)
    // activations resident in LDS (staged once per workgroup, no activation request in the loop) where 4 xr rows x K fit in 64 KB beside the rest;
    // override one_shot = 0 keeps the rings
    const bool xres = (long)K * xr <= 8192 && xr <= 2 && !(xr == 1 && ng == 3) && xres_ovr != 0;
    p->lds_bytes = persistm_lds_bytes(ng, xr, 8, xres, bits); p->lut_copies = 32;
    p->ring_depth = PM_DW; p->visits = visits; p->k_chunks = xr; p->one_shot = xres ? 1 : 0;
    if (oa) { memset(oa, 0, sizeof(*oa)); oa->lg = lg; oa->depth = PM_DW; }
    return FLUTE_OK;
}

// Persistent one-shot kernel (qgemm_persist.h): W <= 8 waves per workgroup, each wave walks `visits` units of `k_chunks`
// segments (D pieces each; D = 4, 3 bits: 2 - halved while K is not a whole number of segments).  Shapes (W, workgroups
// per CU) are ranked by what tools/decode_lab.py time_persist_shape measured on the 8192 x 28672-class layers
// (profiles/r03/persist_lab_shapes.jsonl): first the unit rows the BUSIEST CU decodes (ceil(grid / CUs) x W x visits: the
// lookups are a per-CU cost - 3-bit layers are bound by it - so a grid of 299 workgroups on 256 CUs costs what 512 would:
// 42 us against 25.7 on 28672 x 8192 W3), then the launch closest to EIGHT waves per CU (7 - 8 waves per CU x 4 visits
// beat 14 - 16 x 2 by 4 - 6 %, 4 per CU lose 20 %), then the larger workgroup (fewer table images).
// Stages 2..5 -> rank 0..3.
int plan_persist(int bits, int lg, int M, int N, int K, int num_sms, const flute_template_info& t, const Ovr& ov,
                 flute_plan* p, OneArgs* oa) {
    if (M < 1 || M > 2 || lg < 6 || ((K >> lg) & 1) || K % 512) return FLUTE_ERR_SHAPE;
    int mb = 1; while (mb < M) mb <<= 1;                      // rows per pass (their staged activations must fit LDS: below)
    const int J = (bits == 3) ? 16 : 16 / bits;
    const int units = N / J;
    const int npieces = K / 512;
    int D = (bits == 3) ? 2 : 4;
    while (D > 2 && npieces % D) D /= 2;
    if (ov.depth == 2 || (ov.depth == 4 && bits != 3)) D = ov.depth;
    const int NS = 2;
    if (npieces % D) return FLUTE_ERR_SHAPE;
    const int nch = npieces / D;
    if (nch > 255) return FLUTE_ERR_SHAPE;
    const int runs = oneshot_lut_runs(bits);
    const int wcap = std::min(t.threads, 512) / 64;
    struct Shape { int W, nwg, nvis; long load, off8; size_t lds; };
    std::vector<Shape> cands;
    for (int W = wcap; W >= 4; --W) {
        if (ov.waves > 0 && W != std::min(ov.waves, wcap)) continue;
        if (bits != 2 && ceil_div(runs, W) > 8) continue;
        const size_t lds = persist_lds_bytes(bits, mb, D, NS, lg, K, W);
        if (lds > (size_t)kMaxLds) continue;
        const int per_cu = std::max(1, std::min((int)((size_t)kMaxLds / lds), 16 / W));
        for (int c = 1; c <= per_cu; ++c) {
            if (ov.m_tiles > 0 && c != ov.m_tiles) continue;  // lab: workgroups per CU
            const int nvis = ceil_div(units, c * num_sms * W);
            const int nwg = ceil_div(units, W * nvis);
            bool dup = false;
            for (const Shape& o : cands) dup = dup || (o.W == W && o.nwg == nwg);
            if (dup) continue;
            cands.push_back(Shape{W, nwg, nvis, (long)ceil_div(nwg, num_sms) * W * nvis,
                                  std::labs((long)ceil_div(nwg, num_sms) * W - 8L), lds});
        }
    }
    if (cands.empty()) return FLUTE_ERR_SHAPE;
    std::stable_sort(cands.begin(), cands.end(), [](const Shape& a, const Shape& b) {
        if (a.load != b.load) return a.load < b.load;
        if (a.off8 != b.off8) return a.off8 < b.off8;
        return a.W > b.W;
    });
    size_t pick = std::min((size_t)std::max(0, t.stages - 2), cands.size() - 1);
    if (ov.waves > 0 || ov.m_tiles > 0) pick = 0;
    const Shape& best = cands[pick];
    p->family = 0;
    p->m_block = mb; p->waves = best.W; p->kw = 1; p->splitk = 1; p->k_per_split = K;
    p->grid = (unsigned)best.nwg; p->block = (unsigned)(best.W * 64);
    p->lds_bytes = best.lds; p->lut_copies = 32;
    p->ring_depth = D; p->visits = best.nvis; p->k_chunks = nch; p->one_shot = 3;
    if (oa) { oa->lg = lg; oa->lkw = 0; oa->upw = best.W; oa->pk = D; oa->ipw = ceil_div(runs, best.W); oa->depth = D; oa->pipe = 1;
              oa->nvis = best.nvis; oa->nch = nch; oa->nwg = best.nwg; oa->nsets = NS; }
    return FLUTE_OK;
}

// Skinny MFMA kernel (qgemm_skinny.h): 4-bit, M <= 16.  A wave = one slab (16 units) x D k-steps, the 4 or 8 waves of a
// workgroup share a K slice of the slab, so K = splitk x 32 D KW with D in {4, 8, 16}; splitk > 1 (round 4): the slices of a
// slab are neighbouring workgroups and meet through the workspace inside the launch (xwg.h, L form).
int plan_skinny(int bits, int lg, int M, int N, int K, const Ovr& ov, size_t workspace_bytes, flute_plan* p, OneArgs* ka) {
    if (bits != 4 || M < 1 || M > 16 || lg < 5 || ((K >> lg) & 1) || K % 128) return FLUTE_ERR_SHAPE;
    const int units = N / 4;
    if (units % 16) return FLUTE_ERR_SHAPE;
    if ((size_t)units * K * 2 >= (size_t)0xfffffff0u || (size_t)(M + 16) * K * 2 >= (size_t)0x7ffffff0u) return FLUTE_ERR_SHAPE;
    const int sk = ov.splitk > 1 ? ov.splitk : 1;
    if (sk > 1) {
        // slabs: 4 KB (four 16 x 16 fp32 tiles in fragment order) per slab and slice
        if (sk > 16 || (K / 32) % sk || units / 16 > kXwgMaxTiles || (size_t)sk * (units / 16) * 4096 > slab_room(workspace_bytes) ||
            (size_t)sk * (units / 16) * 4096 >= ((size_t)1 << 31))
            return FLUTE_ERR_SHAPE;
    }
    const int ksteps = K / 32 / sk;
    int KW = 0, D = 0;
    for (int kw : {8, 4}) {
        if (ov.waves > 0 && kw != ov.waves) continue;
        if (ksteps % kw) continue;
        const int d = ksteps / kw;
        if ((d != 4 && d != 8 && d != 16) || ((d * 32) >> lg) > 8) continue;
        KW = kw; D = d;
        break;
    }
    if (!KW) return FLUTE_ERR_SHAPE;
    p->family = kFamilySkinny;
    p->m_block = 1; p->m_tiles = 1; p->slabs_per_wave = 1; p->waves = KW; p->kw = KW; p->splitk = sk;
    p->k_per_split = K / sk;
    p->grid = (unsigned)(units / 16 * sk); p->block = (unsigned)(KW * 64);
    p->lds_bytes = skinny_lds_bytes(4, 1, KW); p->lut_copies = 32; p->ring_depth = D; p->visits = 1; p->k_chunks = 1; p->one_shot = 0;
    p->splitk_mode = sk > 1 ? 1 : 0;
    p->workspace_needed = sk > 1 ? (size_t)sk * (units / 16) * 4096 + kXwgFlagBytes : 0;
    if (ka) { memset(ka, 0, sizeof(*ka)); ka->lg = lg; ka->lkw = ilog2(KW); ka->ipw = ceil_div(oneshot_lut_runs(4), KW); ka->depth = D; }
    return FLUTE_OK;
}

// Split-K block kernel (qgemm_splitk.h): 128 x 128 tiles x `splitk` K slices, one workgroup each; the slices of a tile are
// neighbours in the block order (dispatched together), the row tiles of a column tile follow each other.  Legal splits:
// K / splitk a multiple of 2 x max(64, group) (two K halves per workgroup, each whole 64-k steps and whole groups), at
// most four 8-group scale blocks per K half, slabs + state words inside the workspace.  Cost model (us, measured on
// MI355X, profiles/r04/splitk_lab*.jsonl): rounds x (fixed + steps x per-step) + seam.
int plan_splitk(int bits, int lg, int M, int N, int K, int num_sms, const Ovr& ov, size_t workspace_bytes, flute_plan* p,
                int rank = 0, double* cost_us = nullptr) {
    if (bits != 2 && bits != 4) return FLUTE_ERR_SHAPE;
    const int g = 1 << lg, G = K >> lg;
    if (G % 8 || N % 128 || K % 128) return FLUTE_ERR_SHAPE;
    if ((size_t)(M + 128) * K * 2 >= (size_t)0xfffffff0u || (size_t)N * G * 2 >= (size_t)0xfffffff0u) return FLUTE_ERR_SHAPE;
    // Row tiles per workgroup: 8 (128-row tiles) or 4 (64-row tiles: twice the tiles, half the slab per slice, every weight
    // dequantised by twice as many workgroups); override m_tiles = 8 / 4 fixes it, else both are priced.  K parts per workgroup
    // (round 6): 2 (128-column tiles: four column groups x two K halves) or - 64-row tiles only - 4 (64-column tiles: two column groups
    // x four K quarters: twice the tiles with NO more K slices, i.e. M = 256 on 4096 x 4096 as 256 workgroups without a seam; twice
    // the activation bytes per workgroup); override kw = 2 / 4 fixes it.
    auto tiles_of = [&](int rt, int kp) { return (long)ceil_div(M, rt * 16) * (N / (256 / kp)); };
    auto slab_of = [&](int rt, int kp) { return (long)rt * 16384 / kp; };      // fp32 partial tile in fragment order: 8 waves x rt / kp row tiles x 2 KB
    auto legal = [&](int sk, int rt, int kp) {
        const int align = kp * std::max(64, g);                // every K part: whole 64-k steps and whole groups
        if (sk < 1 || sk > 16 || K % sk || (K / sk) % align) return false;
        if (tiles_of(rt, kp) > kXwgMaxTiles) return false;
        const int gw = (K / sk) >> lg;                         // groups of a workgroup's K range: one scale image of eight 8-group blocks per column group
        if (gw + ((gw % 8) ? 7 : 0) > 64) return false;
        const size_t slabs = (size_t)sk * tiles_of(rt, kp) * slab_of(rt, kp);
        if (sk > 1 && (slabs > slab_room(workspace_bytes) || slabs >= ((size_t)1 << 31))) return false;
        return true;
    };
    // us, fitted to tools/splitk_lab.py on MI355X (profiles/r04/splitk_lab_run8*.jsonl, run9*): a round of workgroups costs ~9.5 us
    // of launch, prologue, K-half exchange, ramp and stores + per 64-k step 0.65 .. 0.96 us (128-row tiles: 30.3 us on 64 CUs,
    // 32.9 on 172, 39.3 on 224, 40.3 on 256 at K = 4096 - the more of the chip is busy the slower) or 0.44 .. 0.52 us (64-row
    // tiles: 23.9 us on 128 CUs, 26.5 on 256); the seam grows with the MB published write-through (E form at 2 slices
    // ~1 + 0.15 / MB, 4 slices and the L form ~1.5 + 0.5 / MB)
    auto model_us = [&](int sk, int rt, int kp) {
        const long wgs = tiles_of(rt, kp) * sk;
        // a last, partly filled round costs less than a whole one (round 6: M = 384 on 8192^2, 128-row tiles x 2 slices = 1.5 rounds: 68 us
        // measured, 83 priced at two whole rounds - and the cheaper-looking 64-row plan, three whole rounds, ran 81): the mean of
        // the whole rounds and the exact share (profiles/r06_planner_regret_between_final.json)
        const double whole = (double)((wgs + num_sms - 1) / num_sms);
        const double rounds = wgs <= (long)num_sms ? 1.0 : 0.5 * (whole + (double)wgs / num_sms);
        const double fill = std::min(1.0, (double)wgs / num_sms);
        const double steps = (double)K / sk / (64.0 * kp);     // 64-k steps of a K part
        const int hr = rt / kp;
        const bool eform = (sk == 2 && hr % 2 == 0) || (sk == 4 && hr % 4 == 0);
        const double mb = sk == 1 ? 0.0 : (double)wgs * (slab_of(rt, kp) * 1e-6) * (eform ? (sk - 1.0) / sk : 1.0);
        // (L form on 32-KB slabs, measured at M = 48 .. 96: 2.2 us at four slices / 3.4 at eight of 8.4 MB - profiles/r04/splitk_64_row_tiles_below_m128.json)
        const double seam = sk == 1 ? 0.0 : (sk == 2 && eform ? 1.0 + 0.15 * mb : (rt == 4 ? 1.5 + 0.15 * mb : 1.5 + 0.5 * mb));
        const double step = rt == 8 ? 0.65 + 0.31 * fill * fill * fill : 0.44 + 0.08 * fill * fill * fill;
        // four K parts (64 x 64 tiles, round 6; tools/splitk_lab.py time_kp4, profiles/r06/call1_*.log, call16_*.log): a round costs 6.0 us
        // + 0.6 us per step whatever share of the chip is busy (16 steps: 15.5 us on 256 CUs, 15.4 on 128, 15.5 on 64; 32 steps 25.1;
        // 8 steps + the L-form seam of two slices 12.6)
        if (kp == 4) return rounds * (6.0 + steps * 0.6) + seam;
        return rounds * (9.5 + steps * step) + seam;
    };
    struct Cand { double us; int sk, rt, kp; };
    std::vector<Cand> c;
    for (int kp : {2, 4}) {
        if ((ov.kw == 2 || ov.kw == 4) && kp != ov.kw) continue;
        for (int rt : {8, 4}) {
            if ((ov.m_tiles == 8 || ov.m_tiles == 4) && rt != ov.m_tiles) continue;
            if (kp == 4 && rt != 4) continue;                  // four K parts: 64-row tiles only (LDS)
            for (int sk = 1; sk <= 16; ++sk) {
                if (ov.splitk > 0 && sk != ov.splitk) continue;
                if (legal(sk, rt, kp)) c.push_back(Cand{model_us(sk, rt, kp), sk, rt, kp});
            }
        }
    }
    if (c.empty()) return FLUTE_ERR_SHAPE;
    std::stable_sort(c.begin(), c.end(), [](const Cand& x, const Cand& y) { return x.us < y.us; });
    const Cand best = c[std::min((size_t)std::max(0, rank), c.size() - 1)];
    if (cost_us) *cost_us = best.us;
    memset(p, 0, sizeof(*p));
    p->family = kFamilySplitK;
    // four loader waves beside the eight compute waves (round 4: K = 4096 per workgroup 34.1 -> 30.3 us on 64 CUs, 35.2 -> 32.9 on
    // 172, 41.8 -> 40.3 on 256; the variant without them - override waves = 8 - was dropped in round 6)
    if (ov.waves > 0 && ov.waves != 12) return FLUTE_ERR_SHAPE;
    const int ldw = SK_LOADERS;
    // m_block: row tiles of a column tile per XCD group (block order of launches without K slices); override 1 / 2 / 4 / 8
    {
        const int tm = ceil_div(M, best.rt * 16);
        // (round 6: as many of a column tile's row tiles as divide them, up to eight - M = 256 on 4096^2, four 64-row tiles: 16.4 / 15.9 / 15.6 us
        // at 1 / 2 / 4 per group, profiles/r06/call20_*.log; M = 512 on 2048 x 4096: 16.4 / 16.3 / 16.1 / 16.1 at 1 / 2 / 4 / 8)
        int E = 8;
        while (E > 1 && (tm % E || ((tm / E) & (tm / E - 1)))) E >>= 1;
        if (ov.m_block == 1 || ov.m_block == 2 || ov.m_block == 4 || ov.m_block == 8) E = ov.m_block;
        if (best.sk != 1 || tm % E || ((tm / E) & (tm / E - 1))) E = 1;
        p->m_block = E;
    }
    p->m_tiles = best.rt; p->slabs_per_wave = 1; p->waves = 8 + ldw; p->kw = best.kp;
    p->splitk = best.sk; p->k_per_split = K / best.sk;
    p->grid = (unsigned)(tiles_of(best.rt, best.kp) * best.sk); p->block = (unsigned)(512 + 64 * ldw);
    p->lds_bytes = (size_t)splitk_lds_bytes(bits, best.rt, best.kp); p->lut_copies = 32;
    p->splitk_mode = best.sk > 1 ? 1 : 0;
    p->workspace_needed = best.sk > 1 ? (size_t)best.sk * tiles_of(best.rt, best.kp) * slab_of(best.rt, best.kp) + kXwgFlagBytes : 0;
    return FLUTE_OK;
}

int plan_stream(int dtype, int bits, int lg, int M, int N, int K, int num_sms, const flute_template_info& t,
                const Ovr& ov, size_t workspace_bytes, flute_plan* p, StreamArgs* sa) {
    const int J = (bits == 3) ? 16 : 16 / bits;
    const int units = N / J;
    const int G = K >> lg;
    int mb = 1; while (mb < M) mb <<= 1;
    const int dec_max = 4;
    if (ov.m_block > 0 && ov.m_block >= M && ov.m_block <= dec_max) mb = floor_pow2(ov.m_block);
    int depth = (bits == 3) ? 2 : (t.sms_multiple == 1 ? 4 : 2);
    if (ov.depth == 2 || (ov.depth == 4 && bits != 3)) depth = ov.depth;
    const int threads_cap = std::min(t.threads, stream_max_threads(bits, mb, depth));
    const int wcap = threads_cap / 64;

    // grid-level K split only for layers too narrow to give every CU eight waves even at the deepest
    // in-workgroup split
    int splitk = 1;
    while ((long)units * 16 * splitk < (long)num_sms * 8 && K / (splitk * 2) >= 1024) splitk *= 2;
    if (ov.splitk > 0) splitk = ov.splitk;
    const int align_k = std::max(512, 8 << lg);
    int kps = round_up(ceil_div(K, splitk), align_k);
    splitk = ceil_div(K, kps);
    while (splitk > 1 && (size_t)splitk * M * N * 4 > workspace_bytes) {
        splitk >>= 1;
        kps = round_up(ceil_div(K, splitk), align_k);
        splitk = ceil_div(K, kps);
    }
    if (splitk == 1) kps = round_up(K, 512);
    const int krange = std::min(K, kps);
    const bool split_aligned = splitk == 1 || kps % (8 << lg) == 0;

    std::vector<StreamShape> cands;
    for (int W = 1; W <= wcap; ++W) {
        if (ov.waves > 0 && W != std::min(ov.waves, wcap)) continue;
        for (int kw = 1; kw <= W; kw <<= 1) {
            if (W % kw) continue;
            if (ov.kw > 0 && kw != std::min(floor_pow2(ov.kw), floor_pow2(W))) continue;
            StreamShape s;
            if (stream_shape(bits, mb, lg, units, krange, G, split_aligned, num_sms, W, kw, threads_cap, &s))
                cands.push_back(s);
        }
    }
    if (cands.empty()) return FLUTE_ERR_SHAPE;
    // shapes of fewer than 8 waves are only worth ranking when nothing larger exists
    if (ov.waves <= 0) {
        const int wmin = std::min(8, wcap);
        bool any = false;
        for (const StreamShape& c : cands) any = any || c.W >= wmin;
        if (any) cands.erase(std::remove_if(cands.begin(), cands.end(), [&](const StreamShape& c) { return c.W < wmin; }),
                             cands.end());
    }
    std::stable_sort(cands.begin(), cands.end(), [](const StreamShape& a, const StreamShape& b) {
        if (a.cost != b.cost) return a.cost < b.cost;
        if (a.W != b.W) return a.W > b.W;
        return a.kw < b.kw;
    });
    // Stages 2..5 -> rank 0..3 among shapes that differ in (W, kw); explicit overrides leave one candidate
    size_t pick = std::min((size_t)std::max(0, t.stages - 2), cands.size() - 1);
    if (ov.waves > 0 || ov.kw > 0) pick = 0;
    const StreamShape& s = cands[pick];

    p->family = 0;
    p->m_block = mb; p->waves = s.W; p->kw = s.kw; p->splitk = splitk;
    p->k_per_split = (splitk == 1) ? K : kps;
    p->grid = (unsigned)(s.nwg * splitk);
    p->block = (unsigned)(s.W * 64);
    p->lds_bytes = s.total;
    p->lut_copies = 32;
    p->ring_depth = depth; p->visits = s.visits; p->k_chunks = s.nchunks;
    p->one_shot = 0;
    if (sa) {
        memset(sa, 0, sizeof(*sa));
        sa->M = M; sa->N = N; sa->K = K; sa->G = G; sa->lg = lg;
        sa->units = units; sa->ngroups = s.ngroups;
        sa->upw = s.upw; sa->kw = s.kw; sa->lkw = ilog2(s.kw);
        sa->nwg = s.nwg; sa->vis_q = s.ngroups / s.nwg; sa->vis_r = s.ngroups % s.nwg;
        sa->splitk = splitk; sa->k_per_split = kps;
        sa->kc = s.kc; sa->nchunks = s.nchunks; sa->kx = s.kx;
        sa->x_off = (int)s.x_off; sa->s_off = (int)s.s_off; sa->red_off = (int)s.red_off;
        sa->s_wave_bytes = s.s_wave_bytes; sa->s_fast = s.s_fast;
    }
    (void)dtype;
    return FLUTE_OK;
}

// Per-wave MFMA kernel (2 / 4 bits), TFLOP/s by batch size, 128 <= M < 512 (end of round 6, profiles/r06/planner_regret_bits_4_2_m128_to_512_end_of_round.json:
// measured 295 / 345 at M = 160 / 192 on 11008 x 4096, 426 / 493 at M = 320 / 384 on 4096^2 - the "520 at M = 256, 465 below" it replaces kept the per-wave
// kernel at M = 160 / 192 / 320 on 11008 x 4096, 14336 x 4096, 14336 x 3584, 3584 x 14336, 3584 x 8192 where split-K tiles run 30 - 70 % faster)
double wave_tf_mid(int M, int bits, bool bf) { return (300.0 + 0.75 * (M - 128)) * (bf ? 0.8 : 1.0) * (bits == 2 ? 0.65 : 1.0); }

// ids whose last digit leaves the kernel choice to the planner: 4-bit QuantMapMode digit 0, every 2- / 3-bit id
bool auto_digit_sk(int bits, int template_id) { return bits != 4 || (template_id % 4) == 0; }

int make_plan_uncached(int dtype, int bits, int group, int M, int N, int K, int template_id, int num_sms,
              size_t workspace_bytes, const Ovr& ov, flute_plan* p, flute_template_info* tinfo,
              StreamArgs* sa, OneArgs* oa) {
    if (dtype != 0 && dtype != 1) return FLUTE_ERR_DTYPE;
    // override families: -1 automatic, 0 decode, 1 / 2 per-wave MFMA kernel, 3 block kernels, 5 skinny MFMA kernel,
    // 6 split-K block kernel, 7 lean MFMA decode kernel, 8 persistent MFMA decode kernel; anything else is a caller error (round 1's family 4 is gone)
    if (ov.family < -1 || ov.family == 4 || ov.family > 8) return FLUTE_ERR_SHAPE;
    if (bits != 2 && bits != 3 && bits != 4) return FLUTE_ERR_NUM_BITS;
    if (group != 32 && group != 64 && group != 128 && group != 256) return FLUTE_ERR_GROUP_SIZE;
    flute_template_info t;
    if (!decode_template(bits, template_id, &t)) return FLUTE_ERR_TEMPLATE_ID;
    if (bits == 3 && t.tile_p != 32) return FLUTE_ERR_TEMPLATE_ID;   // utils.py:137-139
    if (tinfo) *tinfo = t;
    const int J = (bits == 3) ? 16 : 16 / bits;
    if (M < 1 || N < 1 || K < 1) return FLUTE_ERR_SHAPE;
    if (N % (J * t.tile_p) || K % 64 || K % group) return FLUTE_ERR_SHAPE;
    if (num_sms < 1) num_sms = 256;
    const int lg = ilog2(group);
    const int units = N / J;

    memset(p, 0, sizeof(*p));

    // Streaming decode kernel: M <= 2; its four-row variant (2- / 4-bit) only on request (override family 0, which
    // flute_qgemm_hadamard sets for small layers so that the rotation stays fused): measured at M = 3, 4 the MFMA
    // kernel is as fast on 4096^2 (7.6 vs 7.75 us) and 15-25 % faster on every larger layer (8192x28672: 38.6 vs 48.9).
    // (3-bit layers up to 24 M weights also take it by themselves: their MFMA plans are 256 columns wide per wave -
    // 4096^2 M = 4: 9.9 against 13.7 us; larger 3-bit layers are faster on the MFMA kernel.)
    const int dec_max = 4;
    const bool small_b3 = bits == 3 && ov.family < 0 && (size_t)N * K <= ((size_t)24 << 20);
    // round 4: 2- / 4-bit layers up to 16 M weights take the four-row one-shot kernel at M = 3, 4 too (4096^2 M = 4: 6.3 us
    // against 7.1 on the MFMA kernel, profiles/r04_planner_regret_before_fixes.json - the one-shot kernel of round 3 was not
    // there when the MFMA kernel was measured level with the ring kernel)
    // (only ids whose last digit leaves the choice to the planner: a 4-bit id with QuantMapMode digit 3 keeps the skinny MFMA kernel it was
    // tuned on - "a tuned id keeps the kernel it was timed on", tests/test_abi.py)
    const bool small_b24 = bits != 3 && ov.family < 0 && M >= 3 && (size_t)N * K <= ((size_t)16 << 20) && auto_digit_sk(bits, template_id);
    const bool auto_digit = (bits == 4) ? (template_id % 4) == 0 : t.sms_multiple == 1;
    // decode planners: a fused Hadamard rotation prefers 8-wave workgroups (4096x3584 M = 1: 5.5 us with 8 waves, 6.5 with
    // the 4-wave shape the plain product takes)
    Ovr ovd = ov;
    if (ov.had8 && ovd.waves < 0 && ovd.kw < 0 && ovd.one_shot != 0) ovd.waves = 8;
    auto persist_auto_ok = [&](int rows) {
        if (ov.one_shot >= 0 || ov.depth > 0 || ov.splitk > 1 || !auto_digit) return false;
        if ((size_t)N * K < ((size_t)24 << 20) || (long)units < 4L * num_sms) return false;     // (round 4: from 40 M weights / 6 unit rows per CU - 6144 x 4096 M = 2 6.8 -> 6.0 us, 2-bit 10240 x 8192 M = 1 10.8 -> 9.8)
        flute_plan tmp;
        memset(&tmp, 0, sizeof(tmp));
        return plan_persist(bits, lg, rows, N, K, num_sms, t, ovd, &tmp, nullptr) == FLUTE_OK;
    };
    int family = (M <= 2 || (M <= dec_max && (ov.family == 0 || small_b3 || small_b24))) ? 0 : 2;
    if (ov.family >= 1) family = 2;               // any M may be forced through the MFMA kernel
    // Skinny MFMA kernel (qgemm_skinny.h): by override (family 5), by template (4-bit QuantMapMode digit 3 at M <= 16, where
    // the digit's other meaning - two slabs per wave - does not exist; digit 2: never), or automatically (digit 0) for
    // 3 <= M <= 16 on layers whose slabs (64 columns) fill 55 .. 100 % of the CUs in ONE round - a workgroup pulls its slab's
    // weights AND all of X through one CU, so fewer slabs leave CUs idle; wider layers take the per-wave kernel with two
    // slabs per wave (4096 x 28672: 21.1 us against 23.3 here).  Measured (profiles/r03/skinny_lab.jsonl, M = 16, us, per-wave kernel -> skinny):
    // 4096 x 11008 17.9 -> 12.0, 4096 x 14336 18.3 -> 12.6 (M = 4: 17.6 -> 11.8), 4096 x 28672 26.6 -> 23.3, 4096 x 6144
    // 13.1 -> 11.7, 2048 x 8192 7.6 -> 7.0; 4096 x 8192 10.0 -> 11.8 and 4096^2 7.6 -> 11.6 (not taken).
    // Lean MFMA decode kernel (qgemm_fastm.h, round 5): by override (family 7), or automatically for 5 <= M <= 16 rows of a 4-bit layer
    // with K = 2048 / 4096 that ONE round of its workgroups (4 unit rows each) covers, under the ids whose last digit leaves the
    // choice to the planner (QuantMapMode digit 0, SMs_Multiple 1).  Measured (tools/time_cases.py, us, automatic plan of round 4 ->
    // this kernel; profiles/r05/time_cases_fastm*.jsonl): 4096^2 M = 5 / 8 / 16 7.13 / 7.13 / 7.18 -> 5.9 / 5.9 / 6.1, 4096 x 2048
    // 5.7 / 5.7 / 5.8 -> 4.3 / 4.3 / 4.5, 2048 x 4096 7.6 / 7.6 / 7.7 -> 5.4 / 5.4 / 5.6; not taken: more than one round (8192 x 4096:
    // 10.2 .. 10.8 against the skinny kernel's 8.9 .. 9.6: every workgroup pulls all of X through its CU), M <= 4 (the dot-product
    // lean kernel: 5.05 against 5.9)
    // Round 6: two / three column groups per workgroup on ONE staged activation set make wider layers one round of workgroups
    // (profiles/r06/call18_fastm_column_groups.log, M = 16, us, table's plan -> this): 8192 x 4096 10.1 -> 8.9 (2 groups, 256 workgroups),
    // 11008 12.6 -> 11.4 (3 groups, 230), 6144 9.4 -> 8.4, 5120 9.2 -> 8.0, 8192 x 2048 6.9 -> 6.2; M = 8 on 11008 11.6 -> 11.0; not taken:
    // more than one round even at three groups (14336: 16.0 against 13.1; 28672: 30 against 23)
    const long fm_groups = N / 16;
    const long fm_wgs = fm_groups <= (long)num_sms ? fm_groups : (fm_groups <= 2L * num_sms ? (fm_groups + 1) / 2 : (fm_groups + 2) / 3);
    const bool fastm_auto = ov.family < 0 && bits == 4 && M >= 5 && M <= 16 && (template_id % 4) == 0 && t.sms_multiple == 1 &&
                            (K == 4096 || K == 2048) && fm_wgs <= (long)num_sms && fm_wgs * 2 >= (long)num_sms &&
                            ov.m_tiles < 0 && ov.waves < 0 && ov.kw < 0 && ov.splitk < 0 && ov.slabs < 0 && ov.m_block < 0;
    // Persistent MFMA decode kernel (qgemm_persistm.h, round 6): by override (family 8), or automatically - under the ids that leave the choice
    // to the planner, as the lean MFMA decode kernel - for 3 <= M <= 16 rows of a 4-bit layer with K >= 6144 (group size 64 / 128).  Measured (profiles/r06/call31_persistm_xr.log,
    // us, table's plan -> this): M = 4: 8192 x 28672 [N x K] 34.6 -> 29.4, 10240 x 8192 20.6 -> 14.3, 4096 x 11008 14.7 -> 9.2, 3584 x 14336 14.6 -> 10.0,
    // 4096 x 14336 14.7 -> 10.5, 8192^2 12.9 -> 11.7, 28672 x 8192 33.0 -> 31.1; M = 16: 10240 x 8192 22.2 -> 17.4, 4096 x 11008 14.9 -> 11.9,
    // 8192^2 15.8 -> 13.9; not taken: K = 4096 (14336 x 4096: 11.1 against 11.5 .. 12.4; the lean MFMA decode kernel's and the skinny kernel's
    // layers)
    if (ov.family == kFamilyPersistM) return plan_persistm(bits, lg, M, N, K, num_sms, ov.slabs, ov.m_tiles, ov.one_shot, p, oa);
    // (second sweep, profiles/r06/planner_regret_persistm*.json: also K >= 3584 at M <= 4 - 14336 x 3584 13.2 -> 10.3, 11008 x 4096 11.1 -> 9.6,
    // 6144 x 4096 8.3 -> 7.6 - and at every M <= 16 where K is neither 2048 nor 4096, the lean MFMA decode / skinny kernels' depths -
    // 14336 x 3584 M = 8 13.2 -> 11.9; and no limit on the activations at M > 8: equal at M = 16 on the 28672-wide / -deep layers, 7 - 15 % faster at M = 11)
    // (third step, profiles/r06/call37_persistm_resident_activations.log: with the activations RESIDENT in LDS - K * ceil(M / 4) rows within 64 KB - also
    // K = 4096 at M <= 8: 4096^2 M = 8 5.9 -> 5.5, 11008 x 4096 10.9 -> 9.3, 14336 x 4096 11.4 -> 11.3)
    const bool pm_k = K >= 6144 || (K >= 3584 && (M <= 8 || (K != 4096 && K != 2048)));
    // 2-bit member (profiles/r06/call38_persistm_2bit.log, us, table's plan -> this): every 2-bit id leaves the choice to the planner and no lean kernel
    // competes - from K = 3584 and 16 M weights at every 3 <= M <= 16: 4096^2 M = 4 / 16 8.4 / 8.6 -> 5.0 / 6.1, 4096 x 11008 14.1 -> 8.1, 10240 x 8192 18.3 -> 13.8,
    // 8192 x 28672 30.7 -> 25.8; above M = 8 not where the busiest workgroup pulls more than 28672 k of sixteen-row activations (28672 x 8192: 33.4 against 31.5)
    const bool pm4 = bits == 4 && (template_id % 4) == 0 && t.sms_multiple == 1 && pm_k && (size_t)N * K + (M >= 5 ? 1 : 0) > ((size_t)16 << 20);
    const bool pm2 = bits == 2 && K >= 3584 && (size_t)N * K >= ((size_t)16 << 20);
    if (ov.family < 0 && (pm4 || pm2) && M >= 3 && M <= 16 && (lg == 6 || lg == 7) &&
        (long)(N / 16) * 2 >= (long)num_sms &&      // (smaller layers: the decode kernels / too few sets for the chip; 4-bit 4096^2 itself from M = 5)
        ov.m_tiles < 0 && ov.waves < 0 && ov.kw < 0 && ov.splitk < 0 && ov.slabs < 0 && ov.m_block < 0 && ov.one_shot < 0 && ov.depth <= 0) {
        flute_plan pm;
        OneArgs pm_oa;
        if (plan_persistm(bits, lg, M, N, K, num_sms, -1, -1, -1, &pm, &pm_oa) == FLUTE_OK && (bits == 4 || M <= 8 || (long)pm.visits * K <= 28672)) {
            *p = pm;
            if (oa) *oa = pm_oa;
            return FLUTE_OK;
        }
    }
    if (ov.family == kFamilyFastM || fastm_auto) {
        if (plan_fastm(bits, lg, M, N, K, num_sms, ov.family == kFamilyFastM ? ov.slabs : -1, p, oa) == FLUTE_OK) return FLUTE_OK;
        memset(p, 0, sizeof(*p));
    }
    {
        const int q4 = (bits == 4) ? template_id % 4 : -1;
        const long slabs5 = units / 16;
        const bool fill5 = slabs5 * 20 >= 11L * num_sms && slabs5 <= num_sms;
        const bool auto5 = ov.family < 0 && family == 2 && bits == 4 && M >= 3 && M <= 16 &&
                           ((q4 == 0 && K >= 4096 && fill5) || q4 == 3);
        // Round 4: narrower layers through a grid-level K split (the slices of a slab meet inside the launch, xwg.h: 1.5 - 1.7 us
        // of seam, profiles/r04/xwg_seam_price.json) - the smallest power-of-two split that fills 55 % of the CUs, while a
        // slice keeps >= 2048 k.  Measured (profiles/r04_planner_regret_before_fixes.json, us, per-wave kernel -> split skinny):
        // M = 4: 3584 x 8192 10.1 -> 8.7, 8192^2 14.6 -> 13.0, 6144 x 4096 9.2 -> 8.5; M = 16: 3584 x 8192 10.2 -> 9.6; not
        // taken: 4096^2 (four slices of 1024 k: 7.6 against 7.2).
        int sk5 = 0;
        if (ov.family < 0 && ov.splitk < 0 && family == 2 && bits == 4 && M >= 3 && M <= 16 && (q4 == 0 || q4 == 3) && K >= 4096 && !fill5 &&
            slabs5 * 20 < 11L * num_sms) {
            int sk = 2;
            while (sk < 16 && slabs5 * sk * 20 < 11L * num_sms) sk *= 2;
            if (slabs5 * sk <= num_sms && K / sk >= 2048) sk5 = sk;
        }
        if (ov.family == kFamilySkinny || auto5 || sk5) {
            Ovr o5 = ov;
            if (sk5) o5.splitk = sk5;
            if (plan_skinny(bits, lg, M, N, K, o5, workspace_bytes, p, oa) == FLUTE_OK) return FLUTE_OK;
            memset(p, 0, sizeof(*p));
        }
    }
    // Split-K block kernel (qgemm_splitk.h): by override (family 6); by template - Stages 5 of the automatic digit at
    // M >= 128 (SMs_Multiple 1 / 2 / 4: the cost model's best / second / third K split; those ids' old meaning, a quarter of
    // the per-wave kernel's in-workgroup K split, is still reachable through digits 1 .. 3); automatically below, when
    // its modelled time beats what the other MFMA kernels are modelled at
    if (ov.family == kFamilySplitK) {
        const int src = plan_splitk(bits, lg, M, N, K, num_sms, ov, workspace_bytes, p);
        if (src == FLUTE_OK && p->lds_bytes > (size_t)kMaxLds) return FLUTE_ERR_SHAPE;
        return src;
    }
    // (round 4, late: 4-bit layers from M = 33, 2-bit layers from M = 65 - 64-row tiles x K slices against the per-wave kernel:
    // M = 64 x 8192^2 24.7 -> 20.0 us, M = 33 23.7 -> 19.5, M = 96 x 14336 x 4096 41.2 -> 24.6, M = 96 x 8192^2 45.3 -> 28.6;
    // 2 bits M = 96: 8192^2 43.3 -> 29.0, 14336 x 4096 40.5 -> 25.4; 2 bits at M = 64 gain 3 .. 9 % only: left alone)
    // (overrides of the per-wave kernel - m_tiles, waves, kw, splitk, slabs - mean something else in plan_splitk: a call that sets
    // any of them without family = 6 keeps the per-wave / block kernels)
    const bool wave_ovr = ov.m_tiles > 0 || ov.waves > 0 || ov.kw > 0 || ov.splitk > 0 || ov.slabs > 0 || ov.m_block > 0;
    const bool sk_regime = ov.family < 0 && !wave_ovr && family == 2 && bits != 3 && (M >= 128 || M >= 33) && auto_digit_sk(bits, template_id);      // (2 bits: from M = 65 until round 6 - with 64 x 64 tiles M = 48 / 64 gain 25 - 43 %: profiles/r06/planner_regret_before_fixes.json)
    if (sk_regime && t.stages == 5 && M > 64) {       // (M <= 64: the table's Stages-5 ids of that bucket were tuned on the per-wave kernel's K split)
        if (plan_splitk(bits, lg, M, N, K, num_sms, ov, workspace_bytes, p, t.sms_multiple == 1 ? 0 : (t.sms_multiple == 2 ? 1 : 2)) == FLUTE_OK)
            return FLUTE_OK;
        memset(p, 0, sizeof(*p));
    }
    // Block-tiled prefill kernels (qgemm_block2.h: 256 x 256 or 128 x 256 blocks, a wave owns all rows and 32
    // columns, 4- and 2-bit layers; qgemm_block3.h: the same for 3 bits, 128-row blocks); scale rows in
    // whole 16-B granules.  No K split for 2 / 4 bits (qgemm_splitk.h serves that regime; 3 bits: priced below), so what decides is
    // how many blocks the output has.  Cost model fitted to tools/block_lab.py (MI355X, K = 4096; us per block,
    // running alone / with the whole chip busy - the chip clocks down under a full MFMA load):
    //   256-row block fp16 100 / 126, bf16 104 / 129;  128-row block fp16 72 / 81, bf16 80 / 89 (round 2);
    //   round 4 (2- / 4-bit blocks: whole-line activation pieces, one whole-line weight request per step):
    //   256-row 97 / 120, bf16 101 / 124;  128-row 62 / 74, bf16 70 / 78 (profiles/r04/splitk_lab_run7*.jsonl);
    //   per-wave MFMA kernel (family 2): 520 ... 730 TFLOP/s fp16, 400 ... 560 bf16 for M = 256 ... 4096.
    int blk_cfg = -1, blk_sk = 0;                     // blk_sk: grid K split the cost model chose with the block shape (3 bits)
    size_t blk_slabs = 0;                             // bytes of fragment-order slabs when a 3-bit block plan combines its K slices in the launch
    double alt_us = -1.0;                             // modelled time of the best other MFMA kernel (set by the block cost model)
    const int blk_units = 256 / J;                    // units of a 256-column block (4-bit: 64, 2-bit: 32, 3-bit: 16)
    const bool b3_ok = bits != 3 || (size_t)3 * (N >> 4) * K * 2 < (size_t)0xfffffff0u;   // one descriptor over Q
    const bool x32_ok = (size_t)(M + 256) * K * 2 < (size_t)0xfffffff0u;       // activation byte offsets are 32-bit voffsets
    if (b3_ok && x32_ok && (K >> lg) % 8 == 0 && units % blk_units == 0 && K % 64 == 0 &&
        (family == 2 || ov.family == kFamilyBlock) && (ov.family < 0 || ov.family == kFamilyBlock)) {
        const long tiles256 = (long)ceil_div(M, 256) * (units / blk_units), tiles128 = (long)ceil_div(M, 128) * (units / blk_units);
        if (ov.family == kFamilyBlock) {
            blk_cfg = (ov.m_tiles == 4) ? 5 : 4;             // 128- / 256-row blocks of qgemm_block2.h
            if (bits == 3 && ov.m_tiles != 8) blk_cfg = 5;   // 3-bit layers: 128-row blocks of qgemm_block3.h unless 256 rows are asked for ...
            if (bits == 3 && (ov.m_block == 1 || ov.m_block == 2 || ov.m_block == 4))
                blk_cfg = 8 + ov.m_block;                    // ... or its skinny blocks of m_block row tiles
        } else if (bits == 3 && M > 32 && M <= 64 && (size_t)N * K >= ((size_t)56 << 20) &&
                   (M > 48 || skinny3_fills(M, units / blk_units, K, lg, num_sms))) {    // (14336 x 3584, 51 M weights: 29.5 against 26.4 us on the per-wave kernel)
            // 3-bit skinny blocks (64 rows, grid K split): measured against the per-wave kernel at M = 64 - 8192^2 31.6 vs
            // 38.5 us, 28672x8192 86.9 vs 105.7, 4096x14336 31.8 vs 35.4; slower below M = 33 and on 4096^2 (fixed
            // costs of ~8 us per call: prologue, fp32 slabs, reduce launch)
            blk_cfg = 12;
        } else if (M >= 256 || (bits == 3 && M > 64) || (bits != 3 && M > 128)) {          // (3 bits from M = 65: the K-split candidates below; 2 / 4 bits from M = 129: 128-row blocks on the
            // widest layers - 2-bit 28672 x 8192 at M = 192 ran 210 us on the per-wave kernel, 125 on 128-row blocks, profiles/r06/planner_regret_bits_4_2_m96_to_768_unswept_batch_sizes.json)
            const bool bf = dtype == FLUTE_BF16;
            auto block_us = [&](long tiles, double alone, double busy) {
                const long whole = tiles / num_sms, rest = tiles % num_sms;       // full rounds + a last partial one
                const double last = rest == 0 ? 0.0 : (rest * 4 >= (long)num_sms * 3 ? busy : alone);
                return ((double)whole * busy + last) * (double)K / 4096.0 + 3.0;
            };
            // 3-bit layers (qgemm_block3.h): 128-row blocks 78 / 85 us alone, 85 / 90 busy; 256-row blocks (round 3) 108 / 118
            // alone, 124 / 128 busy (profiles/r03/block_lab_w3_256_row_blocks.jsonl); the per-wave kernel runs them at
            // 330-380 TFLOP/s
            const double t256 = (bits == 3) ? block_us(tiles256, bf ? 118.0 : 108.0, bf ? 128.0 : 124.0)
                                            : block_us(tiles256, bf ? 101.0 : 97.0, bf ? 124.0 : 120.0);
            const double t128 = (bits == 3) ? block_us(tiles128, bf ? 85.0 : 78.0, bf ? 90.0 : 85.0)
                                            : block_us(tiles128, bf ? 70.0 : 62.0, bf ? 78.0 : 74.0);
            // per-wave kernel: 520 (bf16 400) TFLOP/s at M = 256, + 55 per doubling of M, up to 730 (560)
            int dbl = 0;
            for (int m = M; m >= 512; m >>= 1) ++dbl;
            // (3 bits, round 4: 14336 x 3584 M = 256 runs at 305, modelled 370 kept it off the 128-row blocks: 86.4 against 70.3 us; round 5's
            // regret sweep, fp16: 3584 x 8192 M = 256 293, 14336 x 3584 M = 128 276, 8192^2 M = 96 253 - fewer rows, fewer MFMAs per lookup)
            const double wave_tf = (bits == 3) ? (bf ? 290.0 : 300.0) * (M >= 256 ? 1.0 : 0.6 + 0.4 * M / 256.0)
                                   : (M < 512 ? wave_tf_mid(M, bits, bf)
                                              : (bf ? std::min(560.0, 400.0 + 55.0 * dbl) : std::min(730.0, 520.0 + 55.0 * dbl)) * (bits == 2 ? 0.65 : 1.0));   // (2 bits: see below)
            const double wave_us = 2.0 * M * (double)N * K / (wave_tf * 1e6);
            if (t256 <= t128 && t256 < wave_us) blk_cfg = 4;
            else if (t128 < t256 && t128 < wave_us) blk_cfg = 5;
            alt_us = std::min(wave_us, std::min(t256, t128));
            if (bits == 3) {
                // 3-bit layers have no split-K block kernel of their own (qgemm_splitk.h: 2 / 4 bits): where whole blocks leave
                // CUs idle, 128- or 64-row blocks of qgemm_block3.h with a grid K split (fp32 slabs + the reduce pass) fill
                // them.  One round of workgroups; a block costs ~4 us + its K share of (128 rows: 78 / 85 us alone, 85 / 90 busy;
                // 64 rows: 66 / 72 - the lookups of a block's 256 columns dominate, the rows are nearly free), the slabs 0.25 us
                // per MB + the reduce launch.  Measured (bf16, profiles/r04/w3_mid_m_forced_plans.jsonl; before -> after):
                // M = 1024 x 4096^2 84.9 -> 56.4 us, M = 512 x 8192^2 160.5 -> 95.6, M = 512 x 4096^2 54.5 -> 45.4
                const double base = blk_cfg == 4 ? t256 : (blk_cfg == 5 ? t128 : wave_us);
                double best = 0.95 * base;                      // a K-split plan has to beat the unsplit best by 5 %; among themselves: the cheapest
                const int align_k = std::max(64, 8 << lg);
                for (int rows = 128; rows >= 64; rows >>= 1) {
                    const long tiles = (long)ceil_div(M, rows) * (units / blk_units);
                    const double alone = rows == 128 ? (bf ? 85.0 : 78.0) : (bf ? 72.0 : 66.0);
                    const double busy = rows == 128 ? (bf ? 90.0 : 85.0) : (bf ? 72.0 : 70.0);
                    for (int sk = (rows == 128 ? 2 : 1); sk <= 4; sk *= 2) {
                        const long wgs = tiles * sk;
                        // slices of kps k, the last one shorter where K is no multiple (K = 3584: 2048 + 1536, or 3 x 1024 + 512 - round 5:
                        // 14336 x 3584 M = 128 47.6 -> 33.9 us, M = 256 61.7 -> 49.3; until then only equal slices were priced)
                        const int kps = round_up(ceil_div(K, sk), align_k);
                        if (sk > 1 && (wgs > (long)num_sms || ceil_div(K, kps) != sk || kps < 1024 ||
                                       (size_t)sk * tiles * rows * 1024 > slab_room(workspace_bytes))) continue;
                        double us;
                        if (sk == 1) us = block_us(tiles, alone, busy);
                        else us = 4.0 + ((wgs * 4 >= (long)num_sms * 3 ? busy : alone) - 4.0) * (double)kps / 4096.0 +
                                  (rows == 128 ? 2.0 : 5.0) +              // (128-row blocks: combined in the launch, round 5; else the reduce launch)
                                  0.25 * (double)sk * M * N * 4.0 / 1e6;
                        if (us < best) { best = us; blk_cfg = rows == 128 ? 5 : 12; blk_sk = sk; }
                    }
                }
                alt_us = std::min(alt_us, blk_sk > 0 ? best : base);
            }
        }
        if (blk_cfg >= 0) family = kFamilyBlock;
    }
    // Split-K block kernel, automatic (the template's default Stages and SMs_Multiple): taken when its modelled time is
    // 8 % under the best of the per-wave kernel (520 TFLOP/s at M = 256, - 55 at 128, + 55 per doubling) and the block kernels.
    // Measured (tools/splitk_lab.py, us, automatic plan of round 3 -> this kernel): M = 256 x 4096 x 11008 44.5 -> 38.1,
    // 4096 x 14336 47.2 -> 40.2, 8192^2 49.0 -> 44.2; M = 1024 x 4096^2 49.6 -> 41.8 (torch.mm 46.9), M = 512 29.0 -> 27.4;
    // not taken: M = 256 x 4096^2 (24.9 against 20.1: the seam of four slices), M = 128 x 4096^2, K = 14336.
    if (sk_regime && t.stages == 2 && t.sms_multiple == 1 && (blk_cfg < 0 || blk_cfg == 4 || blk_cfg == 5)) {
        if (alt_us < 0.0) {
            const bool bf = dtype == FLUTE_BF16;
            int dbl = M < 256 ? -1 : 0;
            for (int m = M; m >= 512; m >>= 1) ++dbl;
            double wave_tf = bf ? std::min(560.0, 400.0 + 55.0 * dbl) : std::min(730.0, 520.0 + 55.0 * dbl);
            // (2-bit layers: twice the lookups per byte - the per-wave kernel ran M = 384 on 4096^2 at 316 TFLOP/s where the 4-bit layer runs 503:
            // profiles/r06_planner_regret_between_final.json)
            if (bits == 2) wave_tf *= 0.65;
            if (M >= 128 && M < 512) wave_tf = wave_tf_mid(M, bits, bf);
            if (M < 128) {
                // per-wave kernel below M = 128 (measured fp16, tools/time_cases.py): 220 .. 280 TFLOP/s at M = 48, 270 .. 350 at
                // M = 64 .. 96 on layers up to 14336 columns; 490 .. 570 on 28672 columns (two slabs per wave, every CU busy)
                wave_tf = std::min(300.0, 5.5 * M) * (bf ? 0.8 : 1.0);
                if (N >= 16384) wave_tf *= 1.8;
                // (the 2-bit factor holds here too - it was applied above the branch only: 2-bit M = 48 on 4096^2 kept the per-wave kernel x 4 K slices,
                // 15.0 us, where 64 x 64 tiles x 4 slices run 11.1: profiles/r06_planner_regret_final_tree_between.json)
                if (bits == 2) wave_tf *= 0.65;
            }
            alt_us = 2.0 * M * (double)N * K / (wave_tf * 1e6);
        }
        flute_plan q;
        double sk_us = 0.0;
        // (one round of workgroups only: multi-round launches are left to the tuner's Stages-5 ids until measured)
        // (round 6: 64 x 64 tiles - four K parts per workgroup - that fill at least half the chip are priced against the per-wave kernel
        // WITH its fixed part, which the TFLOP/s model above lacks: measured - modelled 3.2 .. 5.7 us on 4096-wide layers, 10 on
        // 2048 x 8192; profiles/r06/call17_automatic_plan_vs_forced.log: M = 128 on 4096^2 13.1 against 14.9 us, M = 192 15.1 / 18.8, M = 512 on
        // 2048 x 4096 15.8 / 19.9, M = 256 on 2048 x 8192 18.0 / 26.6, M = 48 on 3584 x 14336 16.7 / 21.6, M = 33 on 8192^2 17.7 / 26.1)
        if (plan_splitk(bits, lg, M, N, K, num_sms, ov, workspace_bytes, &q, 0, &sk_us) == FLUTE_OK &&
            sk_us < ((q.kw == 4 && (long)q.grid * 2 >= (long)num_sms) ? alt_us + 4.0 : 0.92 * alt_us) &&
            (long)q.grid <= (bits == 2 && M <= 64 ? 2L : 1L) * (long)num_sms && q.lds_bytes <= (size_t)kMaxLds) {
            // (2-bit layers up to M = 64: also two rounds - 28672 x 8192 M = 33 / 48 / 64: 448 workgroups of 64 x 128 tiles x 2 slices 46.6 / 47.8 / 49.3 us
            // against the per-wave kernel's 61.3 / 61.5 / 62.3, profiles/r06/planner_regret_b2_m33_96_after_2bit_rate_fix.json)
            *p = q;
            return FLUTE_OK;
        }
    }
    p->family = family;

    int rc = FLUTE_OK;
    if (family == 0) {
        // Three decode kernels: the one-shot kernel (qgemm_oneshot.h: non-persistent workgroups, every request up
        // front) and the persistent ring kernel (qgemm_stream.h).  Forced by override (one_shot 1 / 0; an explicit
        // ring depth or grid K split means the ring kernel) or by the template (4-bit QuantMapMode digit 1, 2:
        // one-shot with 4 / 8 pieces per wave, 3: ring; 2- / 3-bit SMs_Multiple 4: one-shot, 2: ring); automatic:
        // one-shot for layers up to 64 M weights that give at least half the CUs a workgroup; one or two rows on larger layers:
        // the persistent one-shot kernel (qgemm_persist.h; override one_shot = 3, as flute_plan reports it, or 2).
        int want = ov.one_shot == 3 ? 2 : ov.one_shot;       // 3 = flute_plan's code for the persistent kernel (2 kept from ABI v4)
        if (want < 0 && (ov.depth > 0 || ov.splitk > 1)) want = 0;
        if (want < 0 && bits == 4) { const int q = template_id % 4; want = (q == 3) ? 0 : ((q == 1 || q == 2) ? 1 : -1); }
        if (want < 0 && bits != 4) want = (t.sms_multiple == 2) ? 0 : (t.sms_multiple == 4 ? 1 : -1);
        bool taken = false;
        // lean one-row kernel (qgemm_fast.h): by override (one_shot = 4), or automatically for the ids whose last digit leaves the choice
        // to the planner and whose Stages digit asks for the planner's first or second shape (4-bit QuantMapMode digit 0, Stages 2 / 3,
        // SMs_Multiple 1: ids 0 / 4 - TileP 64 - and 16 / 20 - TileP 32), on K = 2048 / 4096 layers up to 48 M weights that give at
        // least half the CUs a workgroup - measured against the persistent and the round-4 one-shot kernel (us, persistent / lean /
        // one-shot): 4096^2 4.42 / 4.05 / 4.14, 5120 4.96 / 4.90 / 5.88, 8192 5.51 / 5.46 / 5.91, 11008 7.54 / 6.86 / 8.04; not taken:
        // 14336 7.85 / 8.27 / 8.43, 28672 13.6 / 13.9 / 14.6, more than three workgroups per CU (16384 x 2048: 5.97 against 5.49 on the
        // one-shot kernel; 8192 x 2048 3.77 / 4.05 is taken), K = 8192 (8192^2 8.42 against 10.42, 4096 x 8192 5.71 / 5.97: by
        // override only).  Two to four rows (dot products per row on the same lookups): while ONE round of workgroups covers the layer
        // (4096^2: M = 2 5.04 -> 4.29 us, M = 3, 4 6.25 -> 5.03; 4096 x 2048: 3.87 -> 3.25, 4.65 -> 3.71; beyond, the persistent kernel
        // (M = 2) and the skinny MFMA kernel (M = 3, 4) win: 8192 x 4096 5.79 against 7.12, 8.69 against 8.98).  Never for a call that
        // fuses the Hadamard rotation.  K = 8192 (shape (8, 2, 8)), round 5's last call series: one row a tie with the one-shot kernel
        // (3584 x 8192: 5.33 / 5.40 us, 4096 x 8192: 5.57 / 5.51), TWO rows on layers that give >= 80 % of the CUs a workgroup 6.11 against
        // 6.78 and 6.33 against 6.82 - taken; narrower layers (2048, 1024 columns: 128 / 64 workgroups) lose 13 - 17 % and are not.
        // K = 2048, two rows: up to two rounds (6144 x 2048 4.34 -> 3.79 us, 8192 x 2048 4.38 -> 4.01; four rows lose there: 5.14 / 5.76)
        if ((want == 4 || (want < 0 && bits == 4 && (template_id % 4) == 0 && t.stages <= 3 && t.sms_multiple == 1 && ov.waves < 0)) &&
            !ov.had8 && ov.kw < 0) {
            flute_plan q;
            memset(&q, 0, sizeof(q));
            if (plan_fast(bits, lg, M, N, K, num_sms, std::max(0, t.stages - 2), want == 4 ? ov.waves : -1, &q, oa) == FLUTE_OK &&
                (want == 4 || ((K != 8192 || (M == 2 && (long)q.grid * 5 >= (long)num_sms * 4)) &&
                               (size_t)N * K <= ((size_t)48 << 20) && (long)q.grid * 2 >= (long)num_sms &&
                               (long)q.grid <= (M == 1 ? 3L : (M == 2 && K == 2048 ? 2L : 1L)) * num_sms))) {
                *p = q;
                taken = true;
            }
        }
        if (want == 4 && !taken) want = -1;
        // persistent one-shot kernel: by override, or automatically (one or two rows; four rows measured 1.7x the one-row
        // time - no faster than the MFMA kernel, profiles/r03/decode_lab_persist_rows.jsonl) on layers of >= 40 M weights that give
        // every CU six whole unit rows (below that the in-workgroup K split of the other two kernels wins:
        // profiles/r03/persist_lab.txt)
        const bool persist_auto = !taken && want < 0 && persist_auto_ok(M);
        if (!taken && (want == 2 || persist_auto)) {
            flute_plan q;
            memset(&q, 0, sizeof(q));
            if (plan_persist(bits, lg, M, N, K, num_sms, t, ovd, &q, oa) == FLUTE_OK) { *p = q; taken = true; }
            else if (want == 2) want = 0;
        }
        if (!taken && want != 0) {
            flute_plan q;
            memset(&q, 0, sizeof(q));
            if (plan_oneshot(bits, lg, M, N, K, num_sms, t, template_id, ovd, &q, oa) == FLUTE_OK &&
                (want == 1 || ((size_t)N * K <= ((size_t)64 << 20) && (long)q.grid * 2 >= (long)num_sms))) {
                *p = q;
                taken = true;
            }
        }
        if (!taken) rc = plan_stream(dtype, bits, lg, M, N, K, num_sms, t, ovd, slab_room(workspace_bytes), p, sa);
    } else if (family == kFamilyBlock) {
        const int bm = block_rows(blk_cfg), tm = bm / 32;
        const int tiles_m = ceil_div(M, bm), tiles_n = units / (256 / J);
        const int align_k = std::max(64, 8 << lg);
        int splitk = (ov.splitk > 0) ? ov.splitk : (blk_sk > 0 ? blk_sk : 1);
        if (blk_cfg >= 8 && ov.splitk <= 0 && blk_sk == 0)   // skinny blocks: the K split fills the chip
            while ((long)tiles_m * tiles_n * splitk * 2 <= (long)num_sms && K / (splitk * 2) >= std::max(256, align_k)) splitk *= 2;
        int kps = round_up(ceil_div(K, splitk), align_k);
        splitk = ceil_div(K, kps);
        while (splitk > 1 && (size_t)splitk * M * N * 4 > slab_room(workspace_bytes)) {
            splitk >>= 1;
            kps = round_up(ceil_div(K, splitk), align_k);
            splitk = ceil_div(K, kps);
        }
        if (splitk == 1) kps = K;
        p->m_block = blk_cfg; p->m_tiles = tm; p->slabs_per_wave = 1; p->waves = 8; p->kw = 1;
        p->splitk = splitk; p->k_per_split = kps;
        p->grid = (unsigned)((long)tiles_m * tiles_n * splitk);
        p->block = 512;
        // 3-bit 128-row blocks x 2 / 4 K slices (round 5): the slices of a block meet inside the launch (xwg.h, E form; slabs in
        // fragment order, whole blocks: tiles x 128 x 256 x 4 B per slice) - no reduce launch.  Measured against the reduce launch
        // (bf16, us, profiles/r05/call24_w3_inlaunch_and_line_planes.log): M = 1024 x 4096^2 56.3 -> 53.0 (fp16 53.4 -> 50.1),
        // M = 256 x 8192^2 56.0 -> 54.4, M = 512 x 8192^2 95.3 -> 92.7, M = 96 x 28672 x 8192 92.1 -> 89.0, M = 512 x 4096^2 a tie; NOT for
        // the skinny blocks (64 rows x 4 slices 46.7 -> 47.8; 8 slices, L form - the last arriver reads seven partials - 30.5 -> 32.6)
        if (bits == 3 && (splitk == 2 || splitk == 4) && bm == 128 && (long)tiles_m * tiles_n <= (long)kXwgMaxTiles && !block3_two_launch()) {
            const size_t slabs = (size_t)splitk * tiles_m * tiles_n * bm * 1024;
            if (slabs <= slab_room(workspace_bytes) && slabs < ((size_t)1 << 31)) { p->splitk_mode = 1; blk_slabs = slabs; }
        }
        // pair table + three activation stages + per wave: two scale blocks and a sink
        // (3-bit 256-row blocks: + 18 KB, the second / third plane pieces of waves 6 and 7)
        p->lds_bytes = (size_t)((128 << (2 * bits)) + 3 * (bm / 16) * 2 * 1024 + 8 * 3 * 1024 + ((bits == 3 && bm == 256) ? 18 * 1024 : 0));
        p->lut_copies = 32;
    } else {
        // M > decode range: MFMA kernel (qgemm_tile.h).  MT 16-row tiles per wave (1 for M <= 16),
        // R lanes share a unit: pick the smallest R whose slab x row-tile count fills the chip; the
        // rest of the parallelism is the in-workgroup K split, a grid-level split only for very
        // narrow layers.
        int mt = (M <= 16) ? 1 : (M <= 32 ? 2 : 4);
        const int mt_cap = (bits == 3) ? 2 : 4;
        if (mt > mt_cap) mt = mt_cap;
        if (t.tile_m / 16 < mt && M > 16) mt = t.tile_m / 16 >= 2 ? t.tile_m / 16 : mt;
        // SMs_Multiple = "more, smaller workgroups": halves / quarters the row tiles per wave (and,
        // below, raises the slab count the choice of R aims for)
        for (int m2 = t.sms_multiple; m2 > 1 && mt > 1; m2 >>= 1) mt >>= 1;
        if (ov.m_tiles == 1 || ov.m_tiles == 2 || ov.m_tiles == 4) mt = ov.m_tiles;
        if (mt > mt_cap) mt = mt_cap;
        // instantiated (R, MT): (J/R)*MT <= 16 accumulator tiles, R in {1,2,4}, MT > 1 needs R <= 2
        auto combo_ok = [&](int r, int m) {
            if (bits == 3) return r == 1 && m == 1;
            return (J / r) * m <= 16 && r <= 4 && (m == 1 || r <= 2);
        };
        while (mt > 1 && !combo_ok(1, mt) && !combo_ok(2, mt)) mt >>= 1;
        const int mtiles = ceil_div(M, mt * 16);
        int R = 1;
        while (!combo_ok(R, mt)) R *= 2;
        // Every workgroup pulls its rows of X through one CU, so lane sharing (R-fold more, narrower slabs) multiplies the
        // activation traffic: it pays only while the workgroups leave more than ~45 % of the CUs idle (round 3,
        // profiles/r03/tile_lab_sw2_m16.jsonl, tile_lab_sw2_m32_m128.jsonl: 10240 x 8192 M = 16, 160 slabs: R = 1 22.2 us,
        // R = 2 26.8; M = 64: 36.7 / 49.9; 4096 x 11008 M = 64: 21.4 / 29.1; but 8192^2 M = 32, 128 slabs: R = 2 16.2, R = 1 19.3)
        auto fills = [&](long wgs) { return wgs * 20 >= 11L * num_sms * t.sms_multiple; };
        // ... and not at all where the grid K split can fill the chip instead (round 5's regret sweep, the in-launch seam of xwg.h being
        // cheap now): deep layers - K >= 10240: two slices, K >= 12288: four - stop sharing lanes as soon as that split fills.
        // 4096 x 11008 (N x K) M = 4: four lanes per unit 16.6 us, two lanes x 2 slices 14.8 (2 bits: 14.2 -> 12.5), M = 16 16.9 -> 15.4;
        // 3584 x 14336 M = 48: two lanes x 2 slices 26.4, no sharing x 4 slices 21.6 (profiles/r05_planner_regret_*.json)
        const long deep_split = (bits != 3 && (bits != 4 || (template_id % 4) == 0) && ov.splitk < 0) ? (K >= 12288 ? 4 : (K >= 10240 ? 2 : 1)) : 1;
        while (combo_ok(R * 2, mt) && !fills((long)units * R / 16 * mtiles) &&
               !(deep_split > 1 && fills((long)units * R / 16 * mtiles * deep_split))) R *= 2;
        // QuantMapMode digit 1 (4-bit ids): no lane sharing above M = 16 - the chip is filled by the grid K split
        // instead (8192^2 M = 64: R = 1, MT = 4, split 2 26.2 us against R = 2, MT = 2 30.3; 4096^2 prefers R = 2:
        // the tuner decides)
        if (bits == 4 && (template_id % 4) == 1 && combo_ok(1, mt)) R = 1;       // (M <= 16: with two slabs per wave, below)
        // QuantMapMode digit 3 above M = 16 (round 5): no lane sharing AND two slabs per wave, the chip filled by the grid K split - the
        // plan round 4's regret sweep wanted on 8192 x 28672 and could not reach through an id (M = 64: 72.6 -> 55.3 us, M = 48 65.0 ->
        // 52.8, M = 32 52.4 -> 45.2).  The automatic digit takes it by itself on layers that deep (K >= 16384: a slice keeps >= 4096 k) whose halved slab count x four slices fills the chip
        const bool deep_sw2 = bits == 4 && M > 16 && combo_ok(1, mt) && (dtype == 0 || mt <= 2) && (units / 16) % 2 == 0 && ov.m_block <= 0 &&
                              ((template_id % 4) == 3 || ((template_id % 4) == 0 && K >= 16384 && !fills((long)(units / 16) * mtiles) && fills((long)(units / 32) * mtiles * 4)));
        if (deep_sw2) R = 1;
        if (ov.m_block > 0 && combo_ok(ov.m_block, mt)) R = ov.m_block;
        // SW = 2 slabs per wave (4-bit, no lane sharing, fp16 up to MT = 4 / bf16 up to MT = 2: the bf16 path
        // keeps a second accumulator set): every activation fragment then serves 8 column tiles and the
        // texture-path traffic per MFMA drops by 40 %.  Worth it once halving the slab count still leaves a
        // workgroup for every CU; QuantMapMode (the last template digit) lets the tuner force either.
        const bool sw_ok = bits == 4 && R == 1 && (dtype == 0 || mt <= 2) && (units / 16) % 2 == 0;
        int sw = 1;
        // ... as soon as the halved slab count still fills 55 % of the CUs (28672 x 8192 M = 16: 448 workgroups 43.4 us, 224
        // workgroups 37.1; M = 64: 73.4 -> 54.2; 4096 x 14336 M = 128: 38.2 -> 28.2) - or, at M <= 16, on the tuner's request (digit 1)
        if (sw_ok && (fills((long)(units / 32) * mtiles) || (mt == 1 && M <= 16 && (template_id % 4) == 1))) sw = 2;
        if (sw_ok && bits == 4 && ((template_id % 4) == 3 || deep_sw2)) sw = 2;
        if (bits == 4 && (template_id % 4) == 2) sw = 1;
        if (sw_ok && ov.slabs == 2) sw = 2;
        if (ov.slabs == 1) sw = 1;
        const int slabs = units * R / 16 / sw;                    // wave-sized column groups
        int nw = (t.threads >= 1024) ? 8 : 4;                     // Threads 1024 / 512 templates
        if (ov.waves > 0 && ov.waves <= 8) nw = floor_pow2(ov.waves);
        while (nw > 1 && tile_geom(bits, R, mt, sw, nw, kMaxLds).depth < 2) nw >>= 1;   // ring of >= 2 slots per wave
        int kw = nw;
        while (kw > 1 && K / kw < 256) kw >>= 1;
        // enough workgroups already: keep more of K per wave (fewer partial tiles to reduce)
        while (kw > 1 && (long)slabs * mtiles / (nw / kw) >= 2L * num_sms * t.sms_multiple && K / kw < 1024) kw >>= 1;
        if (t.stages == 3 && kw > 1) kw >>= 1;                    // the tuner's handle on the K split
        if (t.stages == 4 && kw < nw) kw <<= 1;
        if (t.stages == 5 && kw > 2) kw >>= 2;
        if (ov.kw > 0 && ov.kw <= nw) kw = floor_pow2(ov.kw);
        while (nw % kw) kw >>= 1;
        while (slabs % (nw / kw)) kw <<= 1;
        const long wgs = (long)slabs / (nw / kw) * mtiles;
        int splitk = 1;
        // 3-bit layers have no lane-sharing variants (a wave = 16 units x 16 fields = 256 columns): the grid-level K
        // split is their only way to fill the chip, down to 64 k per wave (4096^2 M = 16: 20.5 -> 14.5 us,
        // 4096x14336: 30.8 -> 20.2 us)
        const int k_min = (bits == 3) ? 64 : 256;
        while (wgs * splitk * 2 <= (long)num_sms && K / (splitk * 2 * kw) >= k_min) splitk *= 2;
        if (ov.splitk > 0) splitk = ov.splitk;
        int kps = round_up(ceil_div(K, splitk), 32 * kw);
        splitk = ceil_div(K, kps);
        while (splitk > 1 && (size_t)splitk * M * N * 4 > slab_room(workspace_bytes)) {
            splitk >>= 1;
            kps = round_up(ceil_div(K, splitk), 32 * kw);
            splitk = ceil_div(K, kps);
        }
        if (splitk == 1) kps = K;
        p->m_block = R; p->m_tiles = mt; p->slabs_per_wave = sw; p->waves = nw; p->kw = kw; p->splitk = splitk;
        p->k_per_split = kps;
        // round 4: the K slices of a (slab group, row tile) meet inside the launch (xwg.h, L form) while the slabs are small -
        // the reduce launch it replaces costs >= 2 us; beyond 4 MB of slabs the all-CU reduce pass reads them faster than the
        // last arrivers would
#ifndef FLUTE_TILE_INLAUNCH_MAX
#define FLUTE_TILE_INLAUNCH_MAX (4 << 20)     // bytes of slabs; development builds set 0 to time the two-launch form
#endif
        if (splitk > 1 && wgs <= kXwgMaxTiles && (size_t)splitk * M * N * 4 <= (size_t)FLUTE_TILE_INLAUNCH_MAX) p->splitk_mode = 1;
        p->grid = (unsigned)(wgs * splitk);
        p->block = (unsigned)(nw * 64);
        p->lds_bytes = (size_t)tile_geom(bits, R, mt, sw, nw, kMaxLds).total;
        p->lut_copies = 32;
    }
    if (rc) return rc;
    p->workspace_needed = p->splitk > 1 ? (blk_slabs ? blk_slabs : (size_t)p->splitk * M * N * 4) + kXwgFlagBytes : 0;
    if (p->lds_bytes > (size_t)kMaxLds) return FLUTE_ERR_SHAPE;
    return FLUTE_OK;
}

// make_plan is a pure function of its arguments and runs on every call of the operator (ranking the decode
// shapes costs a few microseconds of host time - as much as the kernel it plans): memoise the last results
// per host thread.  thread_local, so there is still no shared mutable state.
struct PlanKey {
    int dtype, bits, group, M, N, K, template_id, num_sms;
    size_t workspace_bytes;
    Ovr ov;
    bool operator==(const PlanKey& o) const { return memcmp(this, &o, sizeof(PlanKey)) == 0; }
};
struct PlanEntry { PlanKey key; int rc; flute_plan plan; flute_template_info tinfo; StreamArgs sa; OneArgs oa; bool valid; };

int make_plan(int dtype, int bits, int group, int M, int N, int K, int template_id, int num_sms,
              size_t workspace_bytes, const Ovr& ov, flute_plan* p, flute_template_info* tinfo,
              StreamArgs* sa, OneArgs* oa) {
    constexpr int kEntries = 32;
    thread_local PlanEntry cache[kEntries] = {};
    thread_local int next = 0;
    PlanKey key;
    memset(&key, 0, sizeof(key));
    key.dtype = dtype; key.bits = bits; key.group = group; key.M = M; key.N = N; key.K = K;
    key.template_id = template_id; key.num_sms = num_sms; key.workspace_bytes = workspace_bytes; key.ov = ov;
    for (int i = 0; i < kEntries; ++i) {
        const PlanEntry& e = cache[i];
        if (e.valid && e.key == key) {
            if (e.rc == FLUTE_OK) { *p = e.plan; if (tinfo) *tinfo = e.tinfo; if (sa) *sa = e.sa; if (oa) *oa = e.oa; }
            return e.rc;
        }
    }
    PlanEntry& e = cache[next];
    next = (next + 1) % kEntries;
    e.valid = false;
    e.key = key;
    memset(&e.plan, 0, sizeof(e.plan));
    memset(&e.sa, 0, sizeof(e.sa));
    memset(&e.oa, 0, sizeof(e.oa));
    e.rc = make_plan_uncached(dtype, bits, group, M, N, K, template_id, num_sms, workspace_bytes, ov, &e.plan, &e.tinfo,
                              &e.sa, &e.oa);
    e.valid = true;
    if (e.rc == FLUTE_OK) { *p = e.plan; if (tinfo) *tinfo = e.tinfo; if (sa) *sa = e.sa; if (oa) *oa = e.oa; }
    return e.rc;
}

StreamKernel pick_stream_kernel(int bits, int dtype, int tile_p, int mb, int depth, int one_shot) {
    if (bits == 4) return stream_kernel_b4(dtype, tile_p, mb, depth, one_shot);
    if (bits == 3) return stream_kernel_b3(dtype, tile_p, mb, depth, one_shot);
    return stream_kernel_b2(dtype, tile_p, mb, depth, one_shot);
}

QGemmKernel pick_kernel(int family, int bits, int dtype, int tile_p, int mblk, int mtiles, int sw) {
    if (bits == 4) return tile_kernel_b4(dtype, tile_p, mblk, mtiles, sw);
    if (bits == 3) return tile_kernel_b3(dtype, tile_p, mblk, mtiles);
    return tile_kernel_b2(dtype, tile_p, mblk, mtiles);
}

// (device, kernel) pairs already granted > 64 KB of dynamic LDS: the attribute is per device, and one
// process may drive several GPUs (accelerate-style sharded inference)
struct BigLds { int dev; const void* fn; };
BigLds g_big_lds[512];
int g_big_lds_n = 0;
std::mutex g_big_lds_mu;                       // flute_qgemm may be called from several host threads

int ensure_lds(const void* fn, size_t bytes) {
    if (bytes <= 65536) return 0;
    std::lock_guard<std::mutex> lock(g_big_lds_mu);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return FLUTE_ERR_LAUNCH; }
    for (int i = 0; i < g_big_lds_n; ++i)
        if (g_big_lds[i].fn == fn && g_big_lds[i].dev == dev) return 0;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds) != hipSuccess) {
        (void)hipGetLastError();
        return FLUTE_ERR_LAUNCH;
    }
    if (g_big_lds_n < 512) g_big_lds[g_big_lds_n++] = BigLds{dev, fn};
    return 0;
}

// The decode kernels can rotate the activations while staging them - every workgroup rotates ALL rows for itself, so the
// fused form costs ~0.33 us per 1024 elements of M x K (measured, profiles/r04/hadamard_fused_vs_separate.json: M = 1
// K = 4096 +1.1 us, M = 1 K = 14336 +4.9, M = 4 K = 3584 +4.5) against ~3.1 us for the separate flute_hadamard launch:
// fused only up to 8192 elements (or when the caller forces the decode family: tests, A/B runs).
bool hadamard_worth_fusing(int M, int K, bool forced) { return forced || (size_t)M * K <= 8192; }
bool hadamard_fusable(const flute_plan& p, int hadamard_size, int M, int K, bool forced) {
    return p.family == 0 && p.one_shot != 4 && hadamard_size >= 2 && hadamard_size <= 512 &&
           (hadamard_size & (hadamard_size - 1)) == 0 && K % hadamard_size == 0 && hadamard_worth_fusing(M, K, forced);
}

}  // namespace

extern "C" {

int flute_abi_version(void) { return FLUTE_AMD_ABI_VERSION; }

const char* flute_strerror(int status) {
    switch (status) {
        case FLUTE_OK: return "ok";
        case FLUTE_ERR_NUM_BITS: return "Unsupported num_bits value";
        case FLUTE_ERR_GROUP_SIZE: return "Unsupported group_size value";
        case FLUTE_ERR_TEMPLATE_ID: return "Unsupported template_id value";
        case FLUTE_ERR_SHAPE: return "Unsupported shape: need N % (16/num_bits*TileP) == 0 (N % 512 for 3 bits), K % 64 == 0, K % group_size == 0";
        case FLUTE_ERR_WORKSPACE: return "workspace too small";
        case FLUTE_ERR_LAUNCH: return "CUDA error: invalid argument (HIP kernel launch failed)";   // flute/tune.py:160 string-matches this prefix
        case FLUTE_ERR_DTYPE: return "Unsupported dtype (fp16 / bf16 only)";
        case FLUTE_ERR_HADAMARD_SIZE: return "Only power of two Hadamard sizes up to 2^15 are supported";
        case FLUTE_ERR_NULL: return "null pointer argument";
        default: return "unknown flute_amd status";
    }
}

int flute_num_templates(int num_bits) {
    if (num_bits == 4) return 144;
    if (num_bits == 2 || num_bits == 3) return 36;
    return 0;
}

int flute_get_template_info(int num_bits, int template_id, flute_template_info* out) {
    if (!out) return FLUTE_ERR_NULL;
    if (num_bits != 2 && num_bits != 3 && num_bits != 4) return FLUTE_ERR_NUM_BITS;
    return decode_template(num_bits, template_id, out) ? FLUTE_OK : FLUTE_ERR_TEMPLATE_ID;
}

int flute_qgemm_plan_ex(int dtype, int num_bits, int group_size, int M, int N, int K, int template_id,
                        int num_sms, size_t workspace_bytes, const flute_overrides* ovr, flute_plan* out) {
    if (!out) return FLUTE_ERR_NULL;
    return make_plan(dtype, num_bits, group_size, M, N, K, template_id, num_sms, workspace_bytes,
                     ovr_of(ovr), out, nullptr, nullptr, nullptr);
}

int flute_qgemm_plan(int dtype, int num_bits, int group_size, int M, int N, int K,
                     int template_id, int num_sms, size_t workspace_bytes, flute_plan* out) {
    return flute_qgemm_plan_ex(dtype, num_bits, group_size, M, N, K, template_id, num_sms, workspace_bytes,
                               nullptr, out);
}

int flute_qgemm(int dtype, int num_bits, int group_size, int M, int N, int K, int P,
                const void* A, const void* Q, void* D, const void* S, const void* QM,
                const void* QM2, void* workspace, size_t workspace_bytes, int template_id,
                int num_sms, void* stream) {
    return flute_qgemm_ex(dtype, num_bits, group_size, 0, M, N, K, P, A, Q, D, S, QM, QM2, nullptr,
                          workspace, workspace_bytes, template_id, num_sms, nullptr, stream);
}

// Calls that will fuse the rotation prefer 8-wave workgroups: the rotation is done by the workgroup's waves, 512 k each - 8 waves
// rotate a 4096-k row in one pass (4096x3584 M = 1: 5.5 us with 8 waves, 6.5 with the 4-wave shape the plain product prefers).
// (Applied inside the decode branch of the planner only: make_plan_uncached.  Round 3 also forced the four-row decode
// kernel at M = 3, 4 to keep the rotation fused: measured again in round 4, the fused form loses there - above.)
static Ovr hadamard_ovr(Ovr o, int hadamard_size, int bits, int M, int N, int K) {
    (void)bits; (void)N;
    if (hadamard_size > 1 && hadamard_size <= 512 && M <= 4 && hadamard_worth_fusing(M, K, o.family == 0)) o.had8 = 1;
    return o;
}

int flute_qgemm_hadamard_fused(int dtype, int num_bits, int group_size, int hadamard_size, int M,
                               int N, int K, int template_id, int num_sms, size_t workspace_bytes) {
    flute_plan p;
    if (make_plan(dtype, num_bits, group_size, M, N, K, template_id, num_sms, workspace_bytes,
                  hadamard_ovr(ovr_of(nullptr), hadamard_size, num_bits, M, N, K), &p, nullptr, nullptr, nullptr))
        return 0;
    return hadamard_fusable(p, hadamard_size, M, K, false) ? 1 : 0;
}

int flute_qgemm_hadamard(int dtype, int num_bits, int group_size, int hadamard_size, int M, int N,
                         int K, int P, const void* A, const void* Q, void* D, const void* S,
                         const void* QM, const void* QM2, void* x_scratch, void* workspace,
                         size_t workspace_bytes, int template_id, int num_sms, void* stream) {
    return flute_qgemm_ex(dtype, num_bits, group_size, hadamard_size, M, N, K, P, A, Q, D, S, QM, QM2,
                          x_scratch, workspace, workspace_bytes, template_id, num_sms, nullptr, stream);
}

int flute_qgemm_ex(int dtype, int num_bits, int group_size, int hadamard_size, int M, int N, int K, int P,
                   const void* A, const void* Q, void* D, const void* S, const void* QM,
                   const void* QM2, void* x_scratch, void* workspace, size_t workspace_bytes,
                   int template_id, int num_sms, const flute_overrides* ovr, void* stream) {
    (void)QM;   // single-code table: unused, the kernel reads only the pair table (as the reference)
    if (M == 0) return FLUTE_OK;
    if (hadamard_size > 1 && (hadamard_size & (hadamard_size - 1))) return FLUTE_ERR_HADAMARD_SIZE;
    // everything is validated before anything is enqueued
    flute_plan p;
    flute_template_info t;
    StreamArgs sa;
    OneArgs oa;
    if (!workspace) workspace_bytes = 0;
    const int rc = make_plan(dtype, num_bits, group_size, M, N, K, template_id, num_sms, workspace_bytes,
                             hadamard_ovr(ovr_of(ovr), hadamard_size, num_bits, M, N, K), &p, &t, &sa, &oa);
    if (rc) return rc;
    if (P != num_bits * (N / 16)) return FLUTE_ERR_SHAPE;
    if (!A || !Q || !D || !S || !QM2) return FLUTE_ERR_NULL;
    if (p.splitk > 1 && (!workspace || p.workspace_needed > workspace_bytes)) return FLUTE_ERR_WORKSPACE;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);

    int had_log = 0;
    if (hadamard_size > 1) {
        if (hadamard_fusable(p, hadamard_size, M, K, ovr && ovr->family == 0)) {
            had_log = ilog2(hadamard_size);              // rotated inside the decode kernel's staging
        } else {
            // two launches (qgemm.cpp:201-244): rotate into the caller's scratch, then the plain product
            if (!x_scratch) return FLUTE_ERR_NULL;
            const int hrc = hadamard_dispatch(dtype, A, x_scratch, (size_t)M * K, (uint32_t)hadamard_size, st);
            if (hrc) return hrc;
            A = x_scratch;
        }
    }
    const float had_scale = 1.0f / sqrtf((float)(1 << had_log));     // as flute_hadamard: bit-identical results

    if (p.family == kFamilySkinny) {
        SkinnyKernel fn = skinny_kernel_b4(dtype, t.tile_p, oa.depth);
        if (!fn) return FLUTE_ERR_TEMPLATE_ID;
        if (ensure_lds(reinterpret_cast<const void*>(fn), p.lds_bytes)) return FLUTE_ERR_LAUNCH;
        const uint32_t* q32 = reinterpret_cast<const uint32_t*>(Q);
        const uint32_t* qm2 = reinterpret_cast<const uint32_t*>(QM2);
        uint32_t geo = SkinnyGeo::pack(oa.lg, oa.lkw, oa.ipw, p.splitk);
        uint64_t* stamps = nullptr;
        float* partial = workspace ? reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + kXwgFlagBytes) : nullptr;
        uint32_t* state = reinterpret_cast<uint32_t*>(workspace);
#ifdef FLUTE_STAMPS   // behind the state words and the slabs
        if (workspace && workspace_bytes >= p.workspace_needed + kXwgFlagBytes + (size_t)p.grid * p.waves * 128)
            stamps = reinterpret_cast<uint64_t*>(reinterpret_cast<char*>(workspace) + (p.workspace_needed ? p.workspace_needed : kXwgFlagBytes));
#endif
        void* kargs[] = {&q32, &S, &A, &qm2, &K, &N, &geo, &M, &D, &stamps, &partial, &state};
        if (hipLaunchKernel(reinterpret_cast<const void*>(fn), dim3(p.grid), dim3(p.block), kargs, p.lds_bytes, st) !=
            hipSuccess) {
            (void)hipGetLastError();
            return FLUTE_ERR_LAUNCH;
        }
        return FLUTE_OK;
    }

    if (p.family == 0 && p.one_shot == 3) {
        const int had = had_log > 0 ? 1 : 0;
        PersistKernel fn = num_bits == 4 ? persist_kernel_b4(dtype, t.tile_p, p.m_block, oa.depth, oa.nsets, had)
                           : (num_bits == 2 ? persist_kernel_b2(dtype, t.tile_p, p.m_block, oa.depth, oa.nsets, had)
                                            : persist_kernel_b3(dtype, t.tile_p, p.m_block, oa.depth, oa.nsets, had));
        if (!fn) return FLUTE_ERR_SHAPE;
        if (ensure_lds(reinterpret_cast<const void*>(fn), p.lds_bytes)) return FLUTE_ERR_LAUNCH;
        const uint32_t* q32 = reinterpret_cast<const uint32_t*>(Q);
        const uint32_t* qm2 = reinterpret_cast<const uint32_t*>(QM2);
        uint32_t geo = PersistGeo::pack(oa.lg, p.waves, oa.nch, oa.ipw, had_log, oneshot_x_in_holes(num_bits, p.m_block, K) ? 1 : 0, M);
        float hs = had_scale;
        int nvis = oa.nvis, nwg = oa.nwg;
        void* kargs[] = {&q32, &S, &A, &qm2, &K, &N, &geo, &nvis, &D, &hs, &nwg};
        if (hipLaunchKernel(reinterpret_cast<const void*>(fn), dim3(p.grid), dim3(p.block), kargs, p.lds_bytes, st) !=
            hipSuccess) {
            (void)hipGetLastError();
            return FLUTE_ERR_LAUNCH;
        }
        return FLUTE_OK;
    }

    if (p.family == kFamilyPersistM) {
        PersistMKernel fn = num_bits == 4 ? (dtype == 0 ? persistm_kernel_b4_f16(t.tile_p, oa.lg, p.slabs_per_wave, p.k_chunks, p.waves, p.one_shot)
                                                        : persistm_kernel_b4_bf16(t.tile_p, oa.lg, p.slabs_per_wave, p.k_chunks, p.waves, p.one_shot))
                                          : (dtype == 0 ? persistm_kernel_b2_f16(t.tile_p, oa.lg, p.slabs_per_wave, p.k_chunks, p.waves, p.one_shot)
                                                        : persistm_kernel_b2_bf16(t.tile_p, oa.lg, p.slabs_per_wave, p.k_chunks, p.waves, p.one_shot));
        if (!fn) return FLUTE_ERR_TEMPLATE_ID;
        if (ensure_lds(reinterpret_cast<const void*>(fn), p.lds_bytes)) return FLUTE_ERR_LAUNCH;
        const uint32_t* q32 = reinterpret_cast<const uint32_t*>(Q);
        const uint32_t* qm2 = reinterpret_cast<const uint32_t*>(QM2);
        int nsets = ceil_div(N / 16, p.slabs_per_wave);
        void* kargs[] = {&q32, &S, &A, &qm2, &D, &N, &K, &M, &nsets};
        if (hipLaunchKernel(reinterpret_cast<const void*>(fn), dim3(p.grid), dim3(p.block), kargs, p.lds_bytes, st) != hipSuccess) {
            (void)hipGetLastError();
            return FLUTE_ERR_LAUNCH;
        }
        return FLUTE_OK;
    }

    if (p.family == kFamilyFastM) {
        FastMKernel fn = fastm_kernel_b4(dtype, t.tile_p, p.waves, p.ring_depth, oa.lg, p.slabs_per_wave);
        if (!fn) return FLUTE_ERR_TEMPLATE_ID;
        if (ensure_lds(reinterpret_cast<const void*>(fn), p.lds_bytes)) return FLUTE_ERR_LAUNCH;
        const uint32_t* q32 = reinterpret_cast<const uint32_t*>(Q);
        const uint32_t* qm2 = reinterpret_cast<const uint32_t*>(QM2);
        uint64_t* stamps = nullptr;
#ifdef FLUTE_STAMPS
        if (workspace && workspace_bytes >= kXwgFlagBytes + (size_t)p.grid * p.waves * 128)
            stamps = reinterpret_cast<uint64_t*>(reinterpret_cast<char*>(workspace) + kXwgFlagBytes);
#endif
        void* kargs[] = {&q32, &S, &A, &qm2, &D, &N, &M, &stamps};
        if (hipLaunchKernel(reinterpret_cast<const void*>(fn), dim3(p.grid), dim3(p.block), kargs, p.lds_bytes, st) != hipSuccess) {
            (void)hipGetLastError();
            return FLUTE_ERR_LAUNCH;
        }
        return FLUTE_OK;
    }

    if (p.family == 0 && p.one_shot == 4) {
        FastKernel fn = fast_kernel_b4(dtype, t.tile_p, p.waves, p.kw, p.ring_depth, p.m_block);
        if (!fn) return FLUTE_ERR_TEMPLATE_ID;
        if (ensure_lds(reinterpret_cast<const void*>(fn), p.lds_bytes)) return FLUTE_ERR_LAUNCH;
        const uint32_t* q32 = reinterpret_cast<const uint32_t*>(Q);
        const uint32_t* qm2 = reinterpret_cast<const uint32_t*>(QM2);
        int lg = oa.lg;
        uint64_t* stamps = nullptr;
#ifdef FLUTE_STAMPS
        if (workspace && workspace_bytes >= kXwgFlagBytes + (size_t)p.grid * p.waves * 128)
            stamps = reinterpret_cast<uint64_t*>(reinterpret_cast<char*>(workspace) + kXwgFlagBytes);
#endif
        void* kargs[] = {&q32, &S, &A, &qm2, &D, &N, &lg, &M, &stamps};
        if (hipLaunchKernel(reinterpret_cast<const void*>(fn), dim3(p.grid), dim3(p.block), kargs, p.lds_bytes, st) != hipSuccess) {
            (void)hipGetLastError();
            return FLUTE_ERR_LAUNCH;
        }
        return FLUTE_OK;
    }

    if (p.family == 0 && p.one_shot) {
        OneKernel fn = nullptr;
        const int had = had_log > 0 ? 1 : 0;
        if (num_bits == 4) fn = dtype == 0 ? oneshot_kernel_b4_f16(t.tile_p, p.m_block, oa.depth, had, oa.pipe)
                                           : oneshot_kernel_b4_bf16(t.tile_p, p.m_block, oa.depth, had, oa.pipe);
        else if (num_bits == 2) fn = dtype == 0 ? oneshot_kernel_b2_f16(t.tile_p, p.m_block, oa.depth, had, oa.pipe)
                                                : oneshot_kernel_b2_bf16(t.tile_p, p.m_block, oa.depth, had, oa.pipe);
        else fn = oneshot_kernel_b3(dtype, t.tile_p, p.m_block, oa.depth, had, oa.pipe);
        if (!fn) return FLUTE_ERR_TEMPLATE_ID;
        if (ensure_lds(reinterpret_cast<const void*>(fn), p.lds_bytes)) return FLUTE_ERR_LAUNCH;
        const uint32_t* q32 = reinterpret_cast<const uint32_t*>(Q);
        const uint32_t* qm2 = reinterpret_cast<const uint32_t*>(QM2);
        uint32_t geo = OneGeo::pack(oa.lg, oa.lkw, oa.upw, oa.pk, oa.ipw, had_log, oneshot_x_in_holes(num_bits, p.m_block, K) ? 1 : 0);
        float hs = had_scale;
        uint64_t* stamps = nullptr;
#ifdef FLUTE_STAMPS
        // behind the xwg state words (bytes [0, 64 KB) must stay zero between calls)
        if (workspace && workspace_bytes >= kXwgFlagBytes + (size_t)p.grid * p.waves * 128)
            stamps = reinterpret_cast<uint64_t*>(reinterpret_cast<char*>(workspace) + kXwgFlagBytes);
#endif
        void* kargs[] = {&q32, &S, &A, &qm2, &K, &N, &geo, &M, &D, &hs, &stamps};
        if (hipLaunchKernel(reinterpret_cast<const void*>(fn), dim3(p.grid), dim3(p.block), kargs, p.lds_bytes, st) !=
            hipSuccess) {
            (void)hipGetLastError();
            return FLUTE_ERR_LAUNCH;
        }
        return FLUTE_OK;
    }

    if (p.family == 0) {
        sa.A = A; sa.Q = reinterpret_cast<const uint32_t*>(Q); sa.D = D; sa.S = S;
        sa.QM2 = reinterpret_cast<const uint32_t*>(QM2);
        sa.partial = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + kXwgFlagBytes);
        sa.had_log = had_log; sa.had_scale = had_scale; sa.m0 = 0;
        StreamKernel fn = pick_stream_kernel(num_bits, dtype, t.tile_p, p.m_block, p.ring_depth, 0);
        if (!fn) return FLUTE_ERR_TEMPLATE_ID;
        if (ensure_lds(reinterpret_cast<const void*>(fn), p.lds_bytes)) return FLUTE_ERR_LAUNCH;
        void* kargs[] = {&sa};
        if (hipLaunchKernel(reinterpret_cast<const void*>(fn), dim3(p.grid), dim3(p.block), kargs,
                            p.lds_bytes, st) != hipSuccess) {
            (void)hipGetLastError();
            return FLUTE_ERR_LAUNCH;
        }
        if (p.splitk > 1)
            return splitk_reduce_dispatch(dtype, sa.partial, D, (size_t)M * N, p.splitk, st);
        return FLUTE_OK;
    }

    if (p.family == kFamilySplitK) {
        SplitKArgs b;
        memset(&b, 0, sizeof(b));
        b.A = A; b.Q = reinterpret_cast<const uint32_t*>(Q); b.D = D; b.S = S;
        b.QM2 = reinterpret_cast<const uint32_t*>(QM2);
        b.state = reinterpret_cast<uint32_t*>(workspace);
        b.partial = workspace ? reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + kXwgFlagBytes) : nullptr;
        b.M = M; b.N = N; b.K = K; b.G = K / group_size; b.lg = ilog2(group_size);
        b.tiles_m = ceil_div(M, p.m_tiles * 16);
        b.splitk = p.splitk; b.k_per_split = p.k_per_split;
        b.pair_lg = -1; b.pair_c8 = 0; b.pair_e = 0;
        const int tile_cols = 256 / p.kw;                      // kw = K parts per workgroup: 128- / 64-column tiles
        // m_block = E: the E row tiles of a column tile that run as consecutive blocks of ONE XCD (0 / 1: natural order)
        const int E = p.m_block;
        if (p.splitk == 1 && E >= 2 && b.tiles_m % E == 0 && ((b.tiles_m / E) & (b.tiles_m / E - 1)) == 0) {
            b.pair_e = ilog2(E);
            b.pair_lg = ilog2(b.tiles_m / E);
            b.pair_c8 = ((b.tiles_m / E) * (N / tile_cols)) & ~7;
        }
        SplitKKernel fn = splitk_kernel(num_bits, dtype, t.tile_p, p.m_tiles, p.kw);
        if (!fn) return FLUTE_ERR_TEMPLATE_ID;
        if (ensure_lds(reinterpret_cast<const void*>(fn), p.lds_bytes)) return FLUTE_ERR_LAUNCH;
        void* kargs[] = {&b};
        if (hipLaunchKernel(reinterpret_cast<const void*>(fn), dim3(p.grid), dim3(p.block), kargs, p.lds_bytes, st) !=
            hipSuccess) {
            (void)hipGetLastError();
            return FLUTE_ERR_LAUNCH;
        }
        return FLUTE_OK;
    }

    if (p.family == kFamilyBlock) {
        BlockArgs b;
        memset(&b, 0, sizeof(b));
        b.A = A; b.Q = reinterpret_cast<const uint32_t*>(Q); b.D = D; b.S = S;
        b.QM2 = reinterpret_cast<const uint32_t*>(QM2);
        b.partial = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + kXwgFlagBytes);
        b.M = M; b.N = N; b.K = K; b.G = K / group_size; b.lg = ilog2(group_size);
        const int bm = block_rows(p.m_block);
        b.tiles_m = ceil_div(M, bm); b.tiles_n = N / 256;
        b.splitk = p.splitk; b.k_per_split = p.k_per_split;
        // XCD x (block id % 8) owns a contiguous range of row blocks (their activations then stay in its L2
        // while the weights stream through), else of column blocks
        b.order = (b.tiles_m % 8 == 0) ? 1 : ((b.tiles_n % 8 == 0) ? 2 : 0);
        b.state = (p.splitk > 1 && p.splitk_mode == 1) ? reinterpret_cast<uint32_t*>(workspace) : nullptr;
        BlockKernel fn = (num_bits == 2) ? block_kernel_b2(dtype, t.tile_p, p.m_block)
                         : (num_bits == 3) ? block_kernel_b3(dtype, t.tile_p, p.m_block) : block_kernel_b4(dtype, t.tile_p, p.m_block);
        if (!fn) return FLUTE_ERR_TEMPLATE_ID;
        if (ensure_lds(reinterpret_cast<const void*>(fn), p.lds_bytes)) return FLUTE_ERR_LAUNCH;
        void* kargs[] = {&b};
        if (hipLaunchKernel(reinterpret_cast<const void*>(fn), dim3(p.grid), dim3(p.block), kargs, p.lds_bytes, st) !=
            hipSuccess) {
            (void)hipGetLastError();
            return FLUTE_ERR_LAUNCH;
        }
        if (p.splitk > 1 && p.splitk_mode == 0) return splitk_reduce_dispatch(dtype, b.partial, D, (size_t)M * N, p.splitk, st);
        return FLUTE_OK;
    }

    QGemmArgs a;
    a.A = A; a.Q = reinterpret_cast<const uint32_t*>(Q); a.D = D; a.S = S;
    a.QM2 = reinterpret_cast<const uint32_t*>(QM2);
    a.partial = workspace ? reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + kXwgFlagBytes) : nullptr;
    a.M = M; a.N = N; a.K = K; a.G = K / group_size;
    a.lg = ilog2(group_size);
    a.units = N / ((num_bits == 3) ? 16 : 16 / num_bits);
    a.splitk = p.splitk; a.k_per_split = p.k_per_split; a.kw = p.kw; a.m0 = 0;
    a.lut_shift = 0;
    a.lds_budget = kMaxLds;
    a.lkw = ilog2(p.kw);
    a.state = (p.splitk > 1 && p.splitk_mode == 1) ? reinterpret_cast<uint32_t*>(workspace) : nullptr;
    a.had_log = had_log;
    a.had_scale = had_scale;
    for (int i = 0; i < 10; ++i) a.geo[i] = 0;
    {
        const TileGeom g = tile_geom(num_bits, p.m_block, p.m_tiles, p.slabs_per_wave, p.waves, kMaxLds);
        a.geo[0] = g.depth; a.geo[1] = g.scale_bytes; a.geo[2] = g.slot_bytes; a.geo[3] = g.wave_bytes;
        a.geo[4] = ceil_div(M, p.m_tiles * 16);
        // slab groups a multiple of the 8 XCDs: row tiles of one slab share an XCD (qgemm_tile.h) - as long
        // as the activations (which every XCD then reads in full) are the smaller operand
        a.geo[5] = (((int)p.grid / p.splitk / a.geo[4]) % 8 == 0 && (long)M * 32 <= (long)num_bits * N) ? 1 : 0;
    }

    QGemmKernel fn = pick_kernel(p.family, num_bits, dtype, t.tile_p, p.m_block, p.m_tiles, p.slabs_per_wave);
    if (!fn) return FLUTE_ERR_TEMPLATE_ID;
    if (ensure_lds(reinterpret_cast<const void*>(fn), p.lds_bytes)) return FLUTE_ERR_LAUNCH;

    void* kargs[] = {&a};
    if (hipLaunchKernel(reinterpret_cast<const void*>(fn), dim3(p.grid), dim3(p.block), kargs,
                        p.lds_bytes, st) != hipSuccess) {
        (void)hipGetLastError();
        return FLUTE_ERR_LAUNCH;
    }
    if (p.splitk > 1 && p.splitk_mode == 0)
        return splitk_reduce_dispatch(dtype, a.partial, D, (size_t)M * N, p.splitk, st);
    return FLUTE_OK;
}

int flute_hadamard(int dtype, const void* in, void* out, size_t numel, uint32_t had_size,
                   void* stream) {
    if (!in || !out) return numel == 0 ? FLUTE_OK : FLUTE_ERR_NULL;
    return hadamard_dispatch(dtype, in, out, numel, had_size, reinterpret_cast<hipStream_t>(stream));
}

int flute_unpack(int num_bits, int template_id, int N, int K, const void* Q, void* W,
                 void* stream) {
    flute_template_info t;
    if (num_bits != 2 && num_bits != 3 && num_bits != 4) return FLUTE_ERR_NUM_BITS;
    if (!decode_template(num_bits, template_id, &t)) return FLUTE_ERR_TEMPLATE_ID;
    if (num_bits == 3 && t.tile_p != 32) return FLUTE_ERR_TEMPLATE_ID;
    const int J = (num_bits == 3) ? 16 : 16 / num_bits;
    if (N < 1 || K < 2 || N % (J * t.tile_p) || K % 2) return FLUTE_ERR_SHAPE;
    if (!Q || !W) return FLUTE_ERR_NULL;
    return unpack_dispatch(num_bits, t.tile_p, N, K, Q, W, reinterpret_cast<hipStream_t>(stream));
}

int flute_debug_stream_read(const void* src, void* sink, size_t bytes, int bytes_per_wave,
                            int grid, int block, void* stream) {
    if (!src || !sink || bytes_per_wave < 8192 || bytes_per_wave % 8192) return FLUTE_ERR_SHAPE;
    return stream_read_dispatch(src, sink, bytes, bytes_per_wave, grid, block,
                                reinterpret_cast<hipStream_t>(stream));
}

int flute_debug_timestamp(void* dst, void* stream) {
    if (!dst) return FLUTE_ERR_NULL;
    return timestamp_dispatch(dst, reinterpret_cast<hipStream_t>(stream));
}

}  // extern "C"
