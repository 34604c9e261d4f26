// Decode-regime kernel, second generation (up to 4 rows per pass; the planner takes it for M <= 2, and for M <= 4 on
// small layers - api.hip): HBM-bound streaming LUT-dequant GEMV.
//
// Replaces, for small M, the reference's qgemm_device main loop (flute/csrc/qgemm_kernel.hpp:617-712)
// + Stream-K fixup (tile_scheduler_utils.hpp:58-211).  CDNA4 design, not a translation.  The round-1
// kernel (qgemm_decode.h) spent 4.6 VALU instructions per weight pair (2 are necessary), 2.2 us in its
// prologue and ran 17 us of pure overhead on a 117 MB layer with both its loads and its lookups removed
// (profiles/r01_ring_depth.txt).  This one is built around three rules:
//   1. nothing in the per-piece loop but the necessary work: per 16-B piece of packed weights (16 pairs x J
//      columns) a lane executes 16 v_perm (table address) + 16 ds_read_b32 (pair lookup) + 16 v_dot2 per
//      row + J v_fma_mix (group scale, fp16 operand read directly) + 2 address adds.  Weights arrive by
//      `buffer_load_dwordx4 ... offen` with the row base in the descriptor and the position in a scalar
//      offset (no per-load VALU address arithmetic, out-of-range reads return 0);
//   2. every wave is independent after the prologue barrier: it stages ITS OWN group scales (wave-private
//      LDS, prefetched one unit ahead through the same in-order load queue), streams its unit rows through a
//      register ring that runs across unit boundaries, reduces with DPP and stores its J outputs - no
//      barrier per unit unless K is split across waves (narrow layers) or X is staged in K chunks;
//   3. the launch shape is free: any number of waves per workgroup (not a power of two) and any in-workgroup
//      K split, chosen by the host planner so that units x K divides evenly over 256 CUs
//      (8192 x 28672: 14 waves x 2 units each instead of 16 waves x 1.75).
// LDS: pair table (b=4: 256 entries x 32 copies on a 256-B stride, address = ONE v_perm_b32
// {0, 0, field byte, (lane % 32) * 4}; b=2: byte table, two columns per ds_read_b64; b=3: 64 entries x 32
// copies), the activations of the K range ([MB][KX] T, zero padded to 512-k pieces; the Hadamard
// pre-rotation of flute.qgemm_hadamard is applied on the way in, qgemm.cpp:201-244), per-wave scale images
// and the K-split reduction buffer.
//
// Arithmetic: acc(fp32) += s_g * sum_{8 k of the lane's piece} x_k * lut_k  (the group scale is applied in
// fp32 to an 8-k partial sum): identical to the reference's round_T(lut * s) contract
// (packbits_utils.hpp:139) on one-hot inputs, within 2^-11 relative per term otherwise.
#pragma once
#include <type_traits>
#include <utility>

#include "common.h"
#include "fwht.h"

namespace flute_amd {

// largest workgroup a variant may be launched with (1024 threads = 128 VGPRs): the variants that keep many
// accumulators (J x MB) next to a deep ring get 512 threads (256 VGPRs) instead of spilling
__host__ __device__ constexpr int stream_max_threads(int bits, int mb, int depth) {
    if (bits == 4) return (mb == 4 && depth > 2) ? 512 : 1024;
    if (bits == 2) return (mb == 1 || (mb == 2 && depth <= 2)) ? 1024 : 512;
    return 512;        // 3 bits: 16 columns per unit, three planes per ring slot
}
__host__ __device__ constexpr int stream_lut_bytes(int bits) { return bits == 3 ? 64 * 128 : 65536; }

struct StreamArgs {
    const void* A;          // [M,K] T
    const uint32_t* Q;      // [P,K/2] packed
    void* D;                // [M,N] T
    const void* S;          // [N,G] T
    const uint32_t* QM2;    // [4^b] pair table
    float* partial;         // [splitk][M][N] fp32 when splitk > 1
    int M, N, K, G, lg;
    int units;              // N / J
    int ngroups;            // ceil(units / upw)
    int upw, kw, lkw;       // units per workgroup visit, waves per unit (power of two), log2
    int nwg;                // workgroups per K split
    int vis_q, vis_r;       // visits of workgroup wg: vis_q + (wg < vis_r)
    int splitk, k_per_split;    // k_per_split: multiple of 512 (== K rounded up when splitk == 1)
    int kc, nchunks;        // activations staged kc k at a time (multiple of 512); nchunks = ceil(k range / kc)
    int kx;                 // LDS row stride of the staged activations, in elements (multiple of 512)
    int x_off, s_off, red_off, s_wave_bytes;     // LDS carve
    int s_fast;             // 1: scale rows are 16-B aligned granules for every segment of this launch
    int had_log;            // fused Hadamard block (0 = none, <= 9)
    float had_scale;
    int m0;
};

// ---- hidden loads (hipcc must neither count nor wait for them; see common.h "weight ring") ----
typedef int srd_t __attribute__((ext_vector_type(4)));

// raw buffer descriptor (stride 0): base must be wave-uniform (kernel arguments, blockIdx and
// readfirstlane'd wave ids only), otherwise the "s" constraint of the loads below does not compile
__device__ __forceinline__ srd_t make_srd(const void* base, uint32_t bytes) {
    const uint64_t p = reinterpret_cast<uint64_t>(base);
    srd_t r;
    r.x = (int)(uint32_t)p;
    r.y = (int)(uint32_t)((p >> 32) & 0xffffu);
    r.z = (int)bytes;
    r.w = 0x00020000;
    return r;
}
// 16 B per lane from descriptor base + voff (per lane) + soff (wave-uniform).  The leading s_nop covers the
// VALU-write (v_readfirstlane) -> VMEM-read hazard on the scalar operands, which hipcc does not pad for asm.
__device__ __forceinline__ ring16_t buf_load16(uint32_t voff, srd_t srd, uint32_t soff) {
    ring16_t v;
    asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(v) : "v"(voff), "s"(srd), "s"(soff) : "memory");
    return v;
}
// the same with the non-temporal cache policy: weights are streamed once (MI355X_MICROARCH.md, nt-weights)
__device__ __forceinline__ ring16_t buf_load16_nt(uint32_t voff, srd_t srd, uint32_t soff) {
    ring16_t v;
    asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, %3 offen nt" : "=v"(v) : "v"(voff), "s"(srd), "s"(soff) : "memory");
    return v;
}
__device__ __forceinline__ uint32_t buf_load4(uint32_t voff, srd_t srd) {
    uint32_t v;
    asm volatile("s_nop 4\n\tbuffer_load_dword %0, %1, %2, 0 offen" : "=v"(v) : "v"(voff), "s"(srd) : "memory");
    return v;
}
template <int N> __device__ __forceinline__ void vm_wait_regs(ring16_t& a) {
    asm volatile("s_waitcnt vmcnt(%1)" : "+v"(a) : "n"(N) : "memory");
}
template <int N> __device__ __forceinline__ void vm_wait_regs(uint32_t& a) {
    asm volatile("s_waitcnt vmcnt(%1)" : "+v"(a) : "n"(N) : "memory");
}

// hidden LDS lookups: issued by asm (hipcc inserts no wait), released by lds_lookup_wait
__device__ __forceinline__ uint32_t lds_lookup32(uint32_t addr) {
    uint32_t v;
    asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(addr) : "memory");
    return v;
}
__device__ __forceinline__ u32x2_t lds_lookup64(uint32_t addr) {
    u32x2_t v;
    asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(addr) : "memory");
    return v;
}
template <typename V> __device__ __forceinline__ void lds_lookup_wait(V (&v)[16]) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]),
                   "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15])
                 : : "memory");
}

template <typename V> __device__ __forceinline__ void lds_lookup_wait8(V (&v)[8]) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7])
                 : : "memory");
}

typedef __attribute__((address_space(3))) const uint16_t lds_cu16_t;
__device__ __forceinline__ uint32_t lds_ld16(uint32_t a) { return *(lds_cu16_t*)(uintptr_t)a; }

// sum over the 64 lanes (valid in lane 0): DPP row reductions + three v_readlane
__device__ __forceinline__ float wave_sum64(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
    const int iv = __builtin_bit_cast(int, v);
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 48));
    return (v + r1) + (r2 + r3);
}

template <typename T> __device__ __forceinline__ float scale_to_float(uint32_t raw16);
template <> __device__ __forceinline__ float scale_to_float<F16>(uint32_t raw16) {
    return (float)__builtin_bit_cast(_Float16, (uint16_t)raw16);          // folded into v_fma_mix_f32
}
template <> __device__ __forceinline__ float scale_to_float<BF16>(uint32_t raw16) {
    return __builtin_bit_cast(float, raw16 << 16);
}

// D = ring depth in pieces (one piece = one 1-KiB wave-wide load per plane = 512 k of one unit).
// ONE = one-shot variant for the latency-bound launches (every workgroup visits ONE unit group, no K chunks,
// at most D pieces per wave): all of the wave's weights are requested by the prologue, there is no refill, no
// load cursor and no scale prefetch - a wave issues an instruction every four cycles at best, so on an 8 MB
// layer the instruction count of this path IS the launch time once the loads are out.
template <typename T, int BITS, int TILEP, int MB, int D, bool ONE = false>
__global__ __launch_bounds__(stream_max_threads(BITS, MB, D)) void qgemv_stream_kernel(const StreamArgs args) {
    // All kernel arguments are fetched in ONE batch of scalar loads and made opaque: hipcc otherwise treats
    // every field as rematerialisable and re-reads single dwords from the kernarg segment (s_load + wait, ~100
    // cycles each, a dozen times in the prologue) instead of keeping or lane-spilling them.
    StreamArgs a = args;
    {
#define FLUTE_OPAQUE(x) asm volatile("" : "+s"(x))
        FLUTE_OPAQUE(a.A); FLUTE_OPAQUE(a.Q); FLUTE_OPAQUE(a.D); FLUTE_OPAQUE(a.S); FLUTE_OPAQUE(a.QM2);
        FLUTE_OPAQUE(a.partial); FLUTE_OPAQUE(a.M); FLUTE_OPAQUE(a.N); FLUTE_OPAQUE(a.K); FLUTE_OPAQUE(a.G);
        FLUTE_OPAQUE(a.lg); FLUTE_OPAQUE(a.units); FLUTE_OPAQUE(a.upw); FLUTE_OPAQUE(a.kw);
        FLUTE_OPAQUE(a.lkw); FLUTE_OPAQUE(a.nwg); FLUTE_OPAQUE(a.vis_q); FLUTE_OPAQUE(a.vis_r); FLUTE_OPAQUE(a.splitk);
        FLUTE_OPAQUE(a.k_per_split); FLUTE_OPAQUE(a.kc); FLUTE_OPAQUE(a.nchunks); FLUTE_OPAQUE(a.kx); FLUTE_OPAQUE(a.x_off);
        FLUTE_OPAQUE(a.s_off); FLUTE_OPAQUE(a.red_off); FLUTE_OPAQUE(a.s_wave_bytes); FLUTE_OPAQUE(a.s_fast);
        FLUTE_OPAQUE(a.had_log); FLUTE_OPAQUE(a.had_scale); FLUTE_OPAQUE(a.m0);
#undef FLUTE_OPAQUE
    }
    using L = Layout<BITS>;
    using NT = Num<T>;
    constexpr int J = L::J;
    constexpr int NP = L::NPLANES;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    // the v_perm-built table addresses are absolute: the table must sit at LDS byte 0
    if (lds_base_of(smem) != 0) __builtin_trap();
#ifdef FLUTE_STAMPS   // development build: 100 MHz wall-clock stamps per wave into the (unused) workspace
    uint64_t stamp[8];
    for (int i = 0; i < 8; ++i) stamp[i] = 0;
    stamp[0] = wall_clock64();
#define FLUTE_SSTAMP(i) if (stamp[i] == 0) stamp[i] = wall_clock64()
#else
#define FLUTE_SSTAMP(i)
#endif

    const int tid = threadIdx.x;
    const int nthr = blockDim.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kw = a.kw, lkw = a.lkw;
    const int ul = wave >> lkw;
    const int kpart = wave & (kw - 1);
    const int lg = a.lg;
    const int KX = a.kx;
    const uint32_t lane16 = (uint32_t)lane * 16u;

    int split = 0, wg = blockIdx.x;
    if (a.splitk > 1) { split = blockIdx.x % a.splitk; wg = blockIdx.x / a.splitk; }
    const int kbeg = split * a.k_per_split;
    const int kend = min(a.K, kbeg + a.k_per_split);
    const int nchunks = a.nchunks;
    const int nvis = a.vis_q + (wg < a.vis_r ? 1 : 0);
    const int nseg = nvis * nchunks;
    const uint32_t row_bytes = (uint32_t)a.K * 2u;                 // one Q32 row: K/2 words

    // ---- geometry.  A segment = (visit v, K chunk c) of this wave: np 512-k pieces of unit
    // (wg + v * nwg) * upw + ul starting at k0.  The K side depends on the chunk only: for the usual
    // unchunked launch it is computed once. ----
    struct Geo { int np, k0, p0; };
    auto chunk_geo = [&](int c) -> Geo {
        Geo g;
        const int ck0 = kbeg + c * a.kc;
        const int clen = min(a.kc, kend - ck0);
        const int pc = (clen + 511) >> 9;                          // pieces in the chunk
        const int pk = (pc + kw - 1) >> lkw;                       // pieces per wave of the K split
        g.p0 = kpart * pk;
        g.np = max(0, min(pk, pc - g.p0));
        g.k0 = ck0 + g.p0 * 512;
        return g;
    };
    const Geo geo0 = chunk_geo(0);
    auto unit_of = [&](int v) { return (wg + v * a.nwg) * a.upw + ul; };
    auto slots_of = [&](int np) { return max(D, (np + D - 1) / D * D); };   // every segment: >= D ring slots

    // ---- prologue loads, staging data first (loads return in order): table words, activations of chunk 0,
    // this wave's first scale block, then the ring ----
    constexpr int ENT = (BITS == 3) ? 64 : 256;
    constexpr int EPB = (BITS == 2) ? 256 : 128;                   // bytes written per table entry (32 copies)
    constexpr int ESTRIDE = (BITS == 3) ? 128 : 256;               // entry stride (b=4: upper half of the stride unused)
    constexpr int EPIECES = EPB / 16;                              // 16-B pieces per entry
    // thread t owns entry t % ENT and writes pieces t / ENT, t / ENT + P, ... of it (P = threads / ENT,
    // planned >= 1): ONE table word per thread (two for the 2-bit byte table), whatever the workgroup size
    const srd_t lut_srd = make_srd(a.QM2, (uint32_t)(4 << (2 * BITS)));
    const int lut_e = tid & (ENT - 1);
    uint32_t lut_v0 = buf_load4((uint32_t)((BITS == 2) ? (lut_e & 15) : lut_e) * 4u, lut_srd);
    uint32_t lut_v1 = 0;
    if constexpr (BITS == 2) lut_v1 = buf_load4((uint32_t)(lut_e >> 4) * 4u, lut_srd);

    // activations: row m of the chunk is staged by all threads, XPR 16-B pieces per thread and row in
    // registers (the rest by plain loads at commit time).  One descriptor over the whole of A: reads past the
    // end of A return 0; k >= K inside a row is zeroed at commit.
    constexpr int XPR = (MB == 4) ? 1 : 2;
    const srd_t x_srd = make_srd(a.A, (uint32_t)min((size_t)a.M * a.K * 2, (size_t)0xfffffff0u));
    const int xrow_pieces = KX >> 3;
    ring16_t xv[MB][XPR];
    auto x_voff = [&](int m, int pidx, int c) -> uint32_t {
        const int k = kbeg + c * a.kc + pidx * 8;
        return (pidx < xrow_pieces && k < a.K) ? (uint32_t)(((size_t)min(a.m0 + m, a.M - 1) * a.K + k) * 2) : 0x80000000u;
    };
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int r = 0; r < XPR; ++r) xv[m][r] = buf_load16(x_voff(m, r * nthr + tid, 0), x_srd, 0);

    // this wave's scale block of a segment: LDS image [granule c][column j][8 groups] T, granule = 16 B.
    // Lane q = lane + 64 r fetches granule (j = q >> lgn, c = q & (2^lgn - 1)), 2^lgn >= granules per column.
    // Fast path: one 16-B load per granule (SPR registers per lane); rows of S that are not 16-B aligned, the
    // ragged last granule of a row and very long K ranges are staged element-wise at commit time.
    constexpr int SPR = (BITS == 3) ? 4 : 2;
    const srd_t s_srd = make_srd(a.S, (uint32_t)min((size_t)a.N * a.G * 2, (size_t)0xfffffff0u));
    const uint32_t sbase = (uint32_t)a.s_off + (uint32_t)wave * (uint32_t)a.s_wave_bytes;
    ring16_t sv[SPR];
    struct SGeo { int g0, ng, ngran, lgn; };
    auto sgeo_of = [&](const Geo& g) -> SGeo {
        SGeo s;
        s.g0 = g.k0 >> lg;
        s.ng = (g.np > 0) ? ((min(g.k0 + g.np * 512, kend) - 1) >> lg) - s.g0 + 1 : 0;
        // granules are staged for every lane of every piece, also the lanes of a ragged last piece that lie
        // past the end of K (zero-filled: their activations are zero, but 0 x stale-LDS-NaN would not be)
        s.ngran = (g.np * (512 >> lg) + 7) >> 3;
        s.lgn = (s.ngran > 1) ? 32 - __builtin_clz((unsigned)(s.ngran - 1)) : 0;
        return s;
    };
    SGeo sg = sgeo_of(geo0);
    // per-lane role: relative source offset (bytes, or out of range) and LDS image offset of granule r
    uint32_t s_rel[SPR], s_img[SPR];
    auto s_roles = [&]() {
#pragma unroll
        for (int r = 0; r < SPR; ++r) {
            const int q = lane + 64 * r;
            const int j = q >> sg.lgn;
            const int c = q & ((1 << sg.lgn) - 1);
            const bool mine = j < J && c < sg.ngran;
            const bool fast = mine && a.s_fast && sg.g0 + c * 8 + 8 <= a.G;
            s_rel[r] = fast ? (uint32_t)(j * TILEP * a.G + c * 8) * 2u : 0x80000000u;
            s_img[r] = mine ? (uint32_t)(c * J + j) * 16u : 0xffffffffu;
        }
    };
    auto s_issue = [&](int unit) {                                 // unconditional loads (the count stays uniform)
        const int col0 = unit_col0<BITS, TILEP>(min(unit, a.units - 1));
        const uint32_t ubase = (unit < a.units) ? (uint32_t)(col0 * a.G + sg.g0) * 2u : 0x80000000u;
#pragma unroll
        for (int r = 0; r < SPR; ++r) sv[r] = buf_load16(ubase + s_rel[r], s_srd, 0);   // either part out of range: 0
    };
    auto s_commit = [&](int unit) {
        const uint16_t* S = reinterpret_cast<const uint16_t*>(a.S);
        const int col0 = unit_col0<BITS, TILEP>(min(unit, a.units - 1));
        auto slow_granule = [&](int j, int c) -> uint4 {           // element loads, zero fill
            uint16_t h[8];
            const uint16_t* sp = S + (size_t)(col0 + j * TILEP) * a.G + sg.g0 + c * 8;
#pragma unroll
            for (int e = 0; e < 8; ++e) h[e] = (sg.g0 + c * 8 + e < a.G && c * 8 + e < sg.ng) ? sp[e] : (uint16_t)0;
            return make_uint4(h[0] | ((uint32_t)h[1] << 16), h[2] | ((uint32_t)h[3] << 16),
                              h[4] | ((uint32_t)h[5] << 16), h[6] | ((uint32_t)h[7] << 16));
        };
#pragma unroll
        for (int r = 0; r < SPR; ++r) {
            if (s_img[r] != 0xffffffffu) {
                uint4 v = make_uint4(sv[r].x, sv[r].y, sv[r].z, sv[r].w);
                if (s_rel[r] == 0x80000000u) {                     // not a 16-B granule: element-wise
                    const int q = lane + 64 * r;
                    v = slow_granule(q >> sg.lgn, q & ((1 << sg.lgn) - 1));
                }
                *reinterpret_cast<uint4*>(smem + sbase + s_img[r]) = v;
            }
        }
        for (int q = lane + 64 * SPR; (q >> sg.lgn) < J; q += 64) {    // columns the SPR x 64 lanes did not reach
            const int j = q >> sg.lgn;
            const int c = q & ((1 << sg.lgn) - 1);
            if (c < sg.ngran) *reinterpret_cast<uint4*>(smem + sbase + (uint32_t)(c * J + j) * 16u) = slow_granule(j, c);
        }
    };

    // ---- load cursor: runs D slots ahead of the compute cursor, across segment boundaries.  The slot
    // stream is the concatenation of the wave's segments, each padded to a whole number (>= 1) of groups of
    // D slots (the ring slot of a piece is then a compile-time constant and at least D ring loads separate
    // a scale prefetch from its use); padding and exhausted slots are out-of-range loads (return 0, fetch
    // nothing).  EVERY ring / scale load is unconditional straight-line code and every counted wait is ONE
    // statement: hipcc must never see an in-flight destination register at a control-flow merge
    // (tools/audit_asm_loads.py checks the compiled code for register copies between a load and its wait). ----
    int lv = 0, lc = 0, ls = 0, lp = 0, lnp = 0, lnpad = D;        // visit, chunk, segment ordinal, slot, pieces, slots
    srd_t lsrd[NP];
    uint32_t lsoff = 0;
    auto l_enter = [&]() {
        const Geo g = (nchunks == 1) ? geo0 : chunk_geo(lc);
        const int unit = unit_of(lv);
        const bool live = ls < nseg && unit < a.units;
        lnp = live ? g.np : 0;
        lnpad = slots_of(lnp);
        lsoff = (uint32_t)g.k0 * 2u;
        const int urow = min(unit, a.units - 1);
#pragma unroll
        for (int pl = 0; pl < NP; ++pl)
            lsrd[pl] = make_srd(reinterpret_cast<const char*>(a.Q) + (size_t)unit_row<BITS, TILEP>(urow, pl, a.N) * row_bytes,
                                live ? row_bytes : 0u);            // idle / exhausted: zero-length descriptor
        lp = 0;
    };
    ring16_t q[D][NP];
    auto ring_issue = [&](int i) {
        // the position travels in the VECTOR offset: only voffset takes part in the descriptor's range check
        // (the scalar offset is excluded from it), and the range check is what turns padding slots and reads
        // past the end of a ragged row into zeros
        const uint32_t vo = lane16 + ((lp < lnp) ? lsoff : 0x80000000u);
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) {
#ifdef FLUTE_RING_NO_NT
            q[i][pl] = buf_load16(vo, lsrd[pl], 0);
#else
            q[i][pl] = buf_load16_nt(vo, lsrd[pl], 0);
#endif
        }
        lsoff += 1024u;
        ++lp;
    };
    auto l_advance = [&]() {                                       // segments end on group boundaries only
        if (lp == lnpad) {
            ++ls; if (++lc == nchunks) { lc = 0; ++lv; }
            l_enter();
        }
    };

    __builtin_amdgcn_sched_barrier(0);                             // table / activation loads are out: now the weights
    l_enter();
#pragma unroll
    for (int i = 0; i < D; ++i) ring_issue(i);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (!ONE) l_advance();
    s_roles();
    s_issue(unit_of(0));
    FLUTE_SSTAMP(4);
    {
        // loads issued behind the table / activation loads: the scale block and the ring.  The scale block is
        // HBM-cold like the weights; the table and the activations are L2-hot and are written to LDS while it
        // is still in flight (it is waited for at the first segment start below)
        constexpr int NY = D * NP + SPR;                          // younger loads: the ring, then the scale block
        vm_wait_regs<NY>(lut_v0);
        vm_wait_regs<NY>(lut_v1);
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int r = 0; r < XPR; ++r) vm_wait_regs<NY>(xv[m][r]);
    }
    FLUTE_SSTAMP(5);
    // ---- table image ----
    {
        const int P = nthr / ENT;
        const uint32_t v1 = (BITS == 2) ? lut_v1 : lut_v0;
        const uint4 img = make_uint4(lut_v0, v1, lut_v0, v1);
        if (tid < P * ENT) {
            for (int pc = tid / ENT; pc < EPIECES; pc += P)
                *reinterpret_cast<uint4*>(smem + (size_t)lut_e * ESTRIDE + pc * 16) = img;
        }
        if (P == 0) {                                              // fewer threads than table entries (tiny launches)
            for (int e = tid; e < ENT; e += nthr) {
                uint32_t w0, w1;
                if constexpr (BITS == 2) { w0 = a.QM2[e & 15]; w1 = a.QM2[e >> 4]; }
                else { w0 = a.QM2[e]; w1 = w0; }
                for (int pc = 0; pc < EPIECES; ++pc)
                    *reinterpret_cast<uint4*>(smem + (size_t)e * ESTRIDE + pc * 16) = make_uint4(w0, w1, w0, w1);
            }
        }
    }
    // K-split reduction of single-visit launches: arrival counters (one per unit of the workgroup)
    int* arrive = reinterpret_cast<int*>(smem + a.red_off);
    if (kw > 1 && tid < a.upw) arrive[tid] = 0;
    FLUTE_SSTAMP(6);

    // ---- activations of a chunk -> LDS [MB][KX]; fused pre-rotation: the 64 pieces a wave stages are 512
    // consecutive k of one row = whole Hadamard blocks (K % had == 0, had <= 512) ----
    uint16_t* xs = reinterpret_cast<uint16_t*>(smem + a.x_off);
    const uint16_t* A = reinterpret_cast<const uint16_t*>(a.A);
    auto x_commit = [&](int c, bool from_regs) {
#pragma unroll
        for (int m = 0; m < MB; ++m) {
#pragma unroll
            for (int r = 0; r < XPR; ++r) {
                const int pidx = r * nthr + tid;
                if (pidx < xrow_pieces) {                          // wave-uniform: rows are whole 512-k spans
                    uint32_t w[4];
                    if (from_regs) { w[0] = xv[m][r].x; w[1] = xv[m][r].y; w[2] = xv[m][r].z; w[3] = xv[m][r].w; }
                    else {
                        const int k = min(kbeg + c * a.kc + pidx * 8, a.K - 8);
                        const uint4 t = *reinterpret_cast<const uint4*>(A + (size_t)min(a.m0 + m, a.M - 1) * a.K + k);
                        w[0] = t.x; w[1] = t.y; w[2] = t.z; w[3] = t.w;
                    }
                    if (a.had_log > 0) fwht_piece<T>(w, lane, a.had_log, a.had_scale);
                    const bool inside = kbeg + c * a.kc + pidx * 8 < a.K;
                    *reinterpret_cast<uint4*>(xs + (size_t)m * KX + (size_t)pidx * 8) =
                        inside ? make_uint4(w[0], w[1], w[2], w[3]) : make_uint4(0, 0, 0, 0);
                }
            }
            for (int pidx = XPR * nthr + tid; pidx < xrow_pieces; pidx += nthr) {
                const int k = min(kbeg + c * a.kc + pidx * 8, a.K - 8);
                const uint4 t = *reinterpret_cast<const uint4*>(A + (size_t)min(a.m0 + m, a.M - 1) * a.K + k);
                uint32_t w[4] = {t.x, t.y, t.z, t.w};
                if (a.had_log > 0) fwht_piece<T>(w, lane, a.had_log, a.had_scale);
                const bool inside = kbeg + c * a.kc + pidx * 8 < a.K;
                *reinterpret_cast<uint4*>(xs + (size_t)m * KX + (size_t)pidx * 8) =
                    inside ? make_uint4(w[0], w[1], w[2], w[3]) : make_uint4(0, 0, 0, 0);
            }
        }
    };

    // per-lane constants of the piece loop
    const uint32_t lane_off = (BITS == 2) ? (uint32_t)(lane & 31) * 8u : (uint32_t)(lane & 31) * 4u;
    const int gl = (8 * lane) >> lg;                               // group of the lane's 8 k inside a piece
    const uint32_t s_lane = sbase + (uint32_t)((gl >> 3) * J * 16 + (gl & 7) * 2);
    const uint32_t x_lane = (uint32_t)a.x_off + lane16;
    const int gpp = 512 >> lg;                                     // groups per piece

    float acc[J][MB];
    float* red = reinterpret_cast<float*>(smem + a.red_off) + 16;  // behind the arrival counters
    const int W = nthr >> 6;

    // ---- end of a unit: lanes -> wave (DPP) -> [K split: waves -> LDS -> one wave] -> output ----
    auto store_out = [&](int unit, int j, int m, float v) {
        const int row = a.m0 + m;
        if (row < a.M && unit < a.units) {
            const int n = unit_col0<BITS, TILEP>(unit) + j * TILEP;
            if (a.splitk == 1) reinterpret_cast<uint16_t*>(a.D)[(size_t)row * a.N + n] = NT::from_float(v);
            else a.partial[((size_t)split * a.M + row) * a.N + n] = v;
        }
    };
    auto seg_end = [&](int cv, int cc, int unit) {
        if (cc != nchunks - 1) return;
        FLUTE_SSTAMP(2);
        float tot[J][MB];
#pragma unroll
        for (int j = 0; j < J; ++j)
#pragma unroll
            for (int m = 0; m < MB; ++m) tot[j][m] = wave_sum64(acc[j][m]);
        if (kw == 1) {
            if (lane == 0) {
#pragma unroll
                for (int m = 0; m < MB; ++m)
#pragma unroll
                    for (int j = 0; j < J; ++j) store_out(unit, j, m, tot[j][m]);
            }
        } else if (nvis == 1) {
            // single visit (the latency-bound case): no barrier - every wave leaves its partial sums and an
            // arrival tick in LDS (same wave, in order), the wave that arrives last sums and stores
            float* rb = red;
            if (lane == 0) {
#pragma unroll
                for (int j = 0; j < J; ++j)
#pragma unroll
                    for (int m = 0; m < MB; ++m) rb[wave * (J * MB) + j * MB + m] = tot[j][m];
            }
            int ticket = 0;
            // acq_rel: the partial sums above are ordered before the tick, the reads below after it (LDS: a compiler
            // ordering constraint only)
            if (lane == 0) ticket = __hip_atomic_fetch_add(&arrive[ul], 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
            ticket = __builtin_amdgcn_readfirstlane(ticket);
            if (ticket == kw - 1) {
                for (int t = lane; t < J * MB; t += 64) {
                    float sum = 0.f;
                    for (int kp = 0; kp < kw; ++kp) sum += rb[(ul * kw + kp) * (J * MB) + t];
                    store_out(unit, t / MB, t % MB, sum);
                }
            }
        } else {
            float* rb = red + (size_t)(cv & 1) * W * (J * MB);      // double buffered: one barrier per visit
            if (lane == 0) {
#pragma unroll
                for (int j = 0; j < J; ++j)
#pragma unroll
                    for (int m = 0; m < MB; ++m) rb[wave * (J * MB) + j * MB + m] = tot[j][m];
            }
            __syncthreads();
            for (int t = tid; t < a.upw * J * MB; t += nthr) {
                const int ulc = t / (J * MB);
                const int r = t - ulc * (J * MB);
                float sum = 0.f;
                for (int kp = 0; kp < kw; ++kp) sum += rb[(ulc * kw + kp) * (J * MB) + r];
                store_out((wg + cv * a.nwg) * a.upw + ulc, r / MB, r % MB, sum);
            }
        }
    };

    // ---- one piece: 16 k-pairs x J columns per lane.  i = ring slot (compile-time), pch = piece inside the
    // staged chunk (activations), cp = piece inside the segment (scales) ----
    auto compute_piece = [&](auto slot_tag, int pch, int cp) {
        constexpr int i = decltype(slot_tag)::value;
        const uint32_t xa = x_lane + (uint32_t)pch * 1024u;
        const int gp = cp * gpp;
        const uint32_t sa = s_lane + (uint32_t)((gp >> 3) * (J * 16) + (gp & 7) * 2);
        uint32_t xw[MB][4];
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            const uint4 t = lds_ld128(xa + (uint32_t)(m * KX * 2));
            xw[m][0] = t.x; xw[m][1] = t.y; xw[m][2] = t.z; xw[m][3] = t.w;
        }
        uint32_t sc[J];
#pragma unroll
        for (int j = 0; j < J; ++j) sc[j] = lds_ld16(sa + 16u * j);
        float al[J][MB];
#pragma unroll
        for (int j = 0; j < J; ++j)
#pragma unroll
            for (int m = 0; m < MB; ++m) al[j][m] = 0.f;
        // The pair lookups are hidden ds_reads (hipcc otherwise funnels them through one or two
        // registers: address, read, wait, dot, ...): every lookup of a batch is issued back to back,
        // ONE wait releases them all (cdna_hip_programming.md 5.7 form ii).
        if constexpr (BITS == 2) {
#pragma unroll
            for (int hw = 0; hw < 4; hw += 2) {         // two batches of 8 byte lookups (2 columns each)
                u32x2_t v[8];
#pragma unroll
                for (int ww = 0; ww < 2; ++ww)
#pragma unroll
                    for (int jp = 0; jp < 4; ++jp)
                        v[ww * 4 + jp] = lds_lookup64(__builtin_amdgcn_perm(q[i][0][hw + ww], lane_off, 0x0c0c0400u | ((4u + jp) << 8)));
                lds_lookup_wait8(v);
#pragma unroll
                for (int ww = 0; ww < 2; ++ww)
#pragma unroll
                    for (int jp = 0; jp < 4; ++jp)
#pragma unroll
                        for (int m = 0; m < MB; ++m) {
                            al[2 * jp][m] = NT::dot2(v[ww * 4 + jp].x, xw[m][hw + ww], al[2 * jp][m]);
                            al[2 * jp + 1][m] = NT::dot2(v[ww * 4 + jp].y, xw[m][hw + ww], al[2 * jp + 1][m]);
                        }
            }
        } else if constexpr (BITS == 4) {
            uint32_t v[16];
#pragma unroll
            for (int ww = 0; ww < 4; ++ww)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    v[ww * 4 + j] = lds_lookup32(__builtin_amdgcn_perm(q[i][0][ww], lane_off, 0x0c0c0400u | ((4u + j) << 8)));
            lds_lookup_wait(v);
#pragma unroll
            for (int ww = 0; ww < 4; ++ww)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int m = 0; m < MB; ++m) al[j][m] = NT::dot2(v[ww * 4 + j], xw[m][ww], al[j][m]);
        } else {
#pragma unroll
            for (int ww = 0; ww < 4; ++ww) {            // one k-pair position: 16 fields in three planes
                const uint32_t w[3] = {q[i][0][ww], q[i][1][ww], q[i][2][ww]};
                uint32_t v[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) v[j] = lds_lookup32((field<3>(w, j) << 7) | lane_off);
                lds_lookup_wait(v);
#pragma unroll
                for (int j = 0; j < 16; ++j)
#pragma unroll
                    for (int m = 0; m < MB; ++m) al[j][m] = NT::dot2(v[j], xw[m][ww], al[j][m]);
            }
        }
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const float sf = scale_to_float<T>(sc[j]);
#pragma unroll
            for (int m = 0; m < MB; ++m) acc[j][m] = __builtin_fmaf(al[j][m], sf, acc[j][m]);
        }
                    };

    if constexpr (ONE) {
        // ---- one-shot: a single segment, every piece already requested ----
        const int unit = unit_of(0);
        const int cnp = (unit < a.units) ? geo0.np : 0;
        x_commit(0, true);
        __syncthreads();                                           // table / activations visible to every wave
        if constexpr (NP == 1) {
#pragma unroll
            for (int i = 0; i < D; ++i) asm volatile("s_waitcnt vmcnt(0)" : "+v"(q[i][0]) : : "memory");
        } else {
#pragma unroll
            for (int i = 0; i < D; ++i) asm volatile("s_waitcnt vmcnt(0)" : "+v"(q[i][0]), "+v"(q[i][1]), "+v"(q[i][2]) : : "memory");
        }
#pragma unroll
        for (int r = 0; r < SPR; ++r) vm_wait_regs<0>(sv[r]);
        s_commit(unit);
        FLUTE_SSTAMP(1);
#pragma unroll
        for (int j = 0; j < J; ++j)
#pragma unroll
            for (int m = 0; m < MB; ++m) acc[j][m] = 0.f;
        [&]<int... I>(std::integer_sequence<int, I...>) {
            ((I < cnp ? compute_piece(std::integral_constant<int, I>{}, geo0.p0 + I, I) : (void)0), ...);
        }(std::make_integer_sequence<int, D>{});
        seg_end(0, 0, unit);
#ifdef FLUTE_STAMPS
        __builtin_amdgcn_s_waitcnt(0);
        stamp[3] = wall_clock64();
        if (lane == 0 && a.splitk == 1 && a.partial != nullptr) {
            uint64_t* o = reinterpret_cast<uint64_t*>(a.partial) + ((size_t)blockIdx.x * W + wave) * 8;
            for (int i = 0; i < 8; ++i) o[i] = stamp[i];
        }
#endif
        return;
    }

    // ---- compute cursor: segments in order (idle and empty ones still take part in the barriers) ----
    int cv = 0, cc = 0;
    for (int cs = 0; cs < nseg; ++cs) {
        const Geo cg = (nchunks == 1) ? geo0 : chunk_geo(cc);
        const int unit = unit_of(cv);
        const int cnp = (unit < a.units) ? cg.np : 0;
        if (cs == 0 || nchunks > 1) {                              // (re)stage the activations of this chunk
            if (cs > 0) __syncthreads();                           // every wave is done with the previous chunk
            x_commit(cc, cs == 0);
            __syncthreads();                                       // table / activations visible to every wave
        }
        // scales of this segment: prefetched during the previous one with at least D ring slots issued behind
        // them; the first block is the YOUNGEST load of the prologue
        if (cs == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
#pragma unroll
        for (int r = 0; r < SPR; ++r) vm_wait_regs<(D - 1) * NP>(sv[r]);
        s_commit(unit);
        FLUTE_SSTAMP(1);
        {   // prefetch the next segment's scales behind the ring loads already in flight
            int nv = cv, nc = cc + 1;
            if (nc == nchunks) { nc = 0; ++nv; }
            if (nchunks > 1) { sg = sgeo_of(chunk_geo(nc)); s_roles(); }
            s_issue((cs + 1 < nseg) ? unit_of(nv) : a.units);
        }
        if (cc == 0) {
#pragma unroll
            for (int j = 0; j < J; ++j)
#pragma unroll
                for (int m = 0; m < MB; ++m) acc[j][m] = 0.f;
        }
        const int cpad = slots_of(cnp);
        auto slot_step = [&](auto slot_tag, int p0) {
            constexpr int i = decltype(slot_tag)::value;
            const int cp = p0 + i;
            // slot i is valid once at most the (D-1) younger slots are outstanding
            if constexpr (NP == 1) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(q[i][0]) : "n"((D - 1) * NP) : "memory");
            else asm volatile("s_waitcnt vmcnt(%3)" : "+v"(q[i][0]), "+v"(q[i][1]), "+v"(q[i][2]) : "n"((D - 1) * NP) : "memory");
            if (cp < cnp) compute_piece(slot_tag, cg.p0 + cp, cp);
            ring_issue(i);                                         // refill this slot D slots ahead
        };
        for (int p0 = 0; p0 < cpad; p0 += D) {
            [&]<int... I>(std::integer_sequence<int, I...>) {
                (slot_step(std::integral_constant<int, I>{}, p0), ...);
            }(std::make_integer_sequence<int, D>{});
            l_advance();
        }
        seg_end(cv, cc, unit);
        if (++cc == nchunks) { cc = 0; ++cv; }
    }
    // the last refills are zero-length reads still landing in q[]: drain before the wave ends
#pragma unroll
    for (int i = 0; i < D; ++i) {
        if constexpr (NP == 1) asm volatile("s_waitcnt vmcnt(0)" : "+v"(q[i][0]) : : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" : "+v"(q[i][0]), "+v"(q[i][1]), "+v"(q[i][2]) : : "memory");
    }
#pragma unroll
    for (int r = 0; r < SPR; ++r) vm_wait_regs<0>(sv[r]);
#ifdef FLUTE_STAMPS
    __builtin_amdgcn_s_waitcnt(0);
    stamp[3] = wall_clock64();
    if (lane == 0 && a.splitk == 1 && a.partial != nullptr) {
        uint64_t* o = reinterpret_cast<uint64_t*>(a.partial) + ((size_t)blockIdx.x * W + wave) * 8;
        for (int i = 0; i < 8; ++i) o[i] = stamp[i];
    }
#endif
#undef FLUTE_SSTAMP
}

}  // namespace flute_amd
