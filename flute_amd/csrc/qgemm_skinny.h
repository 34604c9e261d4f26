// MFMA kernel for 4-bit layers at M <= 16 whose 64-column slabs fill the chip in one round (round 3): weights and
// activations go STRAIGHT to the registers of the lanes that feed them to the matrix core - no LDS staging of operands.
// Replaces, for these M, the per-wave kernel of qgemm_tile.h, whose operands travel through wave-private LDS rings (one
// LDS-DMA per MFMA operand).  Reference: qgemm_device's main loop (flute/csrc/qgemm_kernel.hpp:617-712); its Stream-K
// fix-up of partial tiles (tile_scheduler_utils.hpp:460-481) is the in-workgroup K reduction below.
//
// Geometry.  v_mfma_f32_16x16x32: A = weights (lane (u, q): row u = l % 16, k = 8 q .. 8 q + 7), B = activations (lane
// (j, q): row j of X, the same k), D[u][j] in lane (q, j) as rows 4 q .. 4 q + 3.  A wave owns a SLAB of 16 units (lane
// l: unit l % 16) and D k-steps (32 k each) of it: per k-step lane (u, q) loads the 16 B of unit row u that hold k-pairs
// 4 q .. 4 q + 3 of that step - 16 rows x 64 B per request; its four dwords are the four k-pairs, byte t of a dword the
// pair code of column tile t, so the four lookups of a dword feed four MFMAs (four column tiles of 16 units) against the
// SAME activation operand.  Activations: lane (j, q) loads its 16 B of X the same way (rows >= M: zero).
// Per wave: 1 table + 1 scale + D weight + D activation requests, then D x (16 lookups, 16 scale multiplications,
// 4 MFMAs).  (Written for MT row tiles; one is instantiated: two were no faster than the per-wave kernel.)
// Requests: the first two k-steps before the table image is built, the rest of the first half after the barrier,
// k-step i + D / 2 when k-step i is decoded - see the comments at issue_step.
// K is split over the KW (4 or 8) waves of a workgroup only (LDS reduction).  A grid-level split was built and measured
// (profiles/r03/skinny_lab_gridsplit.jsonl): partial tiles in the workspace and a last-arriver ticket need agent-scope
// fences, i.e. an L2 write-back per workgroup on this multi-die part - 51 us at 4 splits on 4096 x 4096 against 14.5 us
// unsplit; a second launch costs ~2 us.  So the kernel serves layers with enough slabs for the chip (N >= ~9 K columns)
// and K = 32 D KW <= 4096; the rest stays on the per-wave kernel.
// Where the time goes (4096 x 14336, M = 16, 224 workgroups; builds with parts removed, profiles/r03/
// skinny_ablation.txt, all requests up front: 14.5 us): launch + table image + reduction 4.2 us, the requests of a
// workgroup (131 KB of weights + 128 KB of X through ONE CU) 4.1 us, the k-steps 3.9 us (as many lookups per CU as the
// M = 1 kernels).  Stamps (profiles/r03/skinny_stamps*.json): the 34 requests of a wave take 9 K cycles to issue, and the
// second wave of a SIMD starts when the first is done - a request that touches 16 half cache lines costs ~60 cycles of the
// CU's addresser (~270 of the SIMD's memory issue); asking for the same bytes line-coalesced (lane l: chunk l % 4 of row
// l / 4) and moving them to the operand lanes with ds_bpermute was slower (16.7 us: the cost is per line touched, and the
// k-steps grow from 520 to 850 cycles).  With the requests interleaved as above: 12.6 us.
// Arithmetic: fp16 w^ = round_T(lut * s) (packbits_utils.hpp:139), fp32 accumulation in the matrix core; bf16 (no packed
// multiply on gfx950) applies the group scale to the fp32 MFMA result of each group run, as qgemm_tile.h.
// Host contract (api.hip: plan_skinny): 4 bits, M <= 16, K = 32 D KW with D in {4, 8, 16} and KW in {4, 8}, G even, group
// size >= 32 and D * 32 / g <= 8, N % 64 == 0 (always: TileP >= 32).
#pragma once
#include "qgemm_oneshot.h"
#include "mfma.h"
#include "xwg.h"

namespace flute_amd {

struct SkinnyGeo {
    static constexpr uint32_t pack(int lg, int lkw, int ipw, int splitk = 1) { return (uint32_t)lg | ((uint32_t)lkw << 4) | ((uint32_t)splitk << 8) | ((uint32_t)ipw << 20); }
};
__host__ __device__ constexpr size_t skinny_lds_bytes(int bits, int mt, int kw) {
    const int J = 16 / bits;
    // > 80 KB on purpose: one workgroup per CU (two resident workgroups share a CU's memory path and were measured
    // slower - 4096 x 14336 M = 16: 15.7 against 14.5 us - when the grid has fewer workgroups than the chip has CUs)
    return (size_t)oneshot_lut_bytes(bits) + (size_t)kw * J * 256 + (size_t)kw * J * mt * 1024;
}

template <typename T, int BITS, int TILEP, int MT, int D, int MAXW>
__global__ __launch_bounds__(MAXW * 64) void qgemm_skinny_kernel(
    const uint32_t* __restrict__ Qp, const void* __restrict__ Sp, const void* __restrict__ Ap,
    const uint32_t* __restrict__ QM2, int K, int N, uint32_t geo, int M, void* __restrict__ Dp, uint64_t* __restrict__ stamps,
    float* __restrict__ partial, uint32_t* __restrict__ state) {
    static_assert(BITS == 4, "2- and 3-bit layers take the per-wave kernel");
    using NT = Num<T>;
    constexpr int J = 16 / BITS;                                   // column tiles per k-step
    constexpr int LJ = (BITS == 4) ? 2 : 3;
    constexpr int NI = J * MT;                                     // 16 x 16 output tiles per wave
    constexpr int NSL = J / 4;                                     // scale requests per lane
    constexpr int LUT = oneshot_lut_bytes(BITS);
    constexpr int NX = MT * D;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (lds_base_of(smem) != 0) __builtin_trap();                  // v_perm-built table addresses are absolute
#ifdef FLUTE_STAMPS
    uint64_t stamp[16];
    for (int i = 0; i < 16; ++i) stamp[i] = 0;
    stamp[0] = wall_clock64();                                    // 100 MHz, chip-wide
    stamp[1] = __builtin_amdgcn_s_memtime();                       // shader cycles: the phases of this wave
#define FLUTE_KSTAMP(i) stamp[i] = __builtin_amdgcn_s_memtime()
#else
#define FLUTE_KSTAMP(i)
#endif

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lg = geo & 15, lkw = (geo >> 4) & 15, ipw = (geo >> 20) & 127;
    const int KW = 1 << lkw;
    const int u = lane & 15, q = lane >> 4;
    const int units = N >> LJ;
    const int slabs = units >> 4;
    // grid-level K split (round 4): workgroup = (slab, K slice), the slices of a slab are neighbours in the block order and
    // meet through the workspace (xwg.h, L form: 4 KB per slice at M = 16); nsplit = 1: the whole of K, as before
    const int nsplit = (geo >> 8) & 255;
    const int slab = nsplit > 1 ? (int)blockIdx.x / nsplit : (int)blockIdx.x;
    const int split = nsplit > 1 ? (int)blockIdx.x % nsplit : 0;
    const int G = K >> lg;
    const int ksteps = (K >> 5) / nsplit;                          // k-steps of this workgroup's slice (K % (32 nsplit) == 0)
    const int kp = split * (ksteps / D) + wave;                    // this wave's K part: k-steps [kp D, kp D + D) of the row (ksteps % D == 0)
    const bool active = wave * D < ksteps;                         // a part is whole or empty
    const uint32_t row_bytes = (uint32_t)K * 2u;
    const uint32_t kbyte0 = (uint32_t)(kp * D) * 64u + (uint32_t)q * 16u;      // this lane's 16 B of k-step 0 inside a row
    const uint32_t dead = 0x80000000u;

    // LDS carve (skinny_lds_bytes)
    const uint32_t s_off = LUT;
    const uint32_t simg = s_off + (uint32_t)wave * (J * 256);
    const uint32_t red_off = s_off + (uint32_t)KW * (J * 256);

    // ---- requests, oldest first: table word, activations, scale words, weights ----
    const srd_t lut_srd = make_srd(QM2, (uint32_t)(4 << (2 * BITS)));
    const int run0 = wave * ipw;
    uint32_t lut_v;
    if constexpr (BITS == 2) lut_v = buf_load4((uint32_t)(lane & 15) * 4u, lut_srd);
    else lut_v = buf_load4((uint32_t)(run0 * 8 + lane) * 4u, lut_srd);

    // scale words: lane (u, q) fetches 8 groups (16 B) of column tile t = q (+ 4) of its unit, from the even group at or
    // below the wave's first one (G even: the address is dword-aligned; a misaligned buffer load reads aligned-down)
    const int unit = slab * 16 + u;
    const int col0 = unit_col0<BITS, TILEP>(unit);
    const int g0 = (kp * D * 32) >> lg;
    const int g0e = g0 & ~1;
    const srd_t s_srd = make_srd(Sp, (uint32_t)min((size_t)N * G * 2, (size_t)0xfffffff0u));
    ring16_t sv[NSL];
#pragma unroll
    for (int r = 0; r < NSL; ++r) {
        const int t = q + 4 * r;
        sv[r] = buf_load16(active ? (uint32_t)(((size_t)(col0 + t * TILEP) * G + g0e) * 2) : dead, s_srd, 0);
    }

    // k-step i: its weight piece, then its activation pieces (the loads of a wave return in order: step i is released
    // by ONE counted wait while the later steps are still in flight)
    const srd_t x_srd = make_srd(Ap, (uint32_t)min((size_t)M * K * 2, (size_t)0xfffffff0u));
    const srd_t q_srd = make_srd(Qp, (uint32_t)min((size_t)units * row_bytes, (size_t)0xfffffff0u));
    ring16_t w[D];
    ring16_t xv[MT][D];
    // The requests of the first PRE k-steps go out before the table image is built (their HBM latency hides it), the
    // rest after the barrier: a request that touches 16 half cache lines keeps the SIMD's memory issue busy for ~270
    // cycles (stamps), so the second wave of a SIMD issues long after the first - which meanwhile decodes.
    constexpr int AHEAD = D / 2;
    constexpr int PRE = AHEAD < 2 ? AHEAD : 2;
    const uint32_t wbase = active ? (uint32_t)unit * row_bytes + kbyte0 : dead;
    uint32_t xbase[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) xbase[mt] = (active && mt * 16 + u < M) ? (uint32_t)(mt * 16 + u) * row_bytes + kbyte0 : dead;
    auto issue_step = [&](auto i_tag) {
        constexpr int i = decltype(i_tag)::value;
        w[i] = buf_load16(wbase + (uint32_t)i * 64u, q_srd, 0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) xv[mt][i] = buf_load16(xbase[mt] + (uint32_t)i * 64u, x_srd, 0);
    };
    static_for<PRE>(issue_step);
    __builtin_amdgcn_sched_barrier(0);
    FLUTE_KSTAMP(2);

    // ---- table image (as qgemm_oneshot.h) ----
    vm_wait_regs<NSL + PRE * (1 + MT)>(lut_v);
    {
        constexpr int RUNS = oneshot_lut_runs(BITS);
        constexpr int ESTRIDE = 256;
        const uint32_t lane16 = (uint32_t)lane * 16u;
        const int nrun = max(0, min(ipw, RUNS - run0));
        for (int i0 = 0; i0 < nrun; i0 += 4) {
            uint32_t tlo[4], thi[4];
#pragma unroll
            for (int uu = 0; uu < 4; ++uu) {
                if constexpr (BITS == 2) {
                    const int e = (run0 + i0 + uu) * 4 + (lane >> 4);
                    tlo[uu] = (uint32_t)__builtin_amdgcn_ds_bpermute((e & 15) * 4, (int)lut_v);
                    thi[uu] = (uint32_t)__builtin_amdgcn_ds_bpermute(((e >> 4) & 15) * 4, (int)lut_v);
                } else {
                    tlo[uu] = (uint32_t)__builtin_amdgcn_ds_bpermute((((i0 + uu) * 8 + (lane >> 3)) & 63) * 4, (int)lut_v);
                    thi[uu] = tlo[uu];
                }
            }
#pragma unroll
            for (int uu = 0; uu < 4; ++uu) {
                if (i0 + uu < nrun) {
                    uint32_t addr;
                    if constexpr (BITS == 2) addr = (uint32_t)(run0 + i0 + uu) * 1024u + lane16;
                    else addr = (uint32_t)((run0 + i0 + uu) * 8 + (lane >> 3)) * ESTRIDE + (uint32_t)(lane & 7) * 16u;
                    *reinterpret_cast<uint4*>(smem + addr) = make_uint4(tlo[uu], thi[uu], tlo[uu], thi[uu]);
                }
            }
        }
    }
    // ---- scale image of the wave: [tile t][unit u] x 8 groups (16 B) ----
#pragma unroll
    for (int r = 0; r < NSL; ++r) {
        vm_wait_regs<PRE * (1 + MT)>(sv[r]);
        *reinterpret_cast<uint4*>(smem + simg + (uint32_t)(((q + 4 * r) * 16 + u) * 16)) = make_uint4(sv[r].x, sv[r].y, sv[r].z, sv[r].w);
    }
    FLUTE_KSTAMP(3);
    __syncthreads();                                               // table image visible to every wave
    // the first half of the k-steps now, k-step i + AHEAD when k-step i is decoded: the two waves of a SIMD then
    // alternate between requesting and decoding (all requests up front: the second wave of a SIMD only starts to request
    // when the first is done - 25 K cycles per workgroup against 20 K: profiles/r03/skinny_stamps*.json)
    static_for<AHEAD>([&](auto i_tag) { if constexpr (decltype(i_tag)::value >= PRE) issue_step(i_tag); });
    __builtin_amdgcn_sched_barrier(0);
    FLUTE_KSTAMP(4);

    const uint32_t lane_off = (BITS == 2) ? (uint32_t)(lane & 31) * 8u : (uint32_t)(lane & 31) * 4u;
    f32x4_t acc[NI];
#pragma unroll
    for (int e = 0; e < NI; ++e) acc[e] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    if (active) {
        // Software-pipelined k-steps (as pipelined_pieces of the decode kernels).  A k-step is two groups of 8 lookups
        // (column tiles 0, 1 | 2, 3 x the lane's four k-pairs); the lookups of group g + 1 - and, on a k-step boundary,
        // its J scale reads (hidden) behind the counted wait for its weight / activation pieces - are issued before
        // the scaling and the MFMAs of group g; LDS returns in order, so "at most NEXT_READS younger operations
        // outstanding" releases group g.
        static_assert(BITS == 4 && J == 4, "the skinny kernel is instantiated for 4-bit layers");
        constexpr int NG = 2;
        constexpr bool PRE = __is_same(T, F16);                    // fp16: w^ = round_T(lut s) by v_pk_mul_f16 before the MFMA
        uint32_t v[2][8];
        uint32_t sc[2][J];
        f32x4_t run[NI];
#pragma unroll
        for (int e = 0; e < NI; ++e) run[e] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        auto issue_group = [&](auto g_tag) {
            constexpr int g = decltype(g_tag)::value;
            constexpr int I = g / NG, GI = g % NG;
            if constexpr (GI == 0) {
                if constexpr (I + AHEAD < D) issue_step(std::integral_constant<int, I + AHEAD>{});
                constexpr int YOUNGER = ((I + AHEAD < D ? I + AHEAD : D - 1) - I) * (1 + MT);      // k-steps requested behind this one
                if constexpr (MT == 1) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(w[I]), "+v"(xv[0][I]) : "n"(YOUNGER) : "memory");
                else asm volatile("s_waitcnt vmcnt(%3)" : "+v"(w[I]), "+v"(xv[0][I]), "+v"(xv[1][I]) : "n"(YOUNGER) : "memory");
                if constexpr (PRE) {
                const uint32_t sa = simg + (uint32_t)(u * 16) + (uint32_t)((((kp * D + I) * 32) >> lg) - g0e) * 2u;
#pragma unroll
                for (int t = 0; t < J; ++t) asm volatile("ds_read_u16 %0, %1 offset:%2" : "=v"(sc[I & 1][t]) : "v"(sa), "n"(t * 256) : "memory");
                }
            }
            const uint32_t wd[4] = {w[I].x, w[I].y, w[I].z, w[I].w};
#pragma unroll
            for (int ww = 0; ww < 4; ++ww)
#pragma unroll
                for (int tt = 0; tt < 2; ++tt)
                    v[g & 1][ww * 2 + tt] = lds_lookup32(__builtin_amdgcn_perm(wd[ww], lane_off, 0x0c0c0400u | ((4u + 2 * GI + tt) << 8)));
        };
        issue_group(std::integral_constant<int, 0>{});
        static_for<NG * D>([&](auto g_tag) {
            constexpr int g = decltype(g_tag)::value;
            constexpr int I = g / NG, GI = g % NG;
            if constexpr (g + 1 < NG * D) issue_group(std::integral_constant<int, g + 1>{});
            constexpr int NEXT_READS = (g + 1 < NG * D) ? 8 + ((PRE && (g + 1) % NG == 0) ? J : 0) : 0;
            uint32_t(&vv)[8] = v[g & 1];
            asm volatile("s_waitcnt lgkmcnt(%8)"
                         : "+v"(vv[0]), "+v"(vv[1]), "+v"(vv[2]), "+v"(vv[3]), "+v"(vv[4]), "+v"(vv[5]), "+v"(vv[6]), "+v"(vv[7])
                         : "n"(NEXT_READS) : "memory");
            if constexpr (GI == 0) {                               // the k-step's scale reads are older than its first group
#pragma unroll
                for (int t = 0; t < J; ++t) { uint32_t& r = sc[I & 1][t]; if constexpr (PRE) asm volatile("" : "+v"(r) : : "memory"); }
                if constexpr (I == 0) { FLUTE_KSTAMP(5); }
                if constexpr (I == 1) { FLUTE_KSTAMP(6); }
                if constexpr (I == D / 2) { FLUTE_KSTAMP(7); }
                if constexpr (I == D - 1) { FLUTE_KSTAMP(8); }
            }
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                const int t = 2 * GI + tt;
                u32x4_t a;
#pragma unroll
                for (int ww = 0; ww < 4; ++ww) a[ww] = PRE ? NT::mul_scale(v[g & 1][ww * 2 + tt], sc[I & 1][t]) : v[g & 1][ww * 2 + tt];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const u32x4_t b = {xv[mt][I].x, xv[mt][I].y, xv[mt][I].z, xv[mt][I].w};
                    if constexpr (PRE) acc[t * MT + mt] = Mfma<T>::run(a, b, acc[t * MT + mt]);
                    else run[t * MT + mt] = Mfma<T>::run(a, b, run[t * MT + mt]);
                }
            }
            if constexpr (!PRE && GI == NG - 1) {
                // bf16 (no packed multiply): the fp32 sums of a group run are scaled when the group (or the wave's range)
                // ends; the accumulator's four rows are units 4 q .. 4 q + 3 of the slab
                const int kabs = kp * D + I;
                if ((((kabs + 1) * 32) & ((1 << lg) - 1)) == 0 || I == D - 1) {
                    const uint32_t sa = simg + (uint32_t)(4 * q * 16) + (uint32_t)(((kabs * 32) >> lg) - g0e) * 2u;
#pragma unroll
                    for (int t = 0; t < J; ++t) {
                        float sf[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) sf[r] = NT::to_float((uint16_t)lds_ld16(sa + (uint32_t)(t * 256 + r * 16)));
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) acc[t * MT + mt][r] = __builtin_fmaf(run[t * MT + mt][r], sf[r], acc[t * MT + mt][r]);
                            run[t * MT + mt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                        }
                    }
                }
            }
        });
    } else {
        // nothing of this wave's is in range: its requests read nothing; drain them before the registers die
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }

    FLUTE_KSTAMP(9);
    // ---- K reduction inside the workgroup: tile e is summed by wave e % KW, in wave order ----
    float4* red = reinterpret_cast<float4*>(smem + red_off);
#pragma unroll
    for (int e = 0; e < NI; ++e) red[(wave * NI + e) * 64 + lane] = make_float4(acc[e][0], acc[e][1], acc[e][2], acc[e][3]);
    __syncthreads();
    uint16_t* Dout = reinterpret_cast<uint16_t*>(Dp);
    // tile e = (t, mt): lane (q, j) holds rows (units) 4 q .. 4 q + 3 of X row mt 16 + j: four consecutive columns
    auto store_tile = [&](int e, float4 s) {
        const int t = e / MT, mt = e % MT;
        const int row = mt * 16 + u;
        if (row < M) {
            const int col = unit_col0<BITS, TILEP>(slab * 16 + 4 * q) + t * TILEP;
            ushort4 o;
            o.x = NT::from_float(s.x); o.y = NT::from_float(s.y); o.z = NT::from_float(s.z); o.w = NT::from_float(s.w);
            *reinterpret_cast<ushort4*>(Dout + (size_t)row * N + col) = o;
        }
    };
#ifdef FLUTE_SK_ABLATE
    constexpr bool no_seam = (FLUTE_SK_ABLATE & 128) != 0;         // development builds: every slice stores its partial as the result
#else
    constexpr bool no_seam = false;
#endif
    if (nsplit == 1 || no_seam) {
        for (int e = wave; e < NI; e += KW) {
            float4 s = red[e * 64 + lane];
            for (int ww = 1; ww < KW; ++ww) {
                const float4 p = red[(ww * NI + e) * 64 + lane];
                s.x += p.x; s.y += p.y; s.z += p.z; s.w += p.w;
            }
            store_tile(e, s);
        }
    } else {
        // the slice's partial tiles go to its fp32 slab write-through; the workgroup whose arrival completes the slab's count
        // sums ALL slices in ascending order (its own from the slab too: the result does not depend on who is last)
        // slabs in fragment order: [slice][slab][tile e][lane] x 16 B (one contiguous KB per store / load instruction)
        const uint32_t slice_bytes = (uint32_t)slabs * (NI * 1024u);
        const __amdgpu_buffer_rsrc_t slabr = xwg_rsrc(partial, (uint32_t)nsplit * slice_bytes);
        auto tile_off = [&](int e) { return ((uint32_t)slab * NI + (uint32_t)e) * 1024u + (uint32_t)lane * 16u; };
        for (int e = wave; e < NI; e += KW) {
            float4 s = red[e * 64 + lane];
            for (int ww = 1; ww < KW; ++ww) {
                const float4 p = red[(ww * NI + e) * 64 + lane];
                s.x += p.x; s.y += p.y; s.z += p.z; s.w += p.w;
            }
            xwg_store(f32x4_t{s.x, s.y, s.z, s.w}, slabr, (uint32_t)split * slice_bytes + tile_off(e));
        }
        xwg_word* st = xwg_state(state + 2 * slab);
        const uint32_t before = xwg_arrive(st, (uint32_t)red_off, tid);
        if (before == (uint32_t)(nsplit - 1)) {
            for (int e = wave; e < NI; e += KW) {
                const uint32_t off = tile_off(e);
                f32x4_t acc4 = f32x4_t{0.f, 0.f, 0.f, 0.f};
                for (int s0 = 0; s0 < nsplit; s0 += 4) {           // four slices in flight
                    f32x4_t ld[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) ld[j] = xwg_load(slabr, s0 + j < nsplit ? (uint32_t)(s0 + j) * slice_bytes + off : 0xfffffff0u);
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc4 += ld[j];        // past the last slice: out of range reads as zero
                }
                store_tile(e, make_float4(acc4[0], acc4[1], acc4[2], acc4[3]));
            }
            xwg_reset(st, tid);
        }
    }
#ifdef FLUTE_STAMPS
    FLUTE_KSTAMP(10);
    __builtin_amdgcn_s_waitcnt(0);
    stamp[11] = __builtin_amdgcn_s_memtime();
    stamp[12] = wall_clock64();
    if (lane == 0 && stamps != nullptr) {
        uint64_t* o = stamps + ((size_t)blockIdx.x * KW + wave) * 16;
        for (int i = 0; i < 16; ++i) o[i] = stamp[i];
    }
#endif
#undef FLUTE_KSTAMP
}

}  // namespace flute_amd
