// Persistent form of the one-shot decode kernel (round 3) for layers whose one-shot launch would rebuild the table
// image many times per CU (8192 x 28672: 1792 workgroups, 7 per CU): the workgroup stages the table image and the
// activations ONCE and every wave then walks its own units, one segment (D pieces = one K chunk of one unit) at a
// time, the scale words and weight pieces of the next NS - 1 segments requested BEFORE the current segment is decoded
// (NS register sets; the loop body is unrolled NS times so that every set index is a compile-time constant).  A wave takes whole unit rows (no K split
// inside the workgroup): after the set-up barrier the waves never synchronise again - no arrival counters, no
// cross-wave reduction - and a unit's outputs are stored by the wave that decoded it.
// Same arithmetic, table / lookup scheme and wire format as qgemm_oneshot.h; decode loop = pipelined_pieces().
// Replaces, like the ring kernel (qgemm_stream.h), the reference's Stream-K main loop for M <= 2
// (flute/csrc/qgemm_kernel.hpp:617-712, tile_scheduler_utils.hpp:460-481).
// Host contract (api.hip: plan_persist): M <= MB in {1, 2} rows whose staged activations fit LDS beside
// the table image, K a whole number of chunks of D pieces (K % (512 D) == 0), G even,
// group size >= 64, at most 8 waves (two register sets of D pieces: the 256-register budget of two waves per SIMD).
#pragma once
#include "qgemm_oneshot.h"

namespace flute_amd {

struct PersistGeo {     // packed launch geometry (one kernel-argument dword, preloaded)
    static constexpr uint32_t pack(int lg, int waves, int nch, int ipw, int had_log, int xh, int M) {
        return (uint32_t)lg | ((uint32_t)waves << 4) | ((uint32_t)nch << 9) | ((uint32_t)ipw << 17) |
               ((uint32_t)had_log << 24) | ((uint32_t)xh << 28) | ((uint32_t)(M - 1) << 29);
    }
};
__host__ __device__ constexpr size_t persist_lds_bytes(int bits, int mb, int depth, int nsets, int lg, int K, int waves) {
    const int J = (bits == 3) ? 16 : 16 / bits;
    return (size_t)oneshot_lut_bytes(bits) + (oneshot_x_in_holes(bits, mb, K) ? 0 : (size_t)mb * ((K + 511) / 512 * 512) * 2) +
           (size_t)waves * nsets * J * depth * (512 >> lg) * 2;
}

template <typename T, int BITS, int TILEP, int MB, int D, int NS, bool HAD>
__global__ __launch_bounds__(512) void qgemv_persist_kernel(
    const uint32_t* __restrict__ Qp, const void* __restrict__ Sp, const void* __restrict__ Ap,
    const uint32_t* __restrict__ QM2, int K, int N, uint32_t geo, int nvis, void* __restrict__ Dp, float had_scale,
    int nwg) {
    using L = Layout<BITS>;
    using NT = Num<T>;
    constexpr int J = L::J;
    constexpr int NP = L::NPLANES;
    constexpr int LJ = (BITS == 4) ? 2 : (BITS == 2 ? 3 : 4);
    constexpr int NSL = oneshot_scale_loads(BITS, D);
    constexpr int LUT = oneshot_lut_bytes(BITS);
    constexpr int ESTRIDE = (BITS == 3) ? 128 : 256;
    constexpr int XPR = 4 / MB;                                    // activation pieces a thread stages from registers, per row
    constexpr int NXV = MB * XPR;
    constexpr int SEG = NSL + D * NP;                              // hidden loads of one segment
    constexpr int AHEAD = (NS - 1) * SEG;                          // loads of the segments requested ahead
    static_assert((D - 1) * NP + NSL + AHEAD <= 63 && NXV + AHEAD <= 63, "vmcnt is a 6-bit counter");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (lds_base_of(smem) != 0) __builtin_trap();

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lg = geo & 15, W = (geo >> 4) & 31, nch = (geo >> 9) & 255;
    const int ipw = (geo >> 17) & 127, had_log = (geo >> 24) & 15;
    const bool xh = (geo >> 28) & 1;
    const int M = (int)((geo >> 29) & 3) + 1;                      // rows (<= MB)
    const int nthr = W * 64;
    const int units = N >> LJ;
    const int G = K >> lg;
    const int KX = K;                                              // whole pieces (host)
    const uint32_t lane16 = (uint32_t)lane * 16u;
    const uint32_t row_bytes = (uint32_t)K * 2u;
    const int gpp = 512 >> lg;
    const int ngm = D * gpp;
    const uint32_t x_off = LUT;
    const uint32_t s_off = x_off + (xh ? 0u : (uint32_t)(MB * KX * 2));
    // byte address of the 16-B piece `pidx` (8 k each) of staged row m (as qgemm_oneshot.h)
    auto x_addr = [&](int m, int pidx) -> uint32_t {
        return xh ? (uint32_t)((m * (KX >> 6) + (pidx >> 3)) * 256 + 128 + (pidx & 7) * 16)
                  : x_off + (uint32_t)(m * KX * 2 + pidx * 16);
    };
    const uint32_t s_wave_bytes = (uint32_t)(J * ngm * 2);
    const uint32_t sbase = s_off + (uint32_t)(wave * NS) * s_wave_bytes;   // one image per register set

    // ---- set-up requests: table word, activations ----
    const srd_t lut_srd = make_srd(QM2, (uint32_t)(4 << (2 * BITS)));
    const int run0 = wave * ipw;
    uint32_t lut_v;
    if constexpr (BITS == 2) lut_v = buf_load4((uint32_t)(lane & 15) * 4u, lut_srd);
    else lut_v = buf_load4((uint32_t)(run0 * 8 + lane) * 4u, lut_srd);
    const srd_t x_srd = make_srd(Ap, (uint32_t)min((size_t)M * K * 2, (size_t)0xfffffff0u));
    const int xrow_pieces = KX >> 3;
    ring16_t xv[MB][XPR];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int r = 0; r < XPR; ++r) {
            const int pidx = r * nthr + tid;
            xv[m][r] = buf_load16(pidx < xrow_pieces ? (uint32_t)(((size_t)min(m, M - 1) * K + (size_t)pidx * 8) * 2) : 0x80000000u, x_srd, 0);
        }

    // ---- segments: (visit v, chunk c) = pieces c D .. c D + D - 1 of unit (blockIdx + v nwg) W + wave ----
    const srd_t s_srd = make_srd(Sp, (uint32_t)min((size_t)N * G * 2, (size_t)0xfffffff0u));
    const int lgh = 31 - __builtin_clz((unsigned)ngm) - 1;          // log2(ngm / 2)
    ring16_t q[NS][D][NP];
    uint32_t sv[NS][NSL];
    auto unit_of = [&](int v) { return ((int)blockIdx.x + v * nwg) * W + wave; };
    auto issue_seg = [&](auto set_tag, int v, int c) {             // unconditional: dead segments read nothing (zero-length descriptors)
        constexpr int set = decltype(set_tag)::value;
        const int unit = unit_of(v);
        const bool live = v < nvis && unit < units;
        const int urow = min(max(unit, 0), units - 1);
        const int col0 = unit_col0<BITS, TILEP>(urow);
        const int g0 = c * ngm;
#pragma unroll
        for (int r = 0; r < NSL; ++r) {
            const int qq = lane + 64 * r;
            const int j = qq >> lgh;
            const int gp = qq & ((1 << lgh) - 1);
            const bool mine = live && j < J;
            sv[set][r] = buf_load4_at(mine ? (uint32_t)(((size_t)(col0 + j * TILEP) * G + g0 + 2 * gp) * 2) : 0x80000000u, s_srd, 0);
        }
        srd_t qsrd[NP];
#pragma unroll
        for (int pl = 0; pl < NP; ++pl)
            qsrd[pl] = make_srd(reinterpret_cast<const char*>(Qp) + (size_t)unit_row<BITS, TILEP>(urow, pl, N) * row_bytes,
                                live ? row_bytes : 0u);
#pragma unroll
        for (int i = 0; i < D; ++i) {
            const uint32_t vo = lane16 + (uint32_t)(c * D + i) * 1024u;
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) q[set][i][pl] = buf_load16_nt(vo, qsrd[pl], 0);
        }
    };
    int vi = 0, ci = 0;                                            // the request front
    auto advance = [&](int& v, int& c) { if (++c == nch) { c = 0; ++v; } };
    __builtin_amdgcn_sched_barrier(0);
    static_for<NS - 1>([&](auto set_tag) { issue_seg(set_tag, vi, ci); advance(vi, ci); });
    __builtin_amdgcn_sched_barrier(0);

    // ---- table image (as qgemm_oneshot.h) ----
    vm_wait_regs<NXV + AHEAD>(lut_v);
    {
        constexpr int RUNS = oneshot_lut_runs(BITS);
        const int nrun = max(0, min(ipw, RUNS - run0));
        for (int i0 = 0; i0 < nrun; i0 += 4) {
            uint32_t tlo[4], thi[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if constexpr (BITS == 2) {
                    const int e = (run0 + i0 + u) * 4 + (lane >> 4);
                    tlo[u] = (uint32_t)__builtin_amdgcn_ds_bpermute((e & 15) * 4, (int)lut_v);
                    thi[u] = (uint32_t)__builtin_amdgcn_ds_bpermute(((e >> 4) & 15) * 4, (int)lut_v);
                } else {
                    tlo[u] = (uint32_t)__builtin_amdgcn_ds_bpermute((((i0 + u) * 8 + (lane >> 3)) & 63) * 4, (int)lut_v);
                    thi[u] = tlo[u];
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (i0 + u < nrun) {
                    uint32_t addr;
                    if constexpr (BITS == 2) addr = (uint32_t)(run0 + i0 + u) * 1024u + lane16;
                    else addr = (uint32_t)((run0 + i0 + u) * 8 + (lane >> 3)) * ESTRIDE + (uint32_t)(lane & 7) * 16u;
                    *reinterpret_cast<uint4*>(smem + addr) = make_uint4(tlo[u], thi[u], tlo[u], thi[u]);
                }
            }
        }
    }
    // ---- activations -> LDS (fused pre-rotation: a wave's 64 pieces are 512 consecutive k of one row) ----
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int r = 0; r < XPR; ++r) vm_wait_regs<AHEAD>(xv[m][r]);
#pragma unroll
    for (int m = 0; m < MB; ++m) {
#pragma unroll
        for (int r = 0; r < XPR; ++r) {
            const int pidx = r * nthr + tid;
            if (pidx < xrow_pieces) {
                uint32_t w[4] = {xv[m][r].x, xv[m][r].y, xv[m][r].z, xv[m][r].w};
                if constexpr (HAD) fwht_piece<T>(w, lane, had_log, had_scale);
                *reinterpret_cast<uint4*>(smem + x_addr(m, pidx)) = make_uint4(w[0], w[1], w[2], w[3]);
            }
        }
        const uint16_t* A = reinterpret_cast<const uint16_t*>(Ap);   // rows longer than XPR x threads pieces
        for (int pidx = XPR * nthr + tid; pidx < xrow_pieces; pidx += nthr) {
            const uint4 t = *reinterpret_cast<const uint4*>(A + (size_t)min(m, M - 1) * K + (size_t)pidx * 8);
            uint32_t w[4] = {t.x, t.y, t.z, t.w};
            if constexpr (HAD) fwht_piece<T>(w, lane, had_log, had_scale);
            *reinterpret_cast<uint4*>(smem + x_addr(m, pidx)) = make_uint4(w[0], w[1], w[2], w[3]);
        }
    }
    __syncthreads();                                               // the only barrier: table image and activations are in LDS

    const uint32_t lane_off = (BITS == 2) ? (uint32_t)(lane & 31) * 8u : (uint32_t)(lane & 31) * 4u;
    const int gl = (8 * lane) >> lg;
    const uint32_t x_pshift = xh ? 11u : 10u;
    const uint32_t x_lane0 = x_addr(0, lane);
    const uint32_t x_row = xh ? (uint32_t)(KX >> 6) * 256u : (uint32_t)KX * 2u;
    const uint32_t s_piece = (uint32_t)(gpp * J) * 2u;

    float acc[J][MB];
    uint16_t* Dout = reinterpret_cast<uint16_t*>(Dp);
    // one segment from register set `set`: its scale image, its D pieces, and - on a unit's last chunk - the outputs
    auto do_seg = [&](auto set_tag, int v_idx, int c) {
        const int v = v_idx;
        constexpr int set = decltype(set_tag)::value;
        const uint32_t simg = sbase + (uint32_t)set * s_wave_bytes;
#pragma unroll
        for (int r = 0; r < NSL; ++r) {
            vm_wait_regs<D * NP + AHEAD>(sv[set][r]);              // younger: this segment's pieces and the segments ahead
            const int qq = lane + 64 * r;
            const int j = qq >> lgh;
            const int gp = qq & ((1 << lgh) - 1);
            if (j < J) {
                const uint32_t w = sv[set][r];
                uint16_t* img = reinterpret_cast<uint16_t*>(smem + simg) + (size_t)(2 * gp) * J + j;
                img[0] = (uint16_t)(w & 0xffffu);
                img[J] = (uint16_t)(w >> 16);
            }
        }
        if (c == 0) {
#pragma unroll
            for (int j = 0; j < J; ++j)
#pragma unroll
                for (int m = 0; m < MB; ++m) acc[j][m] = 0.f;
        }
        pipelined_pieces<T, BITS, MB, D, AHEAD, (BITS != 2) ? 1 : 0>(q[set], x_lane0 + ((uint32_t)(c * D) << x_pshift), x_pshift, x_row,
                                                                  simg + (uint32_t)(gl * J) * 2u, s_piece, lane_off, acc);
        if constexpr (BITS != 2) {
            // round 5: the unit's J x MB sums reduced together, one store instruction (qgemm_oneshot.h: transpose_reduce4 / 16)
            if (c == nch - 1) {
                float v;
                int my_m;
                bool holder;
                if constexpr (BITS == 4) transpose_reduce4<MB>(acc, lane, v, my_m, holder);
                else transpose_reduce16<MB>(acc, lane, v, my_m, holder);
                const int unit = unit_of(v_idx);
                if (holder && my_m < M && v_idx < nvis && unit < units)
                    Dout[(size_t)my_m * N + unit_col0<BITS, TILEP>(unit) + (lane & (J - 1)) * TILEP] = NT::from_float(v);
            }
        } else if (c == nch - 1) {
            float tot[J][MB];
#pragma unroll
            for (int j = 0; j < J; ++j)
#pragma unroll
                for (int m = 0; m < MB; ++m) tot[j][m] = wave_sum64(acc[j][m]);
            const int unit = unit_of(v);
            if (lane == 0 && v < nvis && unit < units) {
                const int col0 = unit_col0<BITS, TILEP>(unit);
#pragma unroll
                for (int m = 0; m < MB; ++m)
                    if (m < M) {
#pragma unroll
                        for (int j = 0; j < J; ++j) Dout[(size_t)m * N + col0 + j * TILEP] = NT::from_float(tot[j][m]);
                    }
            }
        }
    };

    // ---- the walk: segment s from set s % NS, segment s + NS - 1 requested first (past the end: dead requests) ----
    int vd = 0, cd = 0;
    const int nseg = nvis * nch;
    for (int s0 = 0; s0 < nseg; s0 += NS) {
        static_for<NS>([&](auto k_tag) {
            constexpr int k = decltype(k_tag)::value;
            issue_seg(std::integral_constant<int, (k + NS - 1) % NS>{}, vi, ci);
            advance(vi, ci);
            if (s0 + k < nseg) do_seg(k_tag, vd, cd);
            advance(vd, cd);
        });
    }
    // what is still in flight is dead (zero-length) or unused: drain before the wave ends
#pragma unroll
    for (int set = 0; set < NS; ++set) {
#pragma unroll
        for (int i = 0; i < D; ++i) {
            if constexpr (NP == 1) asm volatile("s_waitcnt vmcnt(0)" : "+v"(q[set][i][0]) : : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" : "+v"(q[set][i][0]), "+v"(q[set][i][1]), "+v"(q[set][i][2]) : : "memory");
        }
#pragma unroll
        for (int r = 0; r < NSL; ++r) vm_wait_regs<0>(sv[set][r]);
    }
}

}  // namespace flute_amd
