// Lean MFMA decode kernel (round 5): 5 <= M <= 16 rows of a 4-bit layer with K = 2048 / 4096 - the batch sizes between the
// dot-product decode kernels (M <= 4) and the tiles of the MFMA kernels, BASELINE.json's M = 16 on 4096 x 4096 among them.
//
// The per-wave MFMA kernel (qgemm_tile.h) and the skinny kernel (qgemm_skinny.h) give a wave 16 unit rows (64 columns): a
// 4096-wide layer has 64 such slabs, so the chip is filled by lane sharing (R = 4) or by a grid K split whose seam costs
// 1.5 - 1.7 us (xwg.h) - 7.1 us at M = 16 on 4096^2 where M = 1 takes 3.9.  Here a WORKGROUP owns FOUR unit rows (16
// columns = the N of one v_mfma_f32_16x16x32) for the whole of K - 256 workgroups on 4096^2, no seam - and its W waves
// split K inside the workgroup:
//   * MFMA row i = 4 u + j is column j (byte j of every packed word) of unit u; lane (i, kg) wants the 8 k of its column
//     that the k-chunk kg of a 32-k step holds = 4 words of unit u, of which it uses ONE byte each.  The four lanes of a
//     quad (same u and kg, j = 0..3) therefore share their requests: per 128-k macro-step lane j requests the 16 B of step
//     s = j (a wave-wide request covers 4 units x 256 contiguous bytes: whole cache lines), and in step s the quad's lane s
//     hands its four words to the others by DPP quad_perm broadcast.  Lookup address = (byte << 7) | copy: v_perm with a
//     per-lane selector (the byte position is the lane's j) puts byte j above twice the copy offset, one shift halves it
//     (table image at a 128-B entry stride, 32 KB: the activations need the rest of the LDS);
//   * the activations are the MFMA's other operand: lane (m, kg) reads the 16 B x[m][32 s + 8 kg ..] of its wave's K range
//     from LDS.  A wave stages exactly the rows x chunks IT multiplies (16 rows x its K range, row-contiguous requests:
//     whole lines) into a region only it reads - no barrier for the activations; the layout [chunk][slot = m ^ (chunk % 8)]
//     makes both the staging ds_write_b128 (eight consecutive chunks of a row) and the loop's ds_read_b128 (lane groups
//     of MI355X_MICROARCH.md's LDS table) bank-conflict free;
//   * group scales live in REGISTERS (lane (m, q) holds the scale words of unit q's four columns for its wave's K range:
//     four 16-B requests), applied in fp32 to the MFMA result of each group run (the decode kernels' arithmetic: one-hot
//     rows bit-exact; the MFMA's first step of a group starts from a zero constant);
//   * requests: table word, then the wave's weights (HBM-cold: first), its scale words, then the activations - 4 x the weight
//     bytes, L2-resident, and what bounds this kernel: a CU's texture addresser serves ~1 KB per 22 cycles, 128 KB of
//     activations per workgroup at M = 16 = 3 000 cycles.  They are requested PER MACRO-STEP (a request = 4 rows x the 256 B of
//     one 128-k macro-step: whole lines), two macro-steps ahead: the prologue asks for macro-steps 0 and 1 only, macro-step
//     t's four steps carry the four requests of macro-step t + 2, and the loop starts when macro-step 0's rows are in LDS
//     (first version: all 16 requests up front - the wave's in-order issue stood in the addresser's queue for 2 800 .. 5 200
//     cycles before its first lookup: 6.34 us at M = 16 on 4096^2, profiles/r05/stamps_fastm_run1.jsonl);
//     Measured and dropped (profiles/r05/time_cases_fastm_run3_loader_waves_dropped.jsonl, stamps_fastm_run3.jsonl): four LOADER
//     waves that do nothing but request the activations and write them to LDS, one workgroup barrier per macro-step - 6.11
//     against 6.15 us: what bounds the kernel is not who issues the requests but the 128 KB of activations EVERY workgroup pulls
//     from L2 (all 256 CUs read the same lines at once: ~32 B/clk per CU, 4 000 cycles at M = 16);
//   * epilogue: a wave leaves its 16 x 16 partial tile in its own (no longer needed) activation region, one barrier, every
//     wave sums 256 / W outputs over the W partial tiles in a fixed order and stores them.
// Arithmetic contract: as the decode kernels (include/flute_amd.h): fp32 group scale on the group's partial sum.
// Reference: qgemm_device's main loop for small M (flute/csrc/qgemm_kernel.hpp:617-712), Stream-K fix-up replaced by the
// in-workgroup K split (tile_scheduler_utils.hpp:58-211).
// Round 6 - NG column groups per workgroup (1, 2, 3): the workgroup owns 4 NG unit rows and multiplies NG weight tiles against ONE
// staged activation set - a step's fragment read feeds NG MFMAs, the 128 KB of activations cross the CU's texture addresser once
// for NG x the work.  11008-, 8192- and 6144-wide layers become ONE round of workgroups (230 / 256 / 192) where the NG = 1 form
// needed 2.7 / 2 / 1.5 and lost to the skinny kernel (8192 x 4096: 10.2 against 8.9 us).
// Host contract (api.hip: plan_fastm): num_bits = 4, M <= 16, K == 128 * NM * W, units = N / 4 a multiple of 4 (the last workgroup
// may hold fewer than NG groups: its missing unit rows read as zero and store nothing), group size 2^LG in {64, 128, 256},
// LDS = 32 KB + 32 K bytes <= 160 KB (K <= 4096).
#pragma once
#include "qgemm_oneshot.h"
#include "mfma.h"

namespace flute_amd {

__host__ __device__ constexpr size_t fastm_lds_bytes(int K) { return (size_t)32768 + (size_t)K * 32; }

template <typename T, int TILEP, int W, int NM, int LG, int NG = 1>
__global__ __launch_bounds__(W * 64) void qgemm_fastm_kernel(
    const uint32_t* __restrict__ Qp, const void* __restrict__ Sp, const void* __restrict__ Ap,
    const uint32_t* __restrict__ QM2, void* __restrict__ Dp, int N, int M, uint64_t* __restrict__ stamps) {
    using NT = Num<T>;
    constexpr int KWV = 128 * NM;                                   // k per wave
    constexpr int K = KWV * W;
    constexpr int LK = ilog2_c(K);
    constexpr int lG = LK - LG;                                     // log2(groups per column)
    constexpr int NGW = KWV >> LG;                                  // groups of a wave's K range (8 at g = 64, NM = 4)
    constexpr int SPG = (1 << LG) / 32;                             // 32-k steps per group
    constexpr int CPW = KWV / 8;                                    // 16-B activation chunks per row of a wave's range
    constexpr int XQ = 4;                                           // activation requests per macro-step: 4 rows x 16 chunks each
    constexpr int XA = NM < 2 ? NM : 2;                             // macro-steps of activations requested by the prologue
    constexpr int ENT = 256 / W;
    constexpr int RUNS = 32 / W;
    constexpr uint32_t X_BASE = 32768u;
    constexpr uint32_t REGION = (uint32_t)CPW * 256u;               // a wave's activation region
    static_assert(W == 4 || W == 8, "waves per workgroup");
    static_assert(CPW == 64 || CPW == 32, "a wave's K range: 512 or 256 k");
    static_assert(NM >= 1 && NM <= 4, "macro-steps per wave");
    static_assert(NGW >= 1 && NGW <= 8 && SPG * NGW == 4 * NM, "group runs of whole steps");
    static_assert(32768 + K * 32 <= 160 * 1024, "activations beside the table image");
    static_assert(NG >= 1 && NG <= 3, "column groups per workgroup (a step's 4 NG lookups + 1 fragment read are counted by ONE lgkmcnt)");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (lds_base_of(smem) != 0) __builtin_trap();                  // absolute LDS addresses
#ifdef FLUTE_STAMPS
    uint64_t stamp[16];
    for (int i = 0; i < 16; ++i) stamp[i] = 0;
    stamp[0] = wall_clock64();
    stamp[1] = __builtin_amdgcn_s_memtime();
#define FLUTE_MSTAMP(i) stamp[i] = __builtin_amdgcn_s_memtime()
#else
#define FLUTE_MSTAMP(i)
#endif
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15;                                      // MFMA row (weights) / column (activations)
    const int kg = lane >> 4;                                       // k-chunk of a 32-k step; also: unit of the lane's outputs
    const int ju = lane & 3;                                        // byte of the packed word = column of the unit
    const int uu = i16 >> 2;                                        // unit of the workgroup
    const int unit0 = blockIdx.x * (4 * NG);
    const int nunits = N >> 2;

    // ---- requests, oldest first: table word, weights, scale words, activations ----
    const srd_t lut_srd = make_srd(QM2, 1024u);
    uint32_t lut_v = buf_load4((uint32_t)(wave * ENT + (lane & (ENT - 1))) * 4u, lut_srd);

    // (unit rows past the layer's last - the last workgroup of a layer whose groups NG does not divide - lie past the descriptor: zero)
    const srd_t q_srd = make_srd(Qp + (size_t)unit0 * (K / 2), (uint32_t)min(4 * NG, nunits - unit0) * (uint32_t)K * 2u);
    ring16_t q[NG][NM];
    static_for<NG>([&](auto g_tag) {
        constexpr int g = decltype(g_tag)::value;
        static_for<NM>([&](auto t_tag) {
            constexpr int t = decltype(t_tag)::value;
            const uint32_t vo = (uint32_t)(4 * g + uu) * (uint32_t)(K * 2) + (uint32_t)(wave * KWV + t * 128 + ju * 32 + kg * 8) * 2u;
            q[g][t] = buf_load16_nt(vo, q_srd, 0);
        });
    });
    // scale words of the lane's OUTPUT columns: unit kg, columns r = 0..3, the 8 group slots from the wave's first group
    const srd_t s_srd = make_srd(Sp, (uint32_t)((size_t)N << (lG + 1)));
    ring16_t sc[NG][4];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int unit = unit0 + 4 * g + kg;
        const int col0 = (unit / TILEP) * (4 * TILEP) + (unit % TILEP);
#pragma unroll
        for (int r = 0; r < 4; ++r)      // (a unit past the layer: its columns lie past S - zero)
            sc[g][r] = buf_load16(unit < nunits ? (uint32_t)((((col0 + r * TILEP) << lG) + wave * NGW) * 2) : 0xfffffff0u, s_srd, 0);
    }
    // activations: request (t, r) = rows 4 r .. 4 r + 3 x the 16 chunks of macro-step t; rows >= M lie past the descriptor's
    // range (zero).  Lane l: row 4 r + l / 16, chunk l % 16.
    const srd_t x_srd = make_srd(Ap, (uint32_t)M * (uint32_t)K * 2u);
    ring16_t xv[2][XQ];
    const int xrs = lane >> 4, xcm = lane & 15;
    const uint32_t x_vo = (uint32_t)xrs * (uint32_t)(K * 2) + (uint32_t)(wave * KWV) * 2u + (uint32_t)xcm * 16u;
    auto request_x = [&](auto t_tag, auto r_tag) {
        constexpr int t = decltype(t_tag)::value, r = decltype(r_tag)::value;
        xv[t & 1][r] = buf_load16(x_vo + (uint32_t)(4 * r) * (uint32_t)(K * 2) + (uint32_t)t * 256u, x_srd, 0);
    };
    static_for<XA>([&](auto t_tag) { static_for<XQ>([&](auto r_tag) { request_x(t_tag, r_tag); }); });
    __builtin_amdgcn_sched_barrier(0);
    FLUTE_MSTAMP(2);

    // ---- table image: entry e at [128 e, 128 e + 128): a wave writes RUNS runs of 8 entries (1 KiB, lane-linear) ----
    vm_wait_regs<NG * (NM + 4) + XA * XQ>(lut_v);
    FLUTE_MSTAMP(3);
    {
        uint32_t te[RUNS];
#pragma unroll
        for (int u = 0; u < RUNS; ++u) te[u] = (uint32_t)__builtin_amdgcn_ds_bpermute((u * 8 + (lane >> 3)) * 4, (int)lut_v);
#pragma unroll
        for (int u = 0; u < RUNS; ++u)
            *reinterpret_cast<uint4*>(smem + (uint32_t)(wave * RUNS + u) * 1024u + (uint32_t)lane * 16u) = make_uint4(te[u], te[u], te[u], te[u]);
    }
    FLUTE_MSTAMP(4);
    // ---- activations -> the wave's own region: chunk cl of row m at cl * 256 + (m ^ (cl % 8)) * 16 (cl = 16 t + lane % 16) ----
    const uint32_t xreg = X_BASE + (uint32_t)wave * REGION;
    const uint32_t xw0 = xreg + (uint32_t)xcm * 256u;
    // (YOUNGER: loads of this wave issued after macro-step t's four requests)
    auto stage_x = [&](auto t_tag, auto younger_tag) {
        constexpr int t = decltype(t_tag)::value;
        constexpr int YOUNGER = decltype(younger_tag)::value;
        static_for<XQ>([&](auto r_tag) {
            constexpr int r = decltype(r_tag)::value;
            vm_wait_regs<XQ - 1 - r + YOUNGER>(xv[t & 1][r]);
            const int m = 4 * r + xrs;
            const ring16_t w = xv[t & 1][r];
            *reinterpret_cast<uint4*>(smem + xw0 + (uint32_t)t * 4096u + (uint32_t)((m ^ (xcm & 7)) * 16)) = make_uint4(w.x, w.y, w.z, w.w);
        });
    };
    stage_x(std::integral_constant<int, 0>{}, std::integral_constant<int, (XA - 1) * XQ>{});
    FLUTE_MSTAMP(6);
    // the weights and scale words are older than the activations just waited for: all returned
    static_for<NG>([&](auto g_tag) {
        static_for<NM>([&](auto t_tag) { ring16_t& r = q[decltype(g_tag)::value][decltype(t_tag)::value]; asm volatile("" : "+v"(r) : : "memory"); });
    });
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) { ring16_t& sreg = sc[g][r]; asm volatile("" : "+v"(sreg) : : "memory"); }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                  // the table image is complete
    FLUTE_MSTAMP(7);

    // ---- the loop: NM macro-steps x 4 steps of 32 k; one step of lookahead on the LDS reads; step s of macro-step t carries
    // request s of macro-step t + 2's activations, macro-step t + 1's are written to LDS behind macro-step t's last MFMA ----
    const uint32_t lane_off2 = (uint32_t)(lane & 31) * 8u;          // twice the copy offset (the address is halved after the v_perm)
    const uint32_t sel = 0x0c0c0400u | ((uint32_t)ju << 8);         // {copy offset x 2, byte ju of the word, 0, 0}
    // activation fragment of step (t, s): chunk 16 t + 4 s + kg of the wave's range, slot i16 ^ (4 (s % 2) + kg)
    const uint32_t xa_e = xreg + (uint32_t)kg * 256u + (uint32_t)((i16 ^ kg) * 16);
    const uint32_t xa_o = xreg + (uint32_t)(4 + kg) * 256u + (uint32_t)((i16 ^ (4 + kg)) * 16);
    uint32_t v[2][NG][4];
    ring16_t xb[2];
    auto issue_step = [&](auto n_tag) {
        constexpr int n = decltype(n_tag)::value;
        constexpr int t = n / 4, s = n % 4;
        static_for<NG>([&](auto g_tag) {
            constexpr int g = decltype(g_tag)::value;
            uint32_t ad[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const uint32_t wsrc = (uint32_t)__builtin_amdgcn_mov_dpp((int)q[g][t][c], s * 0x55, 0xF, 0xF, true);       // quad_perm [s, s, s, s]
                ad[c] = __builtin_amdgcn_perm(wsrc, lane_off2, sel) >> 1;
            }
            asm volatile("" : "+v"(ad[0]), "+v"(ad[1]), "+v"(ad[2]), "+v"(ad[3]));
#pragma unroll
            for (int c = 0; c < 4; ++c) v[n & 1][g][c] = lds_lookup32(ad[c]);
        });
        ring16_t& dst = xb[n & 1];
        const uint32_t xa = (s & 1) ? xa_o : xa_e;
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(xa), "n"(t * 4096 + (s >> 1) * 2048) : "memory");
    };
    auto wait_step = [&](auto n_tag, auto younger_tag) {
        constexpr int n = decltype(n_tag)::value;
        uint32_t (&vv)[NG][4] = v[n & 1];
        ring16_t& xx = xb[n & 1];
        constexpr int Y = decltype(younger_tag)::value;
        if constexpr (NG == 1)
            asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(vv[0][0]), "+v"(vv[0][1]), "+v"(vv[0][2]), "+v"(vv[0][3]), "+v"(xx) : "n"(Y) : "memory");
        else if constexpr (NG == 2)
            asm volatile("s_waitcnt lgkmcnt(%9)" : "+v"(vv[0][0]), "+v"(vv[0][1]), "+v"(vv[0][2]), "+v"(vv[0][3]),
                         "+v"(vv[1][0]), "+v"(vv[1][1]), "+v"(vv[1][2]), "+v"(vv[1][3]), "+v"(xx) : "n"(Y) : "memory");
        else
            asm volatile("s_waitcnt lgkmcnt(%13)" : "+v"(vv[0][0]), "+v"(vv[0][1]), "+v"(vv[0][2]), "+v"(vv[0][3]),
                         "+v"(vv[1][0]), "+v"(vv[1][1]), "+v"(vv[1][2]), "+v"(vv[1][3]),
                         "+v"(vv[2 % NG][0]), "+v"(vv[2 % NG][1]), "+v"(vv[2 % NG][2]), "+v"(vv[2 % NG][3]), "+v"(xx) : "n"(Y) : "memory");
    };
    f32x4_t accf[NG], part[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) { accf[g] = f32x4_t{0.f, 0.f, 0.f, 0.f}; part[g] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
    issue_step(std::integral_constant<int, 0>{});
    static_for<4 * NM>([&](auto n_tag) {
        constexpr int n = decltype(n_tag)::value;
        constexpr int t = n / 4, s = n % 4;
        // the next step's reads first - unless it opens a macro-step whose activations are not in LDS yet (written below)
        if constexpr (n + 1 < 4 * NM && s != 3) issue_step(std::integral_constant<int, n + 1>{});
        if constexpr (t + 2 < NM) request_x(std::integral_constant<int, t + 2>{}, std::integral_constant<int, s>{});
        wait_step(n_tag, std::integral_constant<int, (n + 1 < 4 * NM && s != 3) ? 4 * NG + 1 : 0>{});
        const u32x4_t b = {xb[n & 1][0], xb[n & 1][1], xb[n & 1][2], xb[n & 1][3]};
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const u32x4_t a = {v[n & 1][g][0], v[n & 1][g][1], v[n & 1][g][2], v[n & 1][g][3]};
            if constexpr (n % SPG == 0) part[g] = Mfma<T>::run(a, b, f32x4_t{0.f, 0.f, 0.f, 0.f});
            else part[g] = Mfma<T>::run(a, b, part[g]);
            if constexpr (n % SPG == SPG - 1) {
                constexpr int gi = n / SPG;                         // group slot of the wave's range
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const uint32_t w = sc[g][r][gi / 2];
                    accf[g][r] = __builtin_fmaf(part[g][r], scale_to_float<T>((gi & 1) ? (w >> 16) : (w & 0xffffu)), accf[g][r]);
                }
            }
        }
        if constexpr (s == 3 && t + 1 < NM) {
            // macro-step t + 1's rows: requested two macro-steps ago; younger: macro-step t + 2's four requests (just issued)
            stage_x(std::integral_constant<int, t + 1>{}, std::integral_constant<int, (t + 2 < NM) ? XQ : 0>{});
            issue_step(std::integral_constant<int, n + 1>{});
        }
    });
    FLUTE_MSTAMP(9);

    // ---- K split: partial tiles -> the waves' own regions -> one barrier -> every wave sums 256 / W outputs ----
#pragma unroll
    for (int g = 0; g < NG; ++g)
        *reinterpret_cast<float4*>(smem + xreg + (uint32_t)g * 1024u + (uint32_t)lane * 16u) = make_float4(accf[g][0], accf[g][1], accf[g][2], accf[g][3]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    FLUTE_MSTAMP(10);
    constexpr int OPW = 256 / W;                                    // outputs per wave and column group
    if (lane < OPW) {
        const int f = wave * OPW + lane;                            // float f of a partial tile: lane f / 4 of the MFMA layout, register f % 4
        const int ls = f >> 2, r = f & 3;
        const int m = ls & 15;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            float sum = 0.f;
#pragma unroll
            for (int w2 = 0; w2 < W; ++w2) sum += __builtin_bit_cast(float, lds_ld32(X_BASE + (uint32_t)w2 * REGION + (uint32_t)g * 1024u + (uint32_t)f * 4u));
            const int unit = unit0 + 4 * g + (ls >> 4);
            const int col = (unit / TILEP) * (4 * TILEP) + (unit % TILEP) + r * TILEP;
            if (m < M && unit < nunits) reinterpret_cast<uint16_t*>(Dp)[(size_t)m * N + col] = NT::from_float(sum);
        }
    }
#ifdef FLUTE_STAMPS
    FLUTE_MSTAMP(11);
    __builtin_amdgcn_s_waitcnt(0);
    stamp[12] = __builtin_amdgcn_s_memtime();
    stamp[13] = wall_clock64();
    if (lane == 0 && stamps != nullptr) {
        uint64_t* o = stamps + ((size_t)blockIdx.x * W + wave) * 16;
        for (int i = 0; i < 16; ++i) o[i] = stamp[i];
    }
#endif
#undef FLUTE_MSTAMP
}

}  // namespace flute_amd
