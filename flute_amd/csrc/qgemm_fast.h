// The lean one-row decode kernel for power-of-two K (round 5): the headline regime (M = 1, 4-bit, K = N = 4096).
//
// Round 4's one-shot kernel (qgemm_oneshot.h) executes ~1100 instructions per wave on the headline launch, ~420 of
// them before its barrier, with ONE wave per SIMD: a wave issues an instruction every four to five cycles at best, so
// the instruction count of the path IS the launch time (3.4 us of wave life for 1.65 us of HBM stream; stamps in
// profiles/r03/oneshot_lab3_stamps.jsonl, ISA count in DESIGN.md 3.1d).  This kernel is the same algorithm - table image
// in LDS (256-B entry stride, 32 copies, the activations in its holes), wave-private scale image, every request in the
// prologue, pipelined_pieces() as the decode loop - specialised until the address arithmetic disappears:
//   * K = 512 * D * KW is a COMPILE-TIME constant (D pieces per wave, KW waves share a unit row), so are the waves
//     per workgroup W and TileP: no geometry word, no ragged rows, no dead units, no bounds selects - every wave of
//     the grid holds exactly D whole pieces and every unit is live (host contract below);
//   * TWO (or four) waves per SIMD instead of one: the VALU of one wave issues beside the LDS / scalar / memory
//     instructions of the other (the decode loop is 2 VALU + 1 DS per lookup: 12 issue cycles per lookup for a wave
//     alone, 8 per SIMD with two), and a wave's fixed work (requests, table runs, reduction) is spread over more waves;
//   * the four outputs of a unit leave in ONE store instruction (lanes 0..3), not four.
// Arithmetic, wire format and LDS table scheme are those of qgemm_oneshot.h (one-hot rows bit-exact, fp32 group scale
// on an 8-k partial sum).  Reference: qgemm_device's prologue + main loop for M = 1
// (flute/csrc/qgemm_kernel.hpp:546-557, :617-712), Stream-K fix-up replaced by the in-workgroup K split
// (tile_scheduler_utils.hpp:58-211).
//
// Host contract (api.hip: plan_fast): num_bits = 4, M = 1, K == 512 * D * KW, units = N / 4 a multiple of W / KW,
// group size in {64, 128, 256}, K * 2 <= 32768 (the row fits the holes of the table image), N * (K / g) * 2 < 4 GiB.
// LDS: [table image 64 KB, activations in its holes][W scale images of 4 columns x D * (512 / g) groups][arrival
// counters + K-split partials].
#pragma once
#include "qgemm_oneshot.h"

namespace flute_amd {

__host__ __device__ constexpr size_t fast_lds_bytes(int W, int KW, int D, int lg, bool x_in_holes = false) {
    return (size_t)65536 + (x_in_holes ? 0 : (size_t)512 * D * KW * 2) + (size_t)W * 4 * D * (512 >> lg) * 2 + 128 + (size_t)W * 16;
}
constexpr int ilog2_c(int v) { return v <= 1 ? 0 : 1 + ilog2_c(v >> 1); }

// H = 0: every wave takes its share of the set-up (table runs, activation pieces) before its own weight requests.
// H = 4: four HELPER waves beside the W compute waves request the table words and the activations and build the image; a
//        compute wave's first instructions are its scale and weight requests (an in-order wave can only SEE a load once
//        every older one has returned, so set-up loads in front of the weights delay the stream, behind them they would
//        wait for it), helpers leave at the barrier.
// OPT bits (lab): 1 = default-policy weight loads (nt otherwise), 2 = lookups ablated (timing floor), 4 = scale image
// written before the barrier (first version), 8 = activations in the holes of the table image (first version: half the
// banks for the loop's ds_read_b128), 16 = partial sums / ticket / read as three LDS round trips (first version),
// 32 = hipcc's own order of table addresses and lookups and zeroed partial sums (pipelined_pieces BA = 0: run 2), 64 = four wave_sum64
// reductions instead of the transpose-reduce epilogue (run 2)
template <typename T, int TILEP, int W, int KW, int D, int H = 0, int OPT = 0>
__global__ __launch_bounds__((W + H) * 64) void qgemv_fast_kernel(
    const uint32_t* __restrict__ Qp, const void* __restrict__ Sp, const void* __restrict__ Ap,
    const uint32_t* __restrict__ QM2, void* __restrict__ Dp, int N, int lg, uint64_t* __restrict__ stamps) {
    using NT = Num<T>;
    constexpr int K = 512 * D * KW;
    constexpr int LK = ilog2_c(K);
    constexpr int UPW = W / KW;                                     // unit rows per workgroup
    constexpr int PRO = H ? H : W;                                  // waves that share the set-up
    constexpr int ENT = 256 / PRO;                                  // table entries such a wave loads and replicates
    constexpr int RUNS = 32 / PRO;                                  // 1-KiB runs (8 entries x 128 B of copies) it writes
    constexpr int XP = K / 8;                                       // 16-B pieces of the activation row
    constexpr int XPR = (XP + PRO * 64 - 1) / (PRO * 64);
    constexpr int NSL = (2 * D * 8 + 63) / 64;                      // scale dwords per lane at g = 64 (larger groups: idle lanes)
    constexpr bool XH = (OPT & 8) != 0;
    constexpr bool TR = (OPT & 64) == 0;                            // transpose-reduce epilogue
    constexpr uint32_t X_BASE = 65536u;
    constexpr uint32_t S_BASE = X_BASE + (XH ? 0u : (uint32_t)K * 2u);
    static_assert(W == 4 || W == 8 || W == 16, "waves per workgroup");
    static_assert(H == 0 || H == 4, "helper waves");
    static_assert((KW & (KW - 1)) == 0 && (D & (D - 1)) == 0 && W % KW == 0, "power-of-two split");
    static_assert(ENT <= 64 && RUNS >= 1, "a set-up wave holds its table entries one per lane");
    static_assert(K * 2 <= 32768, "one activation row");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (lds_base_of(smem) != 0) __builtin_trap();                  // v_perm-built table addresses are absolute
#ifdef FLUTE_STAMPS
    uint64_t stamp[16];
    for (int i = 0; i < 16; ++i) stamp[i] = 0;
    stamp[0] = wall_clock64();
    stamp[1] = __builtin_amdgcn_s_memtime();
#define FLUTE_FSTAMP(i) stamp[i] = __builtin_amdgcn_s_memtime()
#define FLUTE_FSTAMP_FLUSH() do { __builtin_amdgcn_s_waitcnt(0); stamp[12] = __builtin_amdgcn_s_memtime(); stamp[13] = wall_clock64(); \
        if ((threadIdx.x & 63) == 0 && stamps != nullptr) { uint64_t* o = stamps + ((size_t)blockIdx.x * (W + H) + (threadIdx.x >> 6)) * 16; \
            for (int i = 0; i < 16; ++i) o[i] = stamp[i]; } } while (0)
#else
#define FLUTE_FSTAMP(i)
#define FLUTE_FSTAMP_FLUSH()
#endif
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t lane16 = (uint32_t)lane * 16u;
    auto x_addr = [&](int pidx) -> uint32_t {                       // LDS byte address of activation piece pidx (8 k)
        return XH ? (uint32_t)((pidx >> 3) * 256 + 128 + (pidx & 7) * 16) : X_BASE + (uint32_t)pidx * 16u;
    };
    const int gpp = 512 >> lg;                                      // groups per piece
    int* arrive = reinterpret_cast<int*>(smem + S_BASE + W * 4 * D * gpp * 2);

    // ---- set-up: table word + activation pieces requested, image runs and activations written (wave pw of PRO) ----
    auto setup_request = [&](int pw, uint32_t& lut_v, ring16_t (&xv)[XPR]) {
        const srd_t lut_srd = make_srd(QM2, 1024u);
        lut_v = buf_load4((uint32_t)(pw * ENT + (lane & (ENT - 1))) * 4u, lut_srd);
        const srd_t x_srd = make_srd(Ap, (uint32_t)K * 2u);
#pragma unroll
        for (int r = 0; r < XPR; ++r) {
            const int pidx = (r * PRO + pw) * 64 + lane;
            xv[r] = buf_load16((XP % (PRO * 64) == 0 || pidx < XP) ? (uint32_t)pidx * 16u : 0x80000000u, x_srd, 0);
        }
    };
    // (YOUNGER = loads of this wave issued after the set-up loads: its scale words and weights, none for a helper)
    auto setup_write = [&](auto younger_tag, int pw, uint32_t& lut_v, ring16_t (&xv)[XPR]) {
        constexpr int YOUNGER = decltype(younger_tag)::value;
        vm_wait_regs<XPR + YOUNGER>(lut_v);
        FLUTE_FSTAMP(3);
        uint32_t te[RUNS];
#pragma unroll
        for (int u = 0; u < RUNS; ++u) te[u] = (uint32_t)__builtin_amdgcn_ds_bpermute((u * 8 + (lane >> 3)) * 4, (int)lut_v);
#pragma unroll
        for (int u = 0; u < RUNS; ++u) {
            const uint32_t addr = (uint32_t)(pw * ENT + u * 8 + (lane >> 3)) * 256u + (uint32_t)(lane & 7) * 16u;
            *reinterpret_cast<uint4*>(smem + addr) = make_uint4(te[u], te[u], te[u], te[u]);
        }
        FLUTE_FSTAMP(4);
        if (KW > 1 && pw == 0 && lane < UPW) arrive[lane] = 0;
        static_for<XPR>([&](auto r_tag) {
            constexpr int r = decltype(r_tag)::value;
            vm_wait_regs<XPR - 1 - r + YOUNGER>(xv[r]);
            if constexpr (r == 0) { FLUTE_FSTAMP(5); }
            const int pidx = (r * PRO + pw) * 64 + lane;
            if (XP % (PRO * 64) == 0 || pidx < XP)
                *reinterpret_cast<uint4*>(smem + x_addr(pidx)) = make_uint4(xv[r].x, xv[r].y, xv[r].z, xv[r].w);
        });
        FLUTE_FSTAMP(6);
    };

    if constexpr (H > 0) {
        if (wave >= W) {                                            // ---- helper wave ----
            uint32_t lut_v;
            ring16_t xv[XPR];
            setup_request(wave - W, lut_v, xv);
            FLUTE_FSTAMP(2);
            setup_write(std::integral_constant<int, 0>{}, wave - W, lut_v, xv);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            FLUTE_FSTAMP(7);
            FLUTE_FSTAMP_FLUSH();
            return;
        }
    }

    // ---- compute wave: unit row ul of the workgroup, K part kpart ----
    const int ul = wave / KW, kpart = wave % KW;
    const int unit = blockIdx.x * UPW + ul;
    const int col0 = (unit / TILEP) * (4 * TILEP) + (unit % TILEP);
    uint32_t lut_v = 0;
    ring16_t xv[XPR];
    if constexpr (H == 0) setup_request(wave, lut_v, xv);

    // scale words: lane q = lane + 64 r holds groups (2 gp, 2 gp + 1) of column j of this wave's K range, q = j * (gpw / 2) + gp
    const int lgh = (8 + ilog2_c(D)) - lg;                          // log2(groups of the wave's range / 2)
    const int lG = LK - lg;                                         // log2(G)
    const srd_t s_srd = make_srd(Sp, (uint32_t)((size_t)N << (lG + 1)));
    uint32_t sv[NSL];
#pragma unroll
    for (int r = 0; r < NSL; ++r) {
        const int q = lane + 64 * r;
        const int j = q >> lgh;
        const int gp = q & ((1 << lgh) - 1);
        const uint32_t vo = (uint32_t)((((col0 + j * TILEP) << lG) + (kpart << (lgh + 1)) + 2 * gp) * 2);
        sv[r] = buf_load4_at(j < 4 ? vo : 0x80000000u, s_srd, 0);
    }
    const srd_t q_srd = make_srd(Qp + (size_t)unit * (K / 2), (uint32_t)K * 2u);
    ring16_t q[D][1];
    static_for<D>([&](auto i_tag) {
        constexpr int i = decltype(i_tag)::value;
        const uint32_t vo = lane16 + (uint32_t)(kpart * D + i) * 1024u;
        if constexpr (OPT & 1) q[i][0] = buf_load16(vo, q_srd, 0);
        else q[i][0] = buf_load16_nt(vo, q_srd, 0);
    });
    __builtin_amdgcn_sched_barrier(0);
    FLUTE_FSTAMP(2);
    if constexpr (H == 0) setup_write(std::integral_constant<int, NSL + D>{}, wave, lut_v, xv);

    // ---- scale image [group][column] (wave-private): behind the barrier, so that a late scale word (HBM-cold, unlike the
    // table and the activations) holds up its own wave only ----
    const uint32_t sbase = S_BASE + (uint32_t)(wave * 4 * D * gpp * 2);
    auto scale_image = [&]() {
#pragma unroll
        for (int r = 0; r < NSL; ++r) {
            vm_wait_regs<D>(sv[r]);
            const int qq = lane + 64 * r;
            const int j = qq >> lgh;
            const int gp = qq & ((1 << lgh) - 1);
            if (j < 4) {
                uint16_t* img = reinterpret_cast<uint16_t*>(smem + sbase) + (2 * gp) * 4 + j;
                img[0] = (uint16_t)(sv[r] & 0xffffu);
                img[4] = (uint16_t)(sv[r] >> 16);
            }
        }
    };
    if constexpr (OPT & 4) { scale_image(); FLUTE_FSTAMP(8); }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                  // table image / activations visible to every wave
    FLUTE_FSTAMP(7);
    if constexpr (!(OPT & 4)) { scale_image(); FLUTE_FSTAMP(8); }

    // ---- the decode loop ----
    const uint32_t lane_off = (uint32_t)(lane & 31) * 4u;
    const int gl = (8 * lane) >> lg;                               // group of the lane's 8 k inside a piece
    const uint32_t s_lane = sbase + (uint32_t)(gl * 4) * 2u;
    const uint32_t x_lane = x_addr(kpart * D * 64 + lane);        // this lane's 16 B of the wave's first piece
    float acc[4][1];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j][0] = 0.f;
    if constexpr (OPT & 2) {
        static_for<D>([&](auto i_tag) {
            constexpr int i = decltype(i_tag)::value;
            asm volatile("s_waitcnt vmcnt(%1)" : "+v"(q[i][0]) : "n"(D - 1 - i) : "memory");
#pragma unroll
            for (int ww = 0; ww < 4; ++ww) acc[ww][0] += __builtin_bit_cast(float, q[i][0][ww]);
        });
    } else {
        pipelined_pieces<T, 4, 1, D, 0, (OPT & 32) ? 0 : 1>(q, x_lane, XH ? 11u : 10u, 0u, s_lane, (uint32_t)(gpp * 4) * 2u, lane_off, acc);
    }
    FLUTE_FSTAMP(9);

    // ---- lanes -> wave (DPP; lanes 0..15 all hold the sum) -> [K split: waves -> LDS -> last arriver] -> ONE store ----
    float tot[4];
    float colsum = 0.f;                                            // TR: lane l holds the wave's sum of column l % 4
    if constexpr (TR) {
        // transpose-reduce: 22 instructions instead of 4 x wave_sum64's 11.  Quad: lane keeps column (lane & 1) of each column pair and
        // hands the other to its neighbour, then the same between the pairs - every lane of a quad ends with the quad's sum of column
        // lane & 3; rows: row_ror 4 / 8 (the rotation keeps lane & 3); wave: the two lane-swap instructions of gfx950.
        const bool o1 = (lane & 1) != 0, o2 = (lane & 2) != 0;
        auto dpp_add = [](float keep, float send, auto ctrl_tag) {
            return keep + __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, send), decltype(ctrl_tag)::value, 0xF, 0xF, true));
        };
        const float k01 = dpp_add(o1 ? acc[1][0] : acc[0][0], o1 ? acc[0][0] : acc[1][0], std::integral_constant<int, 0xB1>{});   // quad_perm [1,0,3,2]
        const float k23 = dpp_add(o1 ? acc[3][0] : acc[2][0], o1 ? acc[2][0] : acc[3][0], std::integral_constant<int, 0xB1>{});
        float k = dpp_add(o2 ? k23 : k01, o2 ? k01 : k23, std::integral_constant<int, 0x4E>{});                                     // quad_perm [2,3,0,1]
        k = dpp_add(k, k, std::integral_constant<int, 0x124>{});                                                                  // row_ror:4
        k = dpp_add(k, k, std::integral_constant<int, 0x128>{});                                                                  // row_ror:8
        float a = k, b = k;
        asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));      // a: rows 0, 0, 2, 2 of k; b: rows 1, 1, 3, 3
        k = a + b;
        a = k; b = k;
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));      // a: lower half twice, b: upper half twice
        colsum = a + b;
#pragma unroll
        for (int j = 0; j < 4; ++j) tot[j] = 0.f;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) tot[j] = wave_sum64(acc[j][0]);
    }
    FLUTE_FSTAMP(10);
    uint16_t* Dout = reinterpret_cast<uint16_t*>(Dp) + col0;
    if constexpr (KW == 1) {
        if (lane < 4) {
            const float v = TR ? colsum : (lane == 0 ? tot[0] : (lane == 1 ? tot[1] : (lane == 2 ? tot[2] : tot[3])));
            Dout[lane * TILEP] = NT::from_float(v);
        }
    } else if constexpr (TR) {
        // column sums already one per lane: lanes 0..3 leave theirs (one ds_write_b32), lane 0 draws the ticket behind it
        const uint32_t rb = (uint32_t)(S_BASE + W * 4 * D * gpp * 2) + 128u;
        const uint32_t tick = (uint32_t)(S_BASE + W * 4 * D * gpp * 2) + (uint32_t)ul * 4u;
        uint32_t ticket = 0;
        const uint32_t one = 1u;
        if (lane < 4) {
            const uint32_t mine = rb + (uint32_t)wave * 16u + (uint32_t)lane * 4u;
            asm volatile("ds_write_b32 %0, %1" : : "v"(mine), "v"(colsum) : "memory");
        }
        if (lane == 0)
            asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&v"(ticket) : "v"(tick), "v"(one) : "memory");
        const uint32_t t0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)ticket);
        if (t0 == (uint32_t)(KW - 1) && lane < 4) {
            float sum = 0.f;
#pragma unroll
            for (int kp = 0; kp < KW; ++kp) sum += __builtin_bit_cast(float, lds_ld32(rb + (uint32_t)((ul * KW + kp) * 16 + lane * 4)));
            Dout[lane * TILEP] = NT::from_float(sum);
        }
    } else if constexpr (OPT & 16) {
        // no barrier: every wave leaves its partial sums and an arrival tick in LDS, the last arriver sums and stores
        // (release on the tick / acquire by the reader: the partials are ordered before it)
        float* rb = reinterpret_cast<float*>(arrive + 32);
        if (lane == 0) *reinterpret_cast<float4*>(rb + wave * 4) = make_float4(tot[0], tot[1], tot[2], tot[3]);
        int ticket = 0;
        if (lane == 0) ticket = __hip_atomic_fetch_add(&arrive[ul], 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
        ticket = __builtin_amdgcn_readfirstlane(ticket);
        if (ticket == KW - 1 && lane < 4) {
            float sum = 0.f;
#pragma unroll
            for (int kp = 0; kp < KW; ++kp) sum += rb[(ul * KW + kp) * 4 + lane];
            Dout[lane * TILEP] = NT::from_float(sum);
        }
    } else {
        // The same in TWO LDS round trips: the partial sums and the arrival tick leave back to back - the LDS executes a
        // wave's operations in the order it issued them, and one CU's LDS is one pipeline, so the wave that draws the last
        // ticket finds every other wave's partials written - then the last arriver reads them all (fixed order: K part 0 first).
        const uint32_t rb = (uint32_t)(S_BASE + W * 4 * D * gpp * 2) + 128u;
        const uint32_t mine = rb + (uint32_t)wave * 16u;
        const uint32_t tick = (uint32_t)(S_BASE + W * 4 * D * gpp * 2) + (uint32_t)ul * 4u;
        f32x4_t pv = {tot[0], tot[1], tot[2], tot[3]};
        uint32_t ticket = 0;
        const uint32_t one = 1u;
        if (lane == 0)
            asm volatile("ds_write_b128 %1, %2\n\tds_add_rtn_u32 %0, %3, %4\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(ticket) : "v"(mine), "v"(pv), "v"(tick), "v"(one) : "memory");
        const uint32_t t0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)ticket);
        if (t0 == (uint32_t)(KW - 1) && lane < 4) {
            float sum = 0.f;
#pragma unroll
            for (int kp = 0; kp < KW; ++kp) sum += __builtin_bit_cast(float, lds_ld32(rb + (uint32_t)((ul * KW + kp) * 16 + lane * 4)));
            Dout[lane * TILEP] = NT::from_float(sum);
        }
    }
    FLUTE_FSTAMP(11);
    FLUTE_FSTAMP_FLUSH();
#undef FLUTE_FSTAMP
#undef FLUTE_FSTAMP_FLUSH
}

}  // namespace flute_amd
