// The lean decode kernel for power-of-two K (round 5): M <= 4 rows of a 4-bit layer - the headline regime (M = 1, K = N = 4096).
//
// Round 4's one-shot kernel (qgemm_oneshot.h) executes ~1190 instructions per wave on the headline launch, ~420 of
// them before its barrier, with ONE wave per SIMD.  A wave alone on its SIMD issues an instruction every ~7 cycles
// (dependent VALU -> LDS -> VALU chains, nothing to fill the slots), so the instruction count of the path IS the wave's
// life, and a launch of back-to-back dependent kernels lasts boundary + dispatch ramp (1.9 us for an empty kernel) +
// the life of the workgroups that start last (profiles/r05/fast_lab_run*_stamps.jsonl).  This kernel is the same
// algorithm - table image in LDS (256-B entry stride, 32 copies), wave-private scale image, every request in the
// prologue, pipelined_pieces() as the decode loop - with the instruction count cut to ~760:
//   * K = 512 * D * KW is a COMPILE-TIME constant (D pieces per wave, KW waves share a unit row), so are the waves
//     per workgroup W, the rows MB and TileP: no geometry word, no ragged rows, no dead units, no bounds selects - every
//     wave of the grid holds exactly D whole pieces and every unit is live (host contract below); 145 instructions
//     before the barrier instead of 423;
//   * the scale image is written BEHIND the barrier (the scale words are HBM-cold, unlike the table and the
//     activations: before the barrier a late word held up the whole workgroup: barrier at 2444 -> 1408 cycles);
//   * activations in a linear LDS region (in the holes of the table image the loop's ds_read_b128 used half the banks);
//   * the 8 table addresses of a lookup group are computed before its first lookup (hipcc alternates v_perm / ds_read
//     through one register), a piece's first products start its partial sums (v_dot2 with a zero constant: no
//     register zeroed first): loop 3300 -> 2780 cycles at one wave per SIMD;
//   * transpose-reduce epilogue: the 4 x MB partial sums of a wave are reduced TOGETHER - lanes of a quad keep one
//     column each, the DPP rows keep one activation row each (v_permlane16/32_swap), 22 instructions at MB = 1
//     instead of 4 x 11 - and the outputs of a unit leave in ONE store instruction;
//   * K split across waves (KW > 1): partial sums and arrival tick leave back to back (two LDS round trips, not three).
// Measured and dropped: helper waves that take the set-up loads off the compute waves (a CU's memory pipeline is
// first-in first-out ACROSS its waves: the helpers' small loads queued behind the weight requests the compute waves
// had already issued - table word after 3000 cycles instead of 650, profiles/r05/fast_lab_run2_stamps.jsonl).
// Arithmetic, wire format and LDS table scheme are those of qgemm_oneshot.h (one-hot rows bit-exact, fp32 group scale
// on an 8-k partial sum).  Reference: qgemm_device's prologue + main loop for M <= 4
// (flute/csrc/qgemm_kernel.hpp:546-557, :617-712), Stream-K fix-up replaced by the in-workgroup K split
// (tile_scheduler_utils.hpp:58-211).
//
// Host contract (api.hip: plan_fast): num_bits = 4, 1 <= M <= MB in {1, 2, 4}, K == 512 * D * KW (D = 1 .. 8: K = 3584 is D = 7), units = N / 4 a
// multiple of W / KW, group size in {64, 128, 256}, MB * K * 2 <= 32768, N * (K / g) * 2 < 4 GiB.
// LDS: [table image 64 KB][activations MB x K][W scale images of 4 columns x D * (512 / g) groups][arrival counters +
// K-split partials].
#pragma once
#include "qgemm_oneshot.h"

namespace flute_amd {

__host__ __device__ constexpr size_t fast_lds_bytes(int W, int KW, int D, int lg, int mb = 1) {
    return (size_t)65536 + (size_t)mb * 512 * D * KW * 2 + (size_t)W * 4 * D * (512 >> lg) * 2 + 128 + (size_t)W * 64;
}

// OPT bits (lab): 1 = default-policy weight loads (nt otherwise), 2 = lookups ablated (timing floor), 32 = hipcc's own
// order of table addresses / lookups and zeroed partial sums (pipelined_pieces BA = 0)
// (The fused Hadamard pre-rotation of flute.qgemm_hadamard stays with the round-4 one-shot kernel: this kernel's four waves rotate
// a row of K = 3584 / 4096 in two rounds - measured with the rotation as a template flag, 4.69 / 4.62 us against 4.81 / 4.61 for one
// row on 4096 x 3584 / 4096 x 4096 and 6.28 against 5.99 for two: profiles/r05/call20_hadamard.log - and the flag was removed.)
template <typename T, int TILEP, int W, int KW, int D, int MB = 1, int OPT = 0>
__global__ __launch_bounds__(W * 64) void qgemv_fast_kernel(
    const uint32_t* __restrict__ Qp, const void* __restrict__ Sp, const void* __restrict__ Ap,
    const uint32_t* __restrict__ QM2, void* __restrict__ Dp, int N, int lg, int M,
    uint64_t* __restrict__ stamps) {
    using NT = Num<T>;
    constexpr int K = 512 * D * KW;
    constexpr bool POW2 = (D & (D - 1)) == 0;                       // K a power of two: groups per column and per wave are shifts
    constexpr int LK = ilog2_c(K);
    constexpr int UPW = W / KW;                                     // unit rows per workgroup
    constexpr int ENT = 256 / W;                                    // table entries a wave loads and replicates
    constexpr int RUNS = 32 / W;                                    // 1-KiB runs (8 entries x 128 B of copies) it writes
    constexpr int XP = MB * K / 8;                                  // 16-B pieces of the activation rows
    constexpr int XPR = (XP + W * 64 - 1) / (W * 64);
    constexpr int NSL = (2 * D * 8 + 63) / 64;                      // scale dwords per lane at g = 64 (larger groups: idle lanes)
    constexpr uint32_t X_BASE = 65536u;
    constexpr uint32_t S_BASE = X_BASE + (uint32_t)MB * K * 2u;
    static_assert(W == 4 || W == 8 || W == 16, "waves per workgroup");
    static_assert(MB == 1 || MB == 2 || MB == 4, "rows per pass");
    static_assert((KW & (KW - 1)) == 0 && D >= 1 && D <= 8 && W % KW == 0, "K split: a power of two; 1 .. 8 pieces per wave");
    static_assert(MB * K * 2 <= 32768, "activation rows");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (lds_base_of(smem) != 0) __builtin_trap();                  // v_perm-built table addresses are absolute
#ifdef FLUTE_STAMPS
    uint64_t stamp[16];
    for (int i = 0; i < 16; ++i) stamp[i] = 0;
    stamp[0] = wall_clock64();
    stamp[1] = __builtin_amdgcn_s_memtime();
#define FLUTE_FSTAMP(i) stamp[i] = __builtin_amdgcn_s_memtime()
#else
#define FLUTE_FSTAMP(i)
#endif
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t lane16 = (uint32_t)lane * 16u;
    const int gpp = 512 >> lg;                                      // groups per piece
    const uint32_t red_base = S_BASE + (uint32_t)(W * 4 * D * gpp * 2);      // arrival counters (128 B), then W x 16 partial sums
    const int ul = wave / KW, kpart = wave % KW;
    const int unit = blockIdx.x * UPW + ul;
    const int col0 = (unit / TILEP) * (4 * TILEP) + (unit % TILEP);

    // ---- every request of the wave, oldest first: table word, activations, scale words, weights ----
    const srd_t lut_srd = make_srd(QM2, 1024u);
    uint32_t lut_v = buf_load4((uint32_t)(wave * ENT + (lane & (ENT - 1))) * 4u, lut_srd);
    // rows >= M lie past the descriptor's range and read as zero (the VECTOR offset is what the range check covers)
    const srd_t x_srd = make_srd(Ap, (uint32_t)M * (uint32_t)K * 2u);
    ring16_t xv[XPR];
#pragma unroll
    for (int r = 0; r < XPR; ++r) {
        const int pidx = (r * W + wave) * 64 + lane;
        xv[r] = buf_load16((XP % (W * 64) == 0 || pidx < XP) ? (uint32_t)pidx * 16u : 0x80000000u, x_srd, 0);
    }
    // scale words: lane q = lane + 64 r holds groups (2 gp, 2 gp + 1) of column j of this wave's K range, q = j * (gpw / 2) + gp
    const int lgh = (8 + ilog2_c(D)) - lg;                          // POW2: log2(groups of the wave's range / 2)
    const int lG = LK - lg;                                         // POW2: log2(G)
    const int G = K >> lg;
    const int gpw2 = (D * (512 >> lg)) >> 1;                        // group pairs of the wave's range (7 pieces: 28 / 14 / 7)
    // (q / gpw2 with the three possible divisors as compile-time constants: no integer division in the prologue)
    auto col_of = [&](int q) { return POW2 ? (q >> lgh) : (lg == 6 ? q / (D * 4) : (lg == 7 ? q / (D * 2) : q / D)); };
    const srd_t s_srd = make_srd(Sp, (uint32_t)((size_t)N * (size_t)G * 2));
    uint32_t sv[NSL];
#pragma unroll
    for (int r = 0; r < NSL; ++r) {
        const int q = lane + 64 * r;
        const int j = col_of(q);
        const int gp = POW2 ? (q & ((1 << lgh) - 1)) : (q - j * gpw2);
        const uint32_t vo = POW2 ? (uint32_t)((((col0 + j * TILEP) << lG) + (kpart << (lgh + 1)) + 2 * gp) * 2)
                                 : (uint32_t)(((col0 + j * TILEP) * G + kpart * 2 * gpw2 + 2 * gp) * 2);
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

    // ---- table image: RUNS runs of 8 entries x 128 B; lane l writes 16 B (four copies) of entry l / 8 of the run ----
    vm_wait_regs<XPR + NSL + D>(lut_v);
    FLUTE_FSTAMP(3);
    {
        uint32_t te[RUNS];
#pragma unroll
        for (int u = 0; u < RUNS; ++u) te[u] = (uint32_t)__builtin_amdgcn_ds_bpermute((u * 8 + (lane >> 3)) * 4, (int)lut_v);
#pragma unroll
        for (int u = 0; u < RUNS; ++u) {
            const uint32_t addr = (uint32_t)(wave * ENT + u * 8 + (lane >> 3)) * 256u + (uint32_t)(lane & 7) * 16u;
            *reinterpret_cast<uint4*>(smem + addr) = make_uint4(te[u], te[u], te[u], te[u]);
        }
    }
    FLUTE_FSTAMP(4);
    if (KW > 1 && wave == 0 && lane < UPW) *reinterpret_cast<int*>(smem + red_base + lane * 4) = 0;
    // ---- activations -> LDS [MB][K] ----
    static_for<XPR>([&](auto r_tag) {
        constexpr int r = decltype(r_tag)::value;
        vm_wait_regs<XPR - 1 - r + NSL + D>(xv[r]);
        if constexpr (r == 0) { FLUTE_FSTAMP(5); }
        const int pidx = (r * W + wave) * 64 + lane;
        if (XP % (W * 64) == 0 || pidx < XP) {                      // wave-uniform: a round of a wave is 64 whole pieces
            *reinterpret_cast<uint4*>(smem + X_BASE + (uint32_t)pidx * 16u) = make_uint4(xv[r].x, xv[r].y, xv[r].z, xv[r].w);
        }
    });
    FLUTE_FSTAMP(6);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                  // table image / activations visible to every wave
    FLUTE_FSTAMP(7);

    // ---- scale image [group][column] (wave-private), behind the barrier ----
    const uint32_t sbase = S_BASE + (uint32_t)(wave * 4 * D * gpp * 2);
#pragma unroll
    for (int r = 0; r < NSL; ++r) {
        vm_wait_regs<D>(sv[r]);
        const int qq = lane + 64 * r;
        const int j = col_of(qq);
        const int gp = POW2 ? (qq & ((1 << lgh) - 1)) : (qq - j * gpw2);
        if (j < 4) {
            uint16_t* img = reinterpret_cast<uint16_t*>(smem + sbase) + (2 * gp) * 4 + j;
            img[0] = (uint16_t)(sv[r] & 0xffffu);
            img[4] = (uint16_t)(sv[r] >> 16);
        }
    }
    FLUTE_FSTAMP(8);

    // ---- the decode loop ----
    const uint32_t lane_off = (uint32_t)(lane & 31) * 4u;
    const int gl = (8 * lane) >> lg;                               // group of the lane's 8 k inside a piece
    const uint32_t s_lane = sbase + (uint32_t)(gl * 4) * 2u;
    const uint32_t x_lane = X_BASE + (uint32_t)(kpart * D * 64 + lane) * 16u;       // this lane's 16 B of the wave's first piece, row 0
    float acc[4][MB];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int m = 0; m < MB; ++m) acc[j][m] = 0.f;
    if constexpr (OPT & 2) {
        static_for<D>([&](auto i_tag) {
            constexpr int i = decltype(i_tag)::value;
            asm volatile("s_waitcnt vmcnt(%1)" : "+v"(q[i][0]) : "n"(D - 1 - i) : "memory");
#pragma unroll
            for (int ww = 0; ww < 4; ++ww) acc[ww][0] += __builtin_bit_cast(float, q[i][0][ww]);
        });
    } else {
        pipelined_pieces<T, 4, MB, D, 0, (OPT & 32) ? 0 : 1>(q, x_lane, 10u, (uint32_t)K * 2u, s_lane, (uint32_t)(gpp * 4) * 2u, lane_off, acc);
    }
    FLUTE_FSTAMP(9);

    // ---- transpose-reduce (qgemm_oneshot.h): 4 columns x MB rows of partial sums per lane -> one sum per (column, row) ----
    float v;
    int my_m;
    bool holder;
    transpose_reduce4<MB>(acc, lane, v, my_m, holder);
    FLUTE_FSTAMP(10);
    uint16_t* Dout = reinterpret_cast<uint16_t*>(Dp) + (size_t)my_m * N + col0 + (lane & 3) * TILEP;
    if constexpr (KW == 1) {
        if (holder && my_m < M) *Dout = NT::from_float(v);
    } else {
        // K split: partial sums and the arrival tick leave back to back - the LDS executes a wave's operations in the order it
        // issued them, and one CU's LDS is one pipeline, so the wave that draws the last ticket finds every other wave's
        // partials written - then the last arriver sums them (fixed order: K part 0 first) and stores
        const uint32_t rb = red_base + 128u;
        const uint32_t slot = (uint32_t)(my_m * 4 + (lane & 3)) * 4u;
        uint32_t ticket = 0;
        const uint32_t one = 1u;
        if (holder) {
            const uint32_t mine = rb + (uint32_t)wave * 64u + slot;
            asm volatile("ds_write_b32 %0, %1" : : "v"(mine), "v"(v) : "memory");
        }
        if (lane == 0) {
            const uint32_t tick = red_base + (uint32_t)ul * 4u;
            asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&v"(ticket) : "v"(tick), "v"(one) : "memory");
        }
        const uint32_t t0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)ticket);
        if (t0 == (uint32_t)(KW - 1) && holder && my_m < M) {
            float sum = 0.f;
#pragma unroll
            for (int kp = 0; kp < KW; ++kp) sum += __builtin_bit_cast(float, lds_ld32(rb + (uint32_t)(ul * KW + kp) * 64u + slot));
            *Dout = NT::from_float(sum);
        }
    }
#ifdef FLUTE_STAMPS
    FLUTE_FSTAMP(11);
    __builtin_amdgcn_s_waitcnt(0);
    stamp[12] = __builtin_amdgcn_s_memtime();
    stamp[13] = wall_clock64();
    if (lane == 0 && stamps != nullptr) {
        uint64_t* o = stamps + ((size_t)blockIdx.x * W + wave) * 16;
        for (int i = 0; i < 16; ++i) o[i] = stamp[i];
    }
#endif
#undef FLUTE_FSTAMP
}

}  // namespace flute_amd
