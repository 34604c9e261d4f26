// Latency-bound decode launches (round 3): one visit per workgroup, every wave's whole K slice requested by
// the prologue - the headline regime (M = 1, 4096 x 4096: 8.9 MB in one launch of ~4 us, of which the HBM
// stream is < 1.5 us).  Replaces the one-shot instantiations of qgemm_stream.h; same arithmetic, same table /
// lookup scheme, same wire format.  Reference: qgemm_device's prologue + main loop
// (flute/csrc/qgemm_kernel.hpp:546-557 table staging, :617-712 loop) for M <= 4.
//
// What the round-2 stamps showed and what this kernel does about it (profiles/r02_stamps_decode.json,
// profiles/r03_oneshot_lab*.jsonl):
//   * kernel arguments arrived by s_load (a scalar-cache miss) before anything could be requested.  Here the
//     first 14 argument dwords are PRELOADED into SGPRs by the dispatcher (-mllvm
//     -amdgpu-kernarg-preload-count: the four source pointers, K, N, the packed launch geometry, M, D).
//   * a CU's texture addresser takes ~16 cycles per wave-wide memory instruction, whatever its width: with
//     8 waves x 9 requests the LAST weight request left the CU ~1000 cycles after the wave started.  Here a
//     wave issues 1 table + XPR activation + NSL scale + D weight requests (7 for the headline).
//   * every wave waited for ALL of its weights and its (youngest, HBM-cold) scale block before the first
//     lookup.  Here the scale words are requested BEFORE the weights, each piece is released by its own
//     counted vmcnt, and (PIPE) the lookups of half-piece h+1 are in flight while half-piece h is multiplied.
//   * the table image was written with 8-way bank conflicts; here a wave writes whole 1-KiB runs (8 entries x
//     128 B, lane l -> 16-B piece l of the run), the entry word fetched from the lane that loaded it by
//     ds_bpermute.
//   * scale staging: one aligned dword (two groups) per lane into a wave-private [group][column] image (G must be
//     even: a buffer_load_dword of an odd element index returns the aligned-down dword - measured; odd G takes
//     the ring kernel); a piece reads its J scales with ONE vector read.
// LDS: [table image][activations MB x KX][per-wave scale images][arrival counters + K-split partials].
#pragma once
#include "qgemm_stream.h"

namespace flute_amd {

constexpr int ilog2_c(int v) { return v <= 1 ? 0 : 1 + ilog2_c(v >> 1); }

// launch geometry packed into one kernel-argument dword (preloaded)
struct OneGeo {
    static constexpr uint32_t pack(int lg, int lkw, int upw, int pk, int ipw, int had_log, int xh) {
        return (uint32_t)lg | ((uint32_t)lkw << 4) | ((uint32_t)upw << 8) | ((uint32_t)pk << 13) |
               ((uint32_t)ipw << 17) | ((uint32_t)had_log << 24) | ((uint32_t)xh << 28);
    }
};

__host__ __device__ constexpr int oneshot_max_threads(int bits, int mb) {
    return (bits == 3 || (bits == 2 && mb == 4)) ? 512 : 1024;
}
// table image: bytes, 1-KiB write runs
__host__ __device__ constexpr int oneshot_lut_bytes(int bits) { return bits == 3 ? 64 * 128 : 65536; }
__host__ __device__ constexpr int oneshot_lut_runs(int bits) { return bits == 4 ? 32 : (bits == 3 ? 8 : 64); }
// dwords of scales a wave stages: J columns x D pieces x (512 >> lg) groups / 2; loads per lane for g >= 64
__host__ __device__ constexpr int oneshot_scale_loads(int bits, int depth) {
    return ((bits == 3 ? 16 : 16 / bits) * depth * 8 / 2 + 63) / 64;
}
// 4-bit table image: 256-B entry stride (the lookup address is one v_perm), 128 B of copies per entry - the other
// 128 B of every slot ("holes", 32 KB in all) hold the staged activations when they fit: 64 k of a row per hole
__host__ __device__ constexpr bool oneshot_x_in_holes(int bits, int mb, int K) {
    return bits == 4 && (size_t)mb * ((K + 511) / 512 * 512) * 2 <= 32768;
}
// dynamic LDS of a launch (the kernel carves with the same formulas)
__host__ __device__ constexpr size_t oneshot_lds_bytes(int bits, int mb, int depth, int lg, int K, int waves) {
    const int J = (bits == 3) ? 16 : 16 / bits;
    return (size_t)oneshot_lut_bytes(bits) + (oneshot_x_in_holes(bits, mb, K) ? 0 : (size_t)mb * ((K + 511) / 512 * 512) * 2) +
           (size_t)waves * J * depth * (512 >> lg) * 2 + 128 + (size_t)waves * J * mb * 4;
}

__device__ __forceinline__ uint32_t buf_load4_at(uint32_t voff, srd_t srd, uint32_t soff) {
    uint32_t v;
    asm volatile("s_nop 4\n\tbuffer_load_dword %0, %1, %2, %3 offen" : "=v"(v) : "v"(voff), "s"(srd), "s"(soff) : "memory");
    return v;
}
// compile-time loop: f(std::integral_constant<int, 0>{}), ..., f(std::integral_constant<int, N - 1>{})
template <int N, typename F> __device__ __forceinline__ void static_for(F&& f) {
    [&]<int... I>(std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }(std::make_integer_sequence<int, N>{});
}

// hidden LDS reads of the pipelined piece loop (released by the counted lgkmcnt of the lookups behind them)
__device__ __forceinline__ ring16_t lds_hidden128(uint32_t addr) {
    ring16_t v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
    return v;
}
__device__ __forceinline__ u32x2_t lds_hidden64(uint32_t addr) {
    u32x2_t v;
    asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(addr) : "memory");
    return v;
}

// The software-pipelined piece loop (one-shot and persistent kernels): D pieces already requested into q[][],
// YB = loads of this wave that are younger than q's last piece (0 for the one-shot kernel; the next segment's requests
// in the persistent kernel).  EVERY piece must be a whole, valid piece (the host guarantees it).  A piece is NG groups
// of 8 lookups: 4 bits: (k-pair words 0,1 | 2,3) x 4 columns; 2 bits: the same with byte lookups (two columns each);
// 3 bits: k-pair word ww x fields (0..7 | 8..15).  The lookups of group g+1 - and, on a piece boundary, its
// activation / scale reads (hidden) - are issued before the dot products of group g; LDS returns in order, so "at
// most NEXT_READS younger operations outstanding" releases group g.
// BA = 1 (qgemm_fast.h, 4 bits): the 8 table addresses of a group are all computed BEFORE its first lookup is issued (hipcc otherwise
// alternates v_perm / ds_read through one address register: every lookup then waits for the VALU result it was just handed - visible
// with ONE wave per SIMD, where nothing else fills the slot)
template <typename T, int BITS, int MB, int D, int YB, int BA = 0>
__device__ __forceinline__ void pipelined_pieces(ring16_t (&q)[D][Layout<BITS>::NPLANES], uint32_t x_lane, uint32_t x_pshift,
                                                 uint32_t x_row, uint32_t s_lane, uint32_t s_piece, uint32_t lane_off,
                                                 float (&acc)[Layout<BITS>::J][MB]) {
    using NT = Num<T>;
    constexpr int J = Layout<BITS>::J;
    constexpr int NP = Layout<BITS>::NPLANES;
    constexpr int NG = (BITS == 3) ? 8 : 2;                    // groups per piece
    using look_t = std::conditional_t<BITS == 2, u32x2_t, uint32_t>;
    look_t v[2][8];
    ring16_t xq[2][MB];
    // J scales of a piece: 8 B (J = 4) / 16 B / 32 B - read straight into the registers they are used from (a hidden
    // read's destination must not be copied before its wait)
    u32x2_t sq2[2];
    ring16_t sq4[2][(J + 7) / 8];
    float al[J][MB];
    auto issue_group = [&](auto g_tag) {
        constexpr int g = decltype(g_tag)::value;
        constexpr int I = g / NG, GI = g % NG;
        if constexpr (GI == 0) {
            if constexpr (NP == 1) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(q[I][0]) : "n"((D - 1 - I) * NP + YB) : "memory");
            else asm volatile("s_waitcnt vmcnt(%3)" : "+v"(q[I][0]), "+v"(q[I][1]), "+v"(q[I][2]) : "n"((D - 1 - I) * NP + YB) : "memory");
#pragma unroll
            for (int m = 0; m < MB; ++m) xq[I & 1][m] = lds_hidden128(x_lane + ((uint32_t)I << x_pshift) + (uint32_t)m * x_row);
            const uint32_t sa = s_lane + (uint32_t)I * s_piece;
            if constexpr (J == 4) sq2[I & 1] = lds_hidden64(sa);
            else {
#pragma unroll
                for (int c = 0; c < J / 8; ++c) sq4[I & 1][c] = lds_hidden128(sa + 16u * c);
            }
        }
        if constexpr (BITS == 4 && BA != 0) {
            uint32_t ad[8];
#pragma unroll
            for (int ww = 0; ww < 2; ++ww)
#pragma unroll
                for (int j = 0; j < 4; ++j) ad[ww * 4 + j] = __builtin_amdgcn_perm(q[I][0][2 * GI + ww], lane_off, 0x0c0c0400u | ((4u + j) << 8));
            asm volatile("" : "+v"(ad[0]), "+v"(ad[1]), "+v"(ad[2]), "+v"(ad[3]), "+v"(ad[4]), "+v"(ad[5]), "+v"(ad[6]), "+v"(ad[7]));
#pragma unroll
            for (int n = 0; n < 8; ++n) v[g & 1][n] = lds_lookup32(ad[n]);
        } else if constexpr (BITS == 4) {
#pragma unroll
            for (int ww = 0; ww < 2; ++ww)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    v[g & 1][ww * 4 + j] = lds_lookup32(__builtin_amdgcn_perm(q[I][0][2 * GI + ww], lane_off, 0x0c0c0400u | ((4u + j) << 8)));
        } else if constexpr (BITS == 2) {
#pragma unroll
            for (int ww = 0; ww < 2; ++ww)
#pragma unroll
                for (int jp = 0; jp < 4; ++jp)
                    v[g & 1][ww * 4 + jp] = lds_lookup64(__builtin_amdgcn_perm(q[I][0][2 * GI + ww], lane_off, 0x0c0c0400u | ((4u + jp) << 8)));
        } else if constexpr (BA != 0) {
            constexpr int ww = GI / 2, j0 = (GI % 2) * 8;
            const uint32_t w[3] = {q[I][0][ww], q[I][1][ww], q[I][2][ww]};
            uint32_t ad[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) ad[j] = (field<3>(w, j0 + j) << 7) | lane_off;
            asm volatile("" : "+v"(ad[0]), "+v"(ad[1]), "+v"(ad[2]), "+v"(ad[3]), "+v"(ad[4]), "+v"(ad[5]), "+v"(ad[6]), "+v"(ad[7]));
#pragma unroll
            for (int j = 0; j < 8; ++j) v[g & 1][j] = lds_lookup32(ad[j]);
        } else {
            constexpr int ww = GI / 2, j0 = (GI % 2) * 8;
            const uint32_t w[3] = {q[I][0][ww], q[I][1][ww], q[I][2][ww]};
#pragma unroll
            for (int j = 0; j < 8; ++j) v[g & 1][j] = lds_lookup32((field<3>(w, j0 + j) << 7) | lane_off);
        }
    };
    auto wait_group = [&](auto g_tag, auto younger_tag) {
        constexpr int g = decltype(g_tag)::value;
        constexpr int YOUNGER = decltype(younger_tag)::value;
        constexpr int I = g / NG;
        look_t(&vv)[8] = v[g & 1];
        asm volatile("s_waitcnt lgkmcnt(%8)"
                     : "+v"(vv[0]), "+v"(vv[1]), "+v"(vv[2]), "+v"(vv[3]), "+v"(vv[4]), "+v"(vv[5]), "+v"(vv[6]), "+v"(vv[7])
                     : "n"(YOUNGER) : "memory");
        if constexpr (g % NG == 0) {                           // the piece's activation / scale reads are older than its first group
#pragma unroll
            for (int m = 0; m < MB; ++m) { ring16_t& r = xq[I & 1][m]; asm volatile("" : "+v"(r) : : "memory"); }
            u32x2_t& s2 = sq2[I & 1];
            if constexpr (J == 4) asm volatile("" : "+v"(s2) : : "memory");
#pragma unroll
            for (int c = 0; c < (J + 7) / 8; ++c) { ring16_t& r = sq4[I & 1][c]; if constexpr (J != 4) asm volatile("" : "+v"(r) : : "memory"); }
        }
    };
    issue_group(std::integral_constant<int, 0>{});
    static_for<NG * D>([&](auto g_tag) {
        constexpr int g = decltype(g_tag)::value;
        constexpr int I = g / NG, GI = g % NG;
        if constexpr (g + 1 < NG * D) issue_group(std::integral_constant<int, g + 1>{});
        // what was issued behind group g: the next group's 8 lookups and, on a piece boundary, its MB + ceil(J / 8) reads
        constexpr int NEXT_READS = (g + 1 < NG * D) ? 8 + (((g + 1) % NG == 0) ? MB + (J + 7) / 8 : 0) : 0;
        wait_group(g_tag, std::integral_constant<int, (NEXT_READS < 15 ? NEXT_READS : 15)>{});
        if constexpr (GI == 0 && !(BITS != 2 && BA != 0)) {
#pragma unroll
            for (int j = 0; j < J; ++j)
#pragma unroll
                for (int m = 0; m < MB; ++m) al[j][m] = 0.f;
        }
        if constexpr (BITS == 4) {
#pragma unroll
            for (int ww = 0; ww < 2; ++ww)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int m = 0; m < MB; ++m) {
                        // (BA: a piece's first products start its partial sums - no register zeroed first)
                        if (BA != 0 && GI == 0 && ww == 0) al[j][m] = NT::dot2z(v[g & 1][ww * 4 + j], xq[I & 1][m][2 * GI + ww]);
                        else al[j][m] = NT::dot2(v[g & 1][ww * 4 + j], xq[I & 1][m][2 * GI + ww], al[j][m]);
                    }
        } else if constexpr (BITS == 2) {
#pragma unroll
            for (int ww = 0; ww < 2; ++ww)
#pragma unroll
                for (int jp = 0; jp < 4; ++jp)
#pragma unroll
                    for (int m = 0; m < MB; ++m) {
                        al[2 * jp][m] = NT::dot2(v[g & 1][ww * 4 + jp].x, xq[I & 1][m][2 * GI + ww], al[2 * jp][m]);
                        al[2 * jp + 1][m] = NT::dot2(v[g & 1][ww * 4 + jp].y, xq[I & 1][m][2 * GI + ww], al[2 * jp + 1][m]);
                    }
        } else {
            constexpr int ww = GI / 2, j0 = (GI % 2) * 8;
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int m = 0; m < MB; ++m) {
                    if (BA != 0 && ww == 0) al[j0 + j][m] = NT::dot2z(v[g & 1][j], xq[I & 1][m][ww]);       // the piece's first k pair starts the partial sums
                    else al[j0 + j][m] = NT::dot2(v[g & 1][j], xq[I & 1][m][ww], al[j0 + j][m]);
                }
        }
        if constexpr (GI == NG - 1) {
#pragma unroll
            for (int j = 0; j < J; ++j) {
                uint32_t w;
                if constexpr (J == 4) w = sq2[I & 1][j / 2];
                else w = sq4[I & 1][j / 8][(j % 8) / 2];
                const float sf = scale_to_float<T>((j & 1) ? (w >> 16) : (w & 0xffffu));
#pragma unroll
                for (int m = 0; m < MB; ++m) acc[j][m] = __builtin_fmaf(al[j][m], sf, acc[j][m]);
            }
        }
    });
}

// Transpose-reduce of a 4-bit wave's partial sums (round 5): acc[4 columns][MB rows] per lane -> ONE sum per (column, row),
// left in `v` of the HOLDER lanes (lane & 3 = column, row = my_m): 22 instructions at MB = 1 where four wave_sum64 take 44.
// Quad: a lane keeps column (lane & 1) of each column pair and hands the other to its neighbour, then the same between the
// pairs - every lane of a quad ends with the quad's sum of column lane & 3 (9 instructions per row).  The four 16-lane DPP
// rows: v_permlane32_swap a, b exchanges a's upper half with b's lower half, so a + b holds row-pair sums of a in the lower
// half and of b in the upper; v_permlane16_swap does the same between odd and even rows.  MB = 1: both with copies of the value
// (plain sums); MB = 2: activation row m in half m; MB = 4: DPP row r holds activation row {0, 2, 1, 3}[r].  Last, inside a
// DPP row: row_ror 4 / 8 (the rotation keeps lane & 3).
template <int MB>
__device__ __forceinline__ void transpose_reduce4(const float (&acc)[4][MB], int lane, float& v, int& my_m, bool& holder) {
    auto dpp_add = [](float keep, float send, auto ctrl_tag) {
        return keep + __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, send), decltype(ctrl_tag)::value, 0xF, 0xF, true));
    };
    const bool o1 = (lane & 1) != 0, o2 = (lane & 2) != 0;
    float kq[MB];
#pragma unroll
    for (int m = 0; m < MB; ++m) {
        const float k01 = dpp_add(o1 ? acc[1][m] : acc[0][m], o1 ? acc[0][m] : acc[1][m], std::integral_constant<int, 0xB1>{});   // quad_perm [1,0,3,2]
        const float k23 = dpp_add(o1 ? acc[3][m] : acc[2][m], o1 ? acc[2][m] : acc[3][m], std::integral_constant<int, 0xB1>{});
        kq[m] = dpp_add(o2 ? k23 : k01, o2 ? k01 : k23, std::integral_constant<int, 0x4E>{});                                     // quad_perm [2,3,0,1]
    }
    auto swap32_add = [](float a, float b) { asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b)); return a + b; };
    auto swap16_add = [](float a, float b) { asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b)); return a + b; };
    static_assert(MB == 1 || MB == 2 || MB == 4, "rows per pass");
    if constexpr (MB == 1) { v = swap16_add(kq[0], kq[0]); v = swap32_add(v, v); my_m = 0; }
    else if constexpr (MB == 2) { v = swap32_add(kq[0], kq[1]); v = swap16_add(v, v); my_m = lane >> 5; }
    else { const float ab = swap32_add(kq[0], kq[1]), cd = swap32_add(kq[2], kq[3]); v = swap16_add(ab, cd); my_m = ((lane >> 4) & 1) * 2 + (lane >> 5); }
    v = dpp_add(v, v, std::integral_constant<int, 0x124>{});                                                                      // row_ror:4
    v = dpp_add(v, v, std::integral_constant<int, 0x128>{});                                                                      // row_ror:8
    holder = (lane & (MB == 1 ? 63 : (MB == 2 ? 31 : 15))) < 4;   // lanes 0..3 of the first DPP row that holds the activation row
}

// The same for a 3-bit unit's 16 columns: acc[16][MB] per lane -> one sum per (column, row); the holder of column j of row my_m is lane
// (lane & 15) == j of the first DPP row that holds the activation row.  53 instructions at MB = 1 where sixteen wave_sum64 take 176.
// Levels: lane pairs (column 2 p + (lane & 1) of every column pair), quads (column 4 r + (lane & 3)), the four quads of a DPP row
// (a two-stage barrel shifter rotates the lane's four values by its quad index, row_ror 4 / 8 / 12 hands value k to the quad k
// further on: every lane ends with column lane & 15 summed over its row), the four rows (lane-swap instructions, as above).
template <int MB>
__device__ __forceinline__ void transpose_reduce16(const float (&acc)[16][MB], int lane, float& v, int& my_m, bool& holder) {
    static_assert(MB == 1 || MB == 2, "rows per pass");
    auto dpp_add = [](float keep, float send, auto ctrl_tag) {
        return keep + __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, send), decltype(ctrl_tag)::value, 0xF, 0xF, true));
    };
    const bool o1 = (lane & 1) != 0, o2 = (lane & 2) != 0, q1 = (lane & 4) != 0, q2 = (lane & 8) != 0;
    float vr[MB];
#pragma unroll
    for (int m = 0; m < MB; ++m) {
        float k1[8], k2[4], s1[4], r[4];
#pragma unroll
        for (int p = 0; p < 8; ++p) k1[p] = dpp_add(o1 ? acc[2 * p + 1][m] : acc[2 * p][m], o1 ? acc[2 * p][m] : acc[2 * p + 1][m], std::integral_constant<int, 0xB1>{});
#pragma unroll
        for (int c = 0; c < 4; ++c) k2[c] = dpp_add(o2 ? k1[2 * c + 1] : k1[2 * c], o2 ? k1[2 * c] : k1[2 * c + 1], std::integral_constant<int, 0x4E>{});
#pragma unroll
        for (int i = 0; i < 4; ++i) s1[i] = q1 ? k2[(i + 1) & 3] : k2[i];
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = q2 ? s1[(i + 2) & 3] : s1[i];                 // r[i] = k2[(quad + i) % 4]
        float t = dpp_add(r[0], r[1], std::integral_constant<int, 0x124>{});             // row_ror:4: from the lane one quad back, whose r[1] is MY column
        t = dpp_add(t, r[2], std::integral_constant<int, 0x128>{});
        vr[m] = dpp_add(t, r[3], std::integral_constant<int, 0x12C>{});
    }
    auto swap32_add = [](float a, float b) { asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b)); return a + b; };
    auto swap16_add = [](float a, float b) { asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b)); return a + b; };
    if constexpr (MB == 1) { v = swap16_add(vr[0], vr[0]); v = swap32_add(v, v); my_m = 0; }
    else { v = swap32_add(vr[0], vr[1]); v = swap16_add(v, v); my_m = lane >> 5; }
    holder = (lane & (MB == 1 ? 63 : 31)) < 16;
}

// OPT bits (development / A-B measurements, tools/ubench/oneshot_lab.hip): 1 = nt on the weight loads,
// 2 = ablate the lookups (timing floor: stream + prologue only), 8 = wait for every piece before the first
// lookup (round-2 behaviour), 16 = software-pipelined lookup groups (every wave of the launch must hold D pieces),
// 32 = weight requests interleaved with the table image's LDS work
template <typename T, int BITS, int TILEP, int MB, int D, int XPR, bool HAD = false, int OPT = 0>
__global__ __launch_bounds__(oneshot_max_threads(BITS, MB)) void qgemv_oneshot_kernel(
    const uint32_t* __restrict__ Qp, const void* __restrict__ Sp, const void* __restrict__ Ap,
    const uint32_t* __restrict__ QM2, int K, int N, uint32_t geo, int M, void* __restrict__ Dp, float had_scale,
    uint64_t* __restrict__ stamps) {
    using L = Layout<BITS>;
    using NT = Num<T>;
    constexpr int J = L::J;
    constexpr int NP = L::NPLANES;
    constexpr int LJ = (BITS == 4) ? 2 : (BITS == 2 ? 3 : 4);      // log2(J)
    constexpr int NSL = oneshot_scale_loads(BITS, D);
    constexpr int LUT = oneshot_lut_bytes(BITS);
    constexpr int ESTRIDE = (BITS == 3) ? 128 : 256;
    constexpr bool PIPE = (OPT & 16) != 0;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (lds_base_of(smem) != 0) __builtin_trap();                  // v_perm-built table addresses are absolute
#ifdef FLUTE_STAMPS
    uint64_t stamp[16];
    for (int i = 0; i < 16; ++i) stamp[i] = 0;
    stamp[0] = wall_clock64();                                    // 100 MHz, chip-wide: start skew / end
    stamp[1] = __builtin_amdgcn_s_memtime();                       // shader cycles: the phases of this wave
#define FLUTE_OSTAMP(i) stamp[i] = __builtin_amdgcn_s_memtime()
#else
#define FLUTE_OSTAMP(i)
#endif

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lg = geo & 15, lkw = (geo >> 4) & 15, upw = (geo >> 8) & 31, pk = (geo >> 13) & 15;
    const int ipw = (geo >> 17) & 127, had_log = (geo >> 24) & 15;
    const bool xh = (geo >> 28) & 1;                               // activations in the holes of the table image
    const int kw = 1 << lkw;
    const int nthr = (upw << lkw) * 64;                            // == blockDim.x (an implicit argument: a scalar load away)
    const int ul = wave >> lkw;
    const int kpart = wave & (kw - 1);
    const int units = N >> LJ;
    const int G = K >> lg;
    const int unit = blockIdx.x * upw + ul;
    const bool live = unit < units;
    const int urow = min(unit, units - 1);
    const int npieces = (K + 511) >> 9;
    const int p0 = kpart * pk;
    const int np = live ? max(0, min(pk, npieces - p0)) : 0;
    const int KX = npieces << 9;
    const uint32_t lane16 = (uint32_t)lane * 16u;
    const uint32_t row_bytes = (uint32_t)K * 2u;

    // LDS carve (oneshot_lds_bytes)
    const int gpp = 512 >> lg;                                     // groups per piece
    const int ngm = D * gpp;                                       // groups per wave image
    const uint32_t x_off = LUT;
    const uint32_t s_off = x_off + (xh ? 0u : (uint32_t)(MB * KX * 2));
    // byte address of the 16-B piece `pidx` (8 k each) of staged row m: linear [MB][KX] behind the image, or 64 k per hole
    auto x_addr = [&](int m, int pidx) -> uint32_t {
        return xh ? (uint32_t)((m * (KX >> 6) + (pidx >> 3)) * 256 + 128 + (pidx & 7) * 16)
                  : x_off + (uint32_t)(m * KX * 2 + pidx * 16);
    };
    const uint32_t s_wave_bytes = (uint32_t)(J * ngm * 2);
    const uint32_t sbase = s_off + (uint32_t)wave * s_wave_bytes;
    const uint32_t red_off = s_off + (uint32_t)(nthr >> 6) * s_wave_bytes;

    // ---- requests, oldest first: table word, activations, scale words, then the weights ----
    // table: wave w writes runs [w * ipw, min(RUNS, (w + 1) * ipw)) of the image; b = 4 / 3: a run is 8 entries
    // x 128 B and the wave's (<= 64) entries are loaded one per lane; b = 2: the 16 source words, one per lane
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
            const int k = pidx * 8;
            const uint32_t vo = (pidx < xrow_pieces && k < K) ? (uint32_t)(((size_t)min(m, M - 1) * K + k) * 2) : 0x80000000u;
            xv[m][r] = buf_load16(vo, x_srd, 0);
        }

    // scale words: lane q = lane + 64 r holds groups (2 gp, 2 gp + 1) of column j of this wave's K range,
    // q = j * (ngm / 2) + gp.  G is even (host), so every word is an aligned dword; words past the row / past the
    // wave's range read as zero.
    const int col0 = unit_col0<BITS, TILEP>(urow);
    const int g0 = (p0 * 512) >> lg;
    const int lgh = 31 - __builtin_clz((unsigned)ngm) - 1;          // log2(ngm / 2)
    const srd_t s_srd = make_srd(Sp, (uint32_t)min((size_t)N * G * 2, (size_t)0xfffffff0u));
    uint32_t sv[NSL];
    uint32_t s_keep[NSL];
#pragma unroll
    for (int r = 0; r < NSL; ++r) {
        const int q = lane + 64 * r;
        const int j = q >> lgh;
        const int gp = q & ((1 << lgh) - 1);
        const int gi = g0 + 2 * gp;
        const bool mine = live && j < J && gi < G && 2 * gp < np * gpp;
        s_keep[r] = !mine ? 0u : (gi + 1 < G ? 0xffffffffu : 0x0000ffffu);
        sv[r] = buf_load4_at(mine ? (uint32_t)(((size_t)(col0 + j * TILEP) * G + gi) * 2) : 0x80000000u, s_srd, 0);
    }

    // weights: piece i of plane pl -> q[i][pl]; positions in the VECTOR offset (the only offset the range check
    // covers): pieces past np and bytes past the end of a ragged row read as zero
    srd_t qsrd[NP];
#pragma unroll
    for (int pl = 0; pl < NP; ++pl)
        qsrd[pl] = make_srd(reinterpret_cast<const char*>(Qp) + (size_t)unit_row<BITS, TILEP>(urow, pl, N) * row_bytes,
                            live ? row_bytes : 0u);
    ring16_t q[D][NP];
    constexpr int NX = MB * XPR;
    constexpr int RUNS = oneshot_lut_runs(BITS);
    const int nrun = max(0, min(ipw, RUNS - run0));
    auto issue_piece = [&](auto i_tag) {
        constexpr int i = decltype(i_tag)::value;
        const uint32_t vo = lane16 + ((i < np) ? (uint32_t)(p0 + i) * 1024u : 0x80000000u);
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) {
            if constexpr (OPT & 1) q[i][pl] = buf_load16_nt(vo, qsrd[pl], 0);
            else q[i][pl] = buf_load16(vo, qsrd[pl], 0);
        }
    };
    // table image, one batch of (up to) four 1-KiB runs: the entry words come from the lanes that loaded them
    uint32_t tlo[4], thi[4];
    auto table_fetch = [&](int i0) {
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
    };
    auto table_write = [&](int i0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (i0 + u < nrun) {
                uint32_t addr;
                if constexpr (BITS == 2) addr = (uint32_t)(run0 + i0 + u) * 1024u + lane16;
                else addr = (uint32_t)((run0 + i0 + u) * 8 + (lane >> 3)) * ESTRIDE + (uint32_t)(lane & 7) * 16u;
                *reinterpret_cast<uint4*>(smem + addr) = make_uint4(tlo[u], thi[u], tlo[u], thi[u]);
            }
        }
    };
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (OPT & 32) {
        // Interleaved prologue: the weight requests (each stalls the wave ~90 cycles in the texture addresser's
        // queue) alternate with the table image's LDS work, so that the 32-KB image drains into LDS while the
        // requests are still going out - the image and the barrier behind it were 1.1 us of a wave's 3.4.
        constexpr int NB = (BITS == 2) ? 4 : 2;                    // table batches a wave may own (4 runs each)
        constexpr int QPS = (D + 2 * NB - 1) / (2 * NB);           // pieces issued per table step
        static_for<D>([&](auto i_tag) { if constexpr (decltype(i_tag)::value < QPS) issue_piece(i_tag); });
        vm_wait_regs<NX + NSL + (QPS < D ? QPS : D) * NP>(lut_v);
        FLUTE_OSTAMP(3);
        static_for<2 * NB>([&](auto step_tag) {
            constexpr int STEP = decltype(step_tag)::value;
            constexpr int b = STEP / 2;
            if (b * 4 < nrun) { if constexpr (STEP % 2 == 0) table_fetch(b * 4); else table_write(b * 4); }
            static_for<D>([&](auto i_tag) {
                constexpr int I = decltype(i_tag)::value;
                if constexpr (I >= (STEP + 1) * QPS && I < (STEP + 2) * QPS) issue_piece(i_tag);
            });
        });
        FLUTE_OSTAMP(2);
    } else {
        static_for<D>([&](auto i_tag) { issue_piece(i_tag); });
        __builtin_amdgcn_sched_barrier(0);
        FLUTE_OSTAMP(2);
        vm_wait_regs<NX + NSL + D * NP>(lut_v);
        FLUTE_OSTAMP(3);
        for (int i0 = 0; i0 < nrun; i0 += 4) { table_fetch(i0); table_write(i0); }
    }
    __builtin_amdgcn_sched_barrier(0);
    int* arrive = reinterpret_cast<int*>(smem + red_off);
    if (kw > 1 && tid < upw) arrive[tid] = 0;
    FLUTE_OSTAMP(4);

    // ---- activations -> LDS [MB][KX] (zero beyond K); fused pre-rotation: the 64 pieces a wave stages are 512
    // consecutive k of one row = whole Hadamard blocks (K % had == 0, had <= 512; qgemm.cpp:201-244) ----
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int r = 0; r < XPR; ++r) vm_wait_regs<NSL + D * NP>(xv[m][r]);
    FLUTE_OSTAMP(5);
#pragma unroll
    for (int m = 0; m < MB; ++m) {
#pragma unroll
        for (int r = 0; r < XPR; ++r) {
            const int pidx = r * nthr + tid;
            if (pidx < xrow_pieces) {                              // wave-uniform: rows are whole 512-k spans
                uint32_t w[4] = {xv[m][r].x, xv[m][r].y, xv[m][r].z, xv[m][r].w};
                if constexpr (HAD) fwht_piece<T>(w, lane, had_log, had_scale);
                const bool inside = pidx * 8 < K;
                *reinterpret_cast<uint4*>(smem + x_addr(m, pidx)) = inside ? make_uint4(w[0], w[1], w[2], w[3]) : make_uint4(0, 0, 0, 0);
            }
        }
        const uint16_t* A = reinterpret_cast<const uint16_t*>(Ap);
        for (int pidx = XPR * nthr + tid; pidx < xrow_pieces; pidx += nthr) {     // rows longer than XPR x threads pieces
            const int k = min(pidx * 8, K - 8);
            const uint4 t = *reinterpret_cast<const uint4*>(A + (size_t)min(m, M - 1) * K + k);
            uint32_t w[4] = {t.x, t.y, t.z, t.w};
            if constexpr (HAD) fwht_piece<T>(w, lane, had_log, had_scale);
            const bool inside = pidx * 8 < K;
            *reinterpret_cast<uint4*>(smem + x_addr(m, pidx)) = inside ? make_uint4(w[0], w[1], w[2], w[3]) : make_uint4(0, 0, 0, 0);
        }
    }
    FLUTE_OSTAMP(6);
    __syncthreads();                                               // table / activations visible to every wave
    FLUTE_OSTAMP(7);

    // ---- scale image [group][column] (wave-private: no barrier) ----
#pragma unroll
    for (int r = 0; r < NSL; ++r) {
        vm_wait_regs<D * NP>(sv[r]);
        const int qq = lane + 64 * r;
        const int j = qq >> lgh;
        const int gp = qq & ((1 << lgh) - 1);
        if (j < J) {
            const uint32_t w = sv[r] & s_keep[r];
            uint16_t* img = reinterpret_cast<uint16_t*>(smem + sbase) + (size_t)(2 * gp) * J + j;
            img[0] = (uint16_t)(w & 0xffffu);
            img[J] = (uint16_t)(w >> 16);
        }
    }
    FLUTE_OSTAMP(8);

    // per-lane constants of the piece loop
    const uint32_t lane_off = (BITS == 2) ? (uint32_t)(lane & 31) * 8u : (uint32_t)(lane & 31) * 4u;
    const int gl = (8 * lane) >> lg;                               // group of the lane's 8 k inside a piece
    const uint32_t s_lane = sbase + (uint32_t)(gl * J) * 2u;
    const uint32_t x_lane = x_addr(0, p0 * 64 + lane);             // this lane's 16 B of the wave's first piece (row 0)
    const uint32_t x_pshift = xh ? 11u : 10u;                      // a piece (64 lanes x 16 B) spans 2 KiB of holes / 1 KiB
    const uint32_t x_row = xh ? (uint32_t)(KX >> 6) * 256u : (uint32_t)KX * 2u;

    float acc[J][MB];
#pragma unroll
    for (int j = 0; j < J; ++j)
#pragma unroll
        for (int m = 0; m < MB; ++m) acc[j][m] = 0.f;

    auto compute_piece = [&](auto slot_tag) {
        constexpr int i = decltype(slot_tag)::value;
        const uint32_t xa = x_lane + ((uint32_t)i << x_pshift);
        const uint32_t sa = s_lane + (uint32_t)(i * gpp * J) * 2u;
        uint32_t xw[MB][4];
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            const uint4 t = lds_ld128(xa + (uint32_t)m * x_row);
            xw[m][0] = t.x; xw[m][1] = t.y; xw[m][2] = t.z; xw[m][3] = t.w;
        }
        uint32_t scw[J / 2];                                       // J scales: one vector read (J = 4: 8 B)
        if constexpr (J == 4) { const uint2 t = lds_ld64(sa); scw[0] = t.x; scw[1] = t.y; }
        else {
#pragma unroll
            for (int c = 0; c < J / 8; ++c) {
                const uint4 t = lds_ld128(sa + 16u * c);
                scw[4 * c] = t.x; scw[4 * c + 1] = t.y; scw[4 * c + 2] = t.z; scw[4 * c + 3] = t.w;
            }
        }
        float al[J][MB];
#pragma unroll
        for (int j = 0; j < J; ++j)
#pragma unroll
            for (int m = 0; m < MB; ++m) al[j][m] = 0.f;
        if constexpr (OPT & 2) {
#pragma unroll
            for (int pl = 0; pl < NP; ++pl)
#pragma unroll
                for (int ww = 0; ww < 4; ++ww)
#pragma unroll
                    for (int m = 0; m < MB; ++m) al[ww % J][m] += __builtin_bit_cast(float, q[i][pl][ww] ^ xw[m][ww]);
        } else if constexpr (BITS == 2) {
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
            const float sf = scale_to_float<T>((j & 1) ? (scw[j / 2] >> 16) : (scw[j / 2] & 0xffffu));
#pragma unroll
            for (int m = 0; m < MB; ++m) acc[j][m] = __builtin_fmaf(al[j][m], sf, acc[j][m]);
        }
    };

    if constexpr (PIPE) {
        // Software-pipelined loop over groups of 8 lookups (pipelined_pieces).  The host launches this variant only when
        // EVERY wave of the grid holds exactly D pieces (api.hip: plan_oneshot; a run-time fallback to the plain loop
        // would put two sets of counted waits on the same in-flight registers behind a branch, the arrangement hipcc
        // mis-schedules - measured: wrong results).
        pipelined_pieces<T, BITS, MB, D, 0, (BITS != 2) ? 1 : 0>(q, x_lane, x_pshift, x_row, s_lane, (uint32_t)(gpp * J) * 2u, lane_off, acc);
    } else {
        // ---- pieces, each released by its own counted wait (loads return in order) ----
        [&]<int... I>(std::integer_sequence<int, I...>) {
            ([&] {
                constexpr int YOUNGER = (OPT & 8) ? 0 : (D - 1 - I) * NP;
                if constexpr (NP == 1) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(q[I][0]) : "n"(YOUNGER) : "memory");
                else asm volatile("s_waitcnt vmcnt(%3)" : "+v"(q[I][0]), "+v"(q[I][1]), "+v"(q[I][2]) : "n"(YOUNGER) : "memory");
                if (I < np) compute_piece(std::integral_constant<int, I>{});
            }(), ...);
        }(std::make_integer_sequence<int, D>{});
    }
    FLUTE_OSTAMP(9);

    // ---- lanes -> wave (DPP) -> [K split: waves -> LDS -> last arriver] -> output ----
    uint16_t* Dout = reinterpret_cast<uint16_t*>(Dp);
    auto store_out = [&](int j, int m, float v) {
        if (m < M && live) Dout[(size_t)m * N + col0 + j * TILEP] = NT::from_float(v);
    };
    constexpr bool TR = (BITS == 4) || (BITS == 3 && MB <= 2);      // round 5: the unit's J x MB sums reduced together (transpose_reduce4 / 16)
    if constexpr (TR) {
        float v;
        int my_m;
        bool holder;
        if constexpr (BITS == 4) transpose_reduce4<MB>(acc, lane, v, my_m, holder);
        else transpose_reduce16<(MB <= 2 ? MB : 1)>(reinterpret_cast<float (&)[16][MB <= 2 ? MB : 1]>(acc), lane, v, my_m, holder);
        FLUTE_OSTAMP(10);
        const int jh = lane & (J - 1);
        if (kw == 1) {
            if (holder) store_out(jh, my_m, v);                     // one store instruction per unit
        } else {
            float* rb = reinterpret_cast<float*>(smem + red_off) + 32;
            if (holder) rb[wave * (J * MB) + jh * MB + my_m] = v;
            int ticket = 0;
            if (lane == 0) ticket = __hip_atomic_fetch_add(&arrive[ul], 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
            ticket = __builtin_amdgcn_readfirstlane(ticket);
            if (ticket == kw - 1) {
                for (int t = lane; t < J * MB; t += 64) {
                    float sum = 0.f;
                    for (int kp = 0; kp < kw; ++kp) sum += rb[(ul * kw + kp) * (J * MB) + t];
                    store_out(t / MB, t % MB, sum);
                }
            }
        }
    } else {
        float tot[J][MB];
#pragma unroll
        for (int j = 0; j < J; ++j)
#pragma unroll
            for (int m = 0; m < MB; ++m) tot[j][m] = wave_sum64(acc[j][m]);
        FLUTE_OSTAMP(10);
        if (kw == 1) {
            if (lane == 0) {
#pragma unroll
                for (int m = 0; m < MB; ++m)
#pragma unroll
                    for (int j = 0; j < J; ++j) store_out(j, m, tot[j][m]);
            }
        } else {
            // no barrier: every wave leaves its partial sums and an arrival tick in LDS, the last arriver sums and
            // stores (release on the tick / acquire by the reader: the partials are ordered before it)
            float* rb = reinterpret_cast<float*>(smem + red_off) + 32;
            if (lane == 0) {
#pragma unroll
                for (int j = 0; j < J; ++j)
#pragma unroll
                    for (int m = 0; m < MB; ++m) rb[wave * (J * MB) + j * MB + m] = tot[j][m];
            }
            int ticket = 0;
            if (lane == 0) ticket = __hip_atomic_fetch_add(&arrive[ul], 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
            ticket = __builtin_amdgcn_readfirstlane(ticket);
            if (ticket == kw - 1) {
                for (int t = lane; t < J * MB; t += 64) {
                    float sum = 0.f;
                    for (int kp = 0; kp < kw; ++kp) sum += rb[(ul * kw + kp) * (J * MB) + t];
                    store_out(t / MB, t % MB, sum);
                }
            }
        }
    }
#ifdef FLUTE_STAMPS
    FLUTE_OSTAMP(11);
    __builtin_amdgcn_s_waitcnt(0);
    stamp[12] = __builtin_amdgcn_s_memtime();
    stamp[13] = wall_clock64();
    if (lane == 0 && stamps != nullptr) {
        uint64_t* o = stamps + ((size_t)blockIdx.x * (nthr >> 6) + wave) * 16;
        for (int i = 0; i < 16; ++i) o[i] = stamp[i];
    }
#endif
#undef FLUTE_OSTAMP
}

}  // namespace flute_amd
