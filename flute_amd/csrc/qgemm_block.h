// Shared definitions of the block-tiled prefill kernels (qgemm_block2.h: 4- / 2-bit, qgemm_block3.h: 3-bit).  The
// first block kernel of round 2 (a 2 x 4 wave split of the 256 x 256 block, every weight dequantised by two waves)
// lived here; the 1 x 8 split of qgemm_block2.h superseded it and it was removed in round 3.  What follows describes
// the scheme the three kernels share.
//
// Prefill-regime kernel (M >= ~512): block-tiled LUT-dequant GEMM on MFMA.
//
// gfx950 replacement for qgemm_device's main loop at large M (flute/csrc/qgemm_kernel.hpp:617-712,
// config.hpp:187-558).  The round-1 MFMA kernel (qgemm_tile.h) gives every wave a private 64-row output tile
// with a private K range: no operand is shared between waves, and each weight is LUT-dequantised once per
// 64 rows (64x at M = 4096) - 31 % of the MFMA peak at M = 4096.  Here a WORKGROUP owns a
// (WM x TM x 16) x (WN x 16 units) output block (256 x 256 for 4-bit weights) over the whole K range:
//   * activations: one [BM][64 k] tile per K step, fetched ONCE per workgroup by LDS-DMA
//     (`buffer_load_dwordx4 ... lds`: descriptor + scalar K offset, no per-step VALU address arithmetic, rows
//     beyond M read as zero), three stages in LDS, ONE s_barrier per K step, shared by the WN waves of a row
//     group; the MFMA fragments come out of LDS with conflict-free ds_read_b128 (source-side XOR swizzle);
//   * weights: every wave dequantises the 16 units x J fields (64 columns for 4 bits) it multiplies, straight
//     from packed words in registers (hidden buffer loads, 3-step ring): 16 lookups + 16 v_pk_mul_f16 per
//     32-k half step feed TM x J MFMAs - one lookup per TM MFMA rows instead of one per 4 (TM = 8: a weight is
//     dequantised once per 128 rows, WM = 2 times per workgroup);
//   * scales: 8-group blocks per wave, LDS-DMA through the same in-order queue into a wave-private image;
//   * every hidden load is unconditional, every step issues the same number of them (padding steps read out
//     of range = zeros), every counted wait is one statement (tools/audit_asm_loads.py).
// Arithmetic contract: w^ = round_T(lut * s) (packbits_utils.hpp:139) - v_pk_mul_f16 for fp16, fp32 multiply +
// v_cvt_pk_bf16_f32 for bf16 - fp32 accumulation in the MFMA (config.hpp:323-325), one rounding of the output.
#pragma once
#include <utility>

#include "common.h"
#include "mfma.h"
#include "qgemm_stream.h"      // srd_t, make_srd, buf_load16, hidden LDS lookups

namespace flute_amd {

struct BlockArgs {
    const void* A;          // [M,K] T
    const uint32_t* Q;      // [P,K/2] packed
    void* D;                // [M,N] T
    const void* S;          // [N,G] T
    const uint32_t* QM2;    // [4^b] pair table
    float* partial;         // splitk > 1: fp32 slabs - [splitk][M][N] for the reduce launch, or (state != nullptr) in fragment order (xwg.h)
    uint32_t* state;        // in-launch combine (xwg.h, round 5: qgemm_block3.h): two zero words per output tile; nullptr = reduce launch
    int M, N, K, G, lg;
    int tiles_m, tiles_n;   // workgroup tiles
    int splitk, k_per_split;    // k_per_split: multiple of lcm(64, 8 * group_size)
    int order;              // 0: M tiles fastest; 1: XCD x owns a contiguous range of M tiles; 2: of N tiles
};

constexpr int BLK_STAGES = 3;

// 16-B chunk swizzle of a 16-row x 64-B activation piece (an involution applied to the DMA source address and
// to the fragment read): the 16 lanes of every ds_read_b128 lane group then hit 16 different bank slots
__device__ __forceinline__ int blk_swz(int row) { return (4 - (row >> 2)) & 3; }

// position swizzle of an 8-row x 8-chunk activation piece (rows 8 rh .. 8 rh + 7 of a 16-row tile, 128 B = one 64-k step
// per row - whole cache lines, qgemm_block2.h since round 4): LDS position pos of row8 holds chunk pos ^ blk_swz8; the 16
// lanes of every ds_read_b128 lane group then hit 16 different 16-B slots of the 256-B bank row (tests/test_splitk_layout.py)
__host__ __device__ constexpr int blk_swz8(int row8, int rh) { return (row8 >> 1) | (rh << 2); }

__host__ __device__ constexpr int block_lds_bytes(int bits, int tm, int wm, int wn) {
    const int lut = (1 << (2 * bits)) * 128;
    const int stage = wm * tm * 2 * 1024;                      // BM/16 row tiles x 2 half steps x 1 KB
    return lut + BLK_STAGES * stage + wm * wn * 3 * 1024 * (bits == 2 ? 2 : 1);     // + per wave: two scale blocks + a sink
}

// LDS-DMA through a buffer descriptor: 16 B per lane from base + voff (per lane, range-checked) + soff
// (wave-uniform) to LDS byte m0 + 16 * lane.  No VGPR destination: only the counted vmcnt orders it.
__device__ __forceinline__ void dma16_buf(uint32_t voff, srd_t srd, uint32_t soff, uint32_t lds_addr) {
    uint32_t keep;                                                 // M0 is compiler-reserved: saved and restored
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(srd), "s"(soff), "s"(lds_addr) : "memory");
}

// releases 8 hidden lookups once at most N younger LDS operations are outstanding
template <int N> __device__ __forceinline__ void blk_lookup_wait(uint32_t (&v)[8]) {
    asm volatile("s_waitcnt lgkmcnt(%8)"
                 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7])
                 : "n"(N) : "memory");
}

}  // namespace flute_amd
