// Kernel lookup tables filled by the per-bit-width instantiation units.
#pragma once
#include "common.h"

namespace flute_amd {

typedef void (*QGemmKernel)(const QGemmArgs);

// streaming decode kernel (qgemm_stream.h): mb rows per pass (1/2/4; b=3: 1/2), depth = ring slots (2/4), one_shot = the no-refill variant (b=2/4: depth 4, b=3: depth 2)
struct StreamArgs;
typedef void (*StreamKernel)(const StreamArgs);
StreamKernel stream_kernel_b4(int dtype, int tile_p, int mb, int depth, int one_shot);
StreamKernel stream_kernel_b3(int dtype, int tile_p, int mb, int depth, int one_shot);
StreamKernel stream_kernel_b2(int dtype, int tile_p, int mb, int depth, int one_shot);
// one-shot decode kernel (qgemm_oneshot.h): mb rows per pass (1/2/4; b=3: 1/2), depth = pieces per wave (4/8; b=3: 2/4), had = fused
// Hadamard pre-rotation, pipe = software-pipelined lookup groups (every wave of the launch holds `depth` whole pieces)
typedef void (*OneKernel)(const uint32_t*, const void*, const void*, const uint32_t*, int, int, uint32_t, int, void*, float, uint64_t*);
OneKernel oneshot_kernel_b4_f16(int tile_p, int mb, int depth, int had, int pipe);
OneKernel oneshot_kernel_b4_bf16(int tile_p, int mb, int depth, int had, int pipe);
OneKernel oneshot_kernel_b2_f16(int tile_p, int mb, int depth, int had, int pipe);
OneKernel oneshot_kernel_b2_bf16(int tile_p, int mb, int depth, int had, int pipe);
OneKernel oneshot_kernel_b3(int dtype, int tile_p, int mb, int depth, int had, int pipe);
// lean decode kernel (qgemm_fast.h): 4 bits, K = 512 * depth * kw; waves per workgroup, waves per unit row, pieces per wave, rows per pass (1/2/4)
typedef void (*FastKernel)(const uint32_t*, const void*, const void*, const uint32_t*, void*, int, int, int, uint64_t*);
FastKernel fast_kernel_b4(int dtype, int tile_p, int waves, int kw, int depth, int mb);
// lean MFMA decode kernel (qgemm_fastm.h): 4 bits, M <= 16, a workgroup = 4 unit rows x all of K = 128 * nm * waves; lg = log2(group size)
typedef void (*FastMKernel)(const uint32_t*, const void*, const void*, const uint32_t*, void*, int, int, uint64_t*);
FastMKernel fastm_kernel_b4(int dtype, int tile_p, int waves, int nm, int lg, int ng);   // ng: column groups (4 unit rows each) per workgroup, 1 .. 3
// persistent MFMA decode kernel (qgemm_persistm.h): 4 / 2 bits, M <= 16, K % 128 == 0; lg = log2(group size) (6 / 7), ng: column groups per set (1 .. 3),
// xr: activation requests per macro-step (1, 2, 4: M <= 4 xr)
typedef void (*PersistMKernel)(const uint32_t*, const void*, const void*, const uint32_t*, void*, int, int, int, int);
PersistMKernel persistm_kernel_b4_f16(int tile_p, int lg, int ng, int xr, int waves, int xres);    // waves: 8; xres: activations resident in LDS (K * xr <= 8192; (ng, xr) in (1..2, 1), (1..3, 2))
PersistMKernel persistm_kernel_b4_bf16(int tile_p, int lg, int ng, int xr, int waves, int xres);
PersistMKernel persistm_kernel_b2_f16(int tile_p, int lg, int ng, int xr, int waves, int xres);     // 2-bit member (a group = two unit rows of eight columns)
PersistMKernel persistm_kernel_b2_bf16(int tile_p, int lg, int ng, int xr, int waves, int xres);
// persistent one-shot decode kernel (qgemm_persist.h): mb rows per pass (1/2), depth = pieces per segment, nsets = register sets
typedef void (*PersistKernel)(const uint32_t*, const void*, const void*, const uint32_t*, int, int, uint32_t, int, void*, float, int);
PersistKernel persist_kernel_b4(int dtype, int tile_p, int mb, int depth, int nsets, int had);
PersistKernel persist_kernel_b2(int dtype, int tile_p, int mb, int depth, int nsets, int had);
PersistKernel persist_kernel_b3(int dtype, int tile_p, int mb, int depth, int nsets, int had);
// skinny MFMA kernel (qgemm_skinny.h): 4-bit, M <= 16, depth = k-steps per wave (4/8/16)
typedef void (*SkinnyKernel)(const uint32_t*, const void*, const void*, const uint32_t*, int, int, uint32_t, int, void*, uint64_t*, float*, uint32_t*);
SkinnyKernel skinny_kernel_b4(int dtype, int tile_p, int depth);
// block-tiled prefill kernels (qgemm_block2.h / qgemm_block3.h): cfg 4 = 256 x 256 block, cfg 5 = 128 x 256, 8 + RT = skinny 3-bit blocks
struct BlockArgs;
typedef void (*BlockKernel)(const BlockArgs);
BlockKernel block_kernel_b4(int dtype, int tile_p, int cfg);
BlockKernel block_kernel_b2(int dtype, int tile_p, int cfg);
BlockKernel block_kernel_b3(int dtype, int tile_p, int cfg);
// split-K block kernel (qgemm_splitk.h): 128 x 128 tiles, K split over workgroups, combined in the launch (xwg.h)
struct SplitKArgs;
typedef void (*SplitKKernel)(const SplitKArgs);
SplitKKernel splitk_kernel(int bits, int dtype, int tile_p, int rt, int kp);   // rt: row tiles per workgroup (8 / 4), kp: K parts per workgroup (2; 4 with rt 4)
// MFMA kernel (qgemm_tile.h): r lanes share one unit's words (1, 2, 4; b=3: 1), mt 16-row tiles per wave
QGemmKernel tile_kernel_b4(int dtype, int tile_p, int r, int mt, int sw);   // sw: slabs per wave (1, 2)
QGemmKernel tile_kernel_b3(int dtype, int tile_p, int r, int mt);
QGemmKernel tile_kernel_b2(int dtype, int tile_p, int r, int mt);

int hadamard_dispatch(int dtype, const void* in, void* out, size_t numel, uint32_t h,
                      hipStream_t stream);
int unpack_dispatch(int num_bits, int tile_p, int N, int K, const void* Q, void* W,
                    hipStream_t stream);
int stream_read_dispatch(const void* src, void* sink, size_t bytes, int bytes_per_wave, int grid,
                         int block, hipStream_t stream);
int timestamp_dispatch(void* dst, hipStream_t stream);
int splitk_reduce_dispatch(int dtype, const float* partial, void* D, size_t mn, int splitk,
                           hipStream_t stream);

}  // namespace flute_amd
