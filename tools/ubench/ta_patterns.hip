// Microbenchmark: how fast does one CU take 16-B-per-lane global loads (L2-resident data) as a
// function of the lane -> address pattern?  Motivates the operand staging of the MFMA kernel.
//   pattern 0: MFMA-operand order   lane -> (row = lane%16, 16-B chunk = lane/16), rows 8 KB apart
//   pattern 1: quad-coalesced       lane -> (row = lane/4,  chunk = lane%4)   same 16 rows x 64 B
//   pattern 2: full lines           lane -> (row = lane/8,  chunk = lane%8)   8 rows x 128 B
//   pattern 3: contiguous           lane -> 16*lane                           1 KB
// build: hipcc --offload-arch=gfx950 -O3 -o ta_patterns ta_patterns.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int PAT>
__global__ __launch_bounds__(512) void k(const char* base, size_t row_stride, int steps, int iters, unsigned* sink,
                                         unsigned long long* clk) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // the activation matrix of a prefill call: 256 rows x 8 KB (L2-resident, shared by all
    // workgroups); workgroup -> 64-row tile, wave -> its 1 KB k-range of those rows
    const size_t tile_row0 = (size_t)(blockIdx.x & 3) * 64;
    const size_t kbase = (size_t)wave * 1024;
    size_t off, row16;            // row16: byte distance between the wave's four 16-row fragments
    if (PAT == 0) { off = (tile_row0 + (lane & 15)) * row_stride + kbase + (size_t)(lane >> 4) * 16; row16 = 16 * row_stride; }
    else if (PAT == 1) { off = (tile_row0 + (lane >> 2)) * row_stride + kbase + (size_t)(lane & 3) * 16; row16 = 16 * row_stride; }
    else if (PAT == 2) { off = (tile_row0 + (lane >> 3)) * row_stride + kbase + (size_t)(lane & 7) * 16; row16 = 8 * row_stride; }
    else { off = ((size_t)(blockIdx.x & 3) * 8 + wave) * 65536 + (size_t)lane * 16; row16 = 1024; }
    // one k-step = 4 loads (64 rows x 64 B); pattern 2 covers 32 rows x 128 B per 4 loads
    const size_t step_bytes = (PAT == 2) ? 128 : (PAT == 3 ? 4096 : 64);
    const int nst = steps;
    u32x4 acc = {0, 0, 0, 0};
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        const char* p = base + off;
        for (int s = 0; s < nst; s += 2) {
            u32x4 a, b, c, d, e, f, g, h;
            const char* q = p + ((PAT == 2) ? 4 * row16 : step_bytes);     // second k-step / second 32 rows
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(a) : "v"(p) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(b) : "v"(p + row16) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(c) : "v"(p + 2 * row16) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(d) : "v"(p + 3 * row16) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(e) : "v"(q) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(f) : "v"(q + row16) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(g) : "v"(q + 2 * row16) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(h) : "v"(q + 3 * row16) : "memory");
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h)::"memory");
            acc ^= a ^ b ^ c ^ d ^ e ^ f ^ g ^ h;
            p += (PAT == 2) ? 128 : 2 * step_bytes;
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) sink[0] = 1;
    if (lane == 0) clk[blockIdx.x * 8 + wave] = t1 - t0;
}

__device__ __forceinline__ void glds16(const void* g, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(lds_addr) : "memory");
}

// same traffic through LDS-DMA (global_load_lds_dwordx4) into a wave-private 8 KB LDS ring
template <int PAT>
__global__ __launch_bounds__(512) void kdma(const char* base, size_t row_stride, int steps, int iters, unsigned* sink,
                                            unsigned long long* clk) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t tile_row0 = (size_t)(blockIdx.x & 3) * 64;
    const size_t kbase = (size_t)wave * 1024;
    size_t off, row16;
    if (PAT == 0) { off = (tile_row0 + (lane & 15)) * row_stride + kbase + (size_t)(lane >> 4) * 16; row16 = 16 * row_stride; }
    else if (PAT == 1) { off = (tile_row0 + (lane >> 2)) * row_stride + kbase + (size_t)(lane & 3) * 16; row16 = 16 * row_stride; }
    else { off = (tile_row0 + (lane >> 3)) * row_stride + kbase + (size_t)(lane & 7) * 16; row16 = 8 * row_stride; }
    const size_t step_bytes = (PAT == 2) ? 128 : 64;
    const unsigned ring = wave * 8192;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        const char* p = base + off;
        for (int s = 0; s < steps; s += 2) {
            const char* q = p + ((PAT == 2) ? 4 * row16 : step_bytes);
            glds16(p, ring); glds16(p + row16, ring + 1024); glds16(p + 2 * row16, ring + 2048); glds16(p + 3 * row16, ring + 3072);
            glds16(q, ring + 4096); glds16(q + row16, ring + 5120); glds16(q + 2 * row16, ring + 6144); glds16(q + 3 * row16, ring + 7168);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            p += (PAT == 2) ? 128 : 2 * step_bytes;
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    if (*(volatile unsigned*)(smem + ring + lane * 4) == 0x12345u) sink[0] = 1;
    if (lane == 0) clk[blockIdx.x * 8 + wave] = t1 - t0;
}

int main() {
    const int grid = 256, steps = 16 /* k-steps of 32 halves */, iters = 50;
    const size_t row_stride = 8192;
    const size_t rows = 256;
    char* buf; unsigned* sink; unsigned long long* clk;
    hipMalloc(&buf, rows * row_stride + (1 << 20)); hipMemset(buf, 1, rows * row_stride + (1 << 20));
    hipMalloc(&sink, 4); hipMalloc(&clk, grid * 8 * 8);
    std::vector<unsigned long long> h(grid * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int pat = 0; pat < 4; ++pat) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            size_t rs = row_stride;
            switch (pat) {
                case 0: hipLaunchKernelGGL(k<0>, grid, 512, 0, 0, buf, rs, steps, iters, sink, clk); break;
                case 1: hipLaunchKernelGGL(k<1>, grid, 512, 0, 0, buf, rs, steps, iters, sink, clk); break;
                case 2: hipLaunchKernelGGL(k<2>, grid, 512, 0, 0, buf, rs, steps, iters, sink, clk); break;
                default: hipLaunchKernelGGL(k<3>, grid, 512, 0, 0, buf, rs, steps, iters, sink, clk); break;
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(h.data(), clk, grid * 8 * 8, hipMemcpyDeviceToHost);
            double avg = 0; for (auto v : h) avg += (double)v; avg /= h.size();
            const double bytes_cu = 8.0 * 64 * steps * 64 * iters;          // per CU: 8 waves x 64 rows x 64 B per k-step
            const double instr_cu = bytes_cu / 1024;
            printf("{\"pattern\": %d, \"rep\": %d, \"ms\": %.4f, \"wave_cycles\": %.0f, \"cycles_per_wave_instr_per_CU\": %.1f, "
                   "\"B_per_clk_per_CU\": %.1f, \"agg_TBps\": %.2f}\n", pat, rep, ms, avg, avg / instr_cu,
                   bytes_cu / avg, bytes_cu * grid / (ms * 1e-3) / 1e12);
        }
    }
    hipFuncSetAttribute((const void*)kdma<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute((const void*)kdma<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute((const void*)kdma<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int pat = 0; pat < 3; ++pat) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            switch (pat) {
                case 0: hipLaunchKernelGGL(kdma<0>, grid, 512, 65536, 0, buf, row_stride, steps, iters, sink, clk); break;
                case 1: hipLaunchKernelGGL(kdma<1>, grid, 512, 65536, 0, buf, row_stride, steps, iters, sink, clk); break;
                default: hipLaunchKernelGGL(kdma<2>, grid, 512, 65536, 0, buf, row_stride, steps, iters, sink, clk); break;
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(h.data(), clk, grid * 8 * 8, hipMemcpyDeviceToHost);
            double avg = 0; for (auto v : h) avg += (double)v; avg /= h.size();
            const double bytes_cu = 8.0 * 64 * steps * 64 * iters;
            const double instr_cu = bytes_cu / 1024;
            printf("{\"lds_dma_pattern\": %d, \"rep\": %d, \"ms\": %.4f, \"wave_cycles\": %.0f, \"cycles_per_wave_instr_per_CU\": %.1f, "
                   "\"B_per_clk_per_CU\": %.1f, \"agg_TBps\": %.2f}\n", pat, rep, ms, avg, avg / instr_cu,
                   bytes_cu / avg, bytes_cu * grid / (ms * 1e-3) / 1e12);
        }
    }
    return 0;
}
