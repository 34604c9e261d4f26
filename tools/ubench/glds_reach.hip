// Does LDS-DMA (global_load_lds_dwordx4, M0 = destination) reach LDS addresses above 64 KB on gfx950,
// and is the data readable by the issuing wave right after its own s_waitcnt vmcnt(0) (no barrier)?
// build: hipcc --offload-arch=gfx950 -O3 -o glds_reach glds_reach.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ void glds16(const void* g, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(lds_addr) : "memory");
}

__global__ __launch_bounds__(512) void k(const unsigned* src, unsigned* out, unsigned base) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned dst = base + wave * 2048;                 // two pieces per wave
    // lane-permuted source (what a pre-swizzled load does): lane L fetches piece word group L^5
    glds16(src + ((size_t)wave * 128 + (lane ^ 5)) * 4, dst);
    glds16(src + ((size_t)wave * 128 + 64 + lane) * 4, dst + 1024);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    typedef unsigned u4v __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) const u4v lds_u4;
    const u4v a = *(lds_u4*)(uintptr_t)(dst + lane * 16);
    const u4v b = *(lds_u4*)(uintptr_t)(dst + 1024 + lane * 16);
    unsigned* o = out + ((size_t)blockIdx.x * 512 + threadIdx.x) * 8;
    o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
}

int main() {
    const int n = 8 * 128 * 4;
    std::vector<unsigned> h(n);
    for (int i = 0; i < n; ++i) h[i] = 0x1000000u + i;
    unsigned *src, *out;
    hipMalloc(&src, n * 4); hipMemcpy(src, h.data(), n * 4, hipMemcpyHostToDevice);
    hipMalloc(&out, 512 * 8 * 4);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    std::vector<unsigned> r(512 * 8);
    for (unsigned base : {0u, 16384u, 49152u, 65536u, 66560u, 100000u / 16 * 16, 147456u - 1024}) {
        hipMemset(out, 0, 512 * 8 * 4);
        hipLaunchKernelGGL(k, 1, 512, 160 * 1024, 0, src, out, base);
        hipError_t e = hipDeviceSynchronize();
        hipMemcpy(r.data(), out, 512 * 8 * 4, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int t = 0; t < 512; ++t) {
            const int wave = t >> 6, lane = t & 63;
            for (int j = 0; j < 4; ++j) {
                if (r[t * 8 + j] != h[(wave * 128 + (lane ^ 5)) * 4 + j]) ++bad;
                if (r[t * 8 + 4 + j] != h[(wave * 128 + 64 + lane) * 4 + j]) ++bad;
            }
        }
        printf("{\"lds_base\": %u, \"err\": \"%s\", \"mismatches\": %d}\n", base, hipGetErrorString(e), bad);
    }
    return 0;
}
