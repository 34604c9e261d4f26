// Stand-alone lab for the latency-bound decode launch (development tool, not shipped): checks variants of
// qgemv_oneshot_kernel against a CPU restatement on the headline shape and times them HBM-cold (a hipGraph of
// one launch per weight copy, copies exceeding the 256 MiB Infinity Cache), next to the round-2 one-shot
// kernel, a pure read of the same bytes in the same geometry, and an empty kernel (the launch boundary).
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 -mllvm -amdgpu-kernarg-preload-count=12 -Iflute_amd/csrc \
//         tools/ubench/oneshot_lab.hip -o tools/ubench/oneshot_lab
// Prints one JSON object per line.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <functional>
#include <string>
#include <vector>

#include "qgemm_oneshot.h"
#include "qgemm_fast.h"

using namespace flute_amd;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("{\"error\": \"%s at %s:%d\"}\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (uint32_t)(rng_state >> 16); }
static float frand() { return (float)(rnd() & 0xffffff) / 16777216.0f; }
static float nrand() { float s = 0; for (int i = 0; i < 12; ++i) s += frand(); return s - 6.0f; }

__global__ void empty_kernel(const uint32_t* p, int n) { if (n < 0) ((volatile uint32_t*)p)[0] = 1; }

__global__ __launch_bounds__(1024) void ramp_kernel(uint64_t* out) {
    extern __shared__ char smem[];
    const uint64_t t = wall_clock64();
    if ((threadIdx.x & 63) == 0) out[(size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t;
    if (out == nullptr) smem[threadIdx.x] = 1;
}

// pure read in the one-shot geometry: every wave requests `np` 1-KiB pieces of its row up front
template <int D, bool NT_>
__global__ __launch_bounds__(1024) void read_kernel(const uint32_t* __restrict__ Qp, uint32_t* sink, int K, uint32_t geo) {
    extern __shared__ char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lkw = (geo >> 4) & 15, upw = (geo >> 8) & 31, pk = (geo >> 13) & 15;     // (four bits: 8 pieces per wave in the lean kernel's shape)
    const int ul = wave >> lkw, kpart = wave & ((1 << lkw) - 1);
    const int unit = blockIdx.x * upw + ul;
    const srd_t srd = make_srd(reinterpret_cast<const char*>(Qp) + (size_t)unit * K * 2, (uint32_t)K * 2u);
    ring16_t q[D];
#pragma unroll
    for (int i = 0; i < D; ++i) {
        const uint32_t vo = (uint32_t)lane * 16u + ((i < pk) ? (uint32_t)(kpart * pk + i) * 1024u : 0x80000000u);
        q[i] = NT_ ? buf_load16_nt(vo, srd, 0) : buf_load16(vo, srd, 0);
    }
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < D; ++i) {
        asm volatile("s_waitcnt vmcnt(%1)" : "+v"(q[i]) : "n"(D - 1 - i) : "memory");
        acc ^= q[i].x ^ q[i].y ^ q[i].z ^ q[i].w;
    }
    if (acc == 0x12345678u) { sink[0] = acc; smem[tid] = 1; }
}

struct Problem {
    int N, K, g, M;
    std::vector<uint32_t> Q;           // [P][K/2]
    std::vector<_Float16> S, X, table;
    std::vector<uint32_t> table2;
    std::vector<double> ref;           // [M][N]
};

static uint16_t h2u(_Float16 h) { uint16_t u; memcpy(&u, &h, 2); return u; }

static void make_problem(Problem& p, int N, int K, int g, int M) {
    p.N = N; p.K = K; p.g = g; p.M = M;
    const int J = 4, TP = 32, G = K / g;
    std::vector<uint8_t> W((size_t)K * N);
    for (auto& w : W) w = rnd() & 15;
    p.table.resize(16);
    for (int i = 0; i < 16; ++i) p.table[i] = (_Float16)nrand();
    p.table2.resize(256);
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) p.table2[i * 16 + j] = (uint32_t)h2u(p.table[i]) | ((uint32_t)h2u(p.table[j]) << 16);
    p.S.resize((size_t)N * G);
    for (auto& s : p.S) s = (_Float16)nrand();
    p.X.resize((size_t)M * K);
    for (auto& x : p.X) x = (_Float16)(nrand() / 100.0f);
    const int P = N / J;
    p.Q.assign((size_t)P * (K / 2), 0);
    for (int pr = 0; pr < P; ++pr) {
        const int nb = pr / TP, t = pr % TP;
        for (int kap = 0; kap < K / 2; ++kap) {
            uint32_t w = 0;
            for (int j = 0; j < J; ++j) {
                const int n = nb * J * TP + j * TP + t;
                const uint32_t f = ((uint32_t)W[(size_t)(2 * kap) * N + n] << 4) | W[(size_t)(2 * kap + 1) * N + n];
                w |= f << (8 * j);
            }
            p.Q[(size_t)pr * (K / 2) + kap] = w;
        }
    }
    p.ref.assign((size_t)M * N, 0.0);
    for (int m = 0; m < M; ++m)
        for (int k = 0; k < K; ++k) {
            const double x = (double)(float)p.X[(size_t)m * K + k];
            for (int n = 0; n < N; ++n) {
                const _Float16 w = (_Float16)((float)p.table[W[(size_t)k * N + n]] * (float)p.S[(size_t)n * G + k / g]);
                p.ref[(size_t)m * N + n] += x * (double)(float)w;
            }
        }
}

struct Dev {
    int ncopy;
    uint32_t* Q; _Float16* S; _Float16* X; uint32_t* T2; _Float16* D; uint32_t* sink;
    size_t qwords, swords;
};

typedef void (*OneKernel)(const uint32_t*, const void*, const void*, const uint32_t*, int, int, uint32_t, int, void*, float, uint64_t*);

struct Variant { const char* name; OneKernel fn; int W, kw, pk, depth; };

static double time_graph(hipStream_t st, int ncopy, int reps, const std::function<void(int)>& launch) {
    hipGraph_t graph; hipGraphExec_t exec;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int c = 0; c < ncopy; ++c) launch(c);
    CK(hipStreamEndCapture(st, &graph));
    CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(exec, st));
    CK(hipStreamSynchronize(st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    double best = 1e30;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(e0, st));
        CK(hipGraphLaunch(exec, st));
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, (double)ms * 1e3 / ncopy);
    }
    CK(hipGraphExecDestroy(exec)); CK(hipGraphDestroy(graph));
    return best;
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 4096, K = argc > 2 ? atoi(argv[2]) : 4096, g = argc > 3 ? atoi(argv[3]) : 64;
    const int M = 1;
    const char* tag = argc > 4 ? argv[4] : "";
    Problem p;
    make_problem(p, N, K, g, M);
    const int G = K / g, lg = (int)log2((double)g);
    Dev d;
    d.qwords = p.Q.size(); d.swords = p.S.size();
    const size_t copies_mb = getenv("LAB_COPIES_MB") ? (size_t)atoi(getenv("LAB_COPIES_MB")) : 300;      // rotating weight copies: total size
    d.ncopy = (int)((copies_mb << 20) / (d.qwords * 4)) + 1;
    CK(hipMalloc(&d.Q, d.qwords * 4 * d.ncopy)); CK(hipMalloc(&d.S, d.swords * 2 * d.ncopy));
    CK(hipMalloc(&d.X, p.X.size() * 2)); CK(hipMalloc(&d.T2, 1024)); CK(hipMalloc(&d.D, (size_t)M * N * 2 * d.ncopy)); CK(hipMalloc(&d.sink, 64));
    for (int c = 0; c < d.ncopy; ++c) {
        CK(hipMemcpy(d.Q + (size_t)c * d.qwords, p.Q.data(), d.qwords * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(d.S + (size_t)c * d.swords, p.S.data(), d.swords * 2, hipMemcpyHostToDevice));
    }
    CK(hipMemcpy(d.X, p.X.data(), p.X.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(d.T2, p.table2.data(), 1024, hipMemcpyHostToDevice));
    hipStream_t st; CK(hipStreamCreate(&st));
    const int units = N / 4, npieces = (K + 511) / 512;
    const double bytes = 2.0 * (N / 4) * K + 2.0 * N * G + 2.0 * M * K + 2.0 * M * N + 32 + 1024;

    const char* mode = argc > 5 ? argv[5] : "all";              // all | fast (round 5: the lean kernel next to the shipped one) | floors
    const bool run_old = strcmp(mode, "all") == 0, run_fast = strcmp(mode, "all") == 0 || strcmp(mode, "fast") == 0;
    // check one launch against the CPU restatement, (stamps build: one stamped HBM-cold launch,) then time the graph
    auto check_and_time = [&](const char* name, int W, int kw, int pk, int grid, size_t lds, const std::function<void(int, uint64_t*)>& launch_s) {
        auto launch = [&](int c) { launch_s(c, nullptr); };
        CK(hipMemsetAsync(d.D, 0xff, (size_t)M * N * 2, st));
        launch(0);
        CK(hipStreamSynchronize(st)); CK(hipGetLastError());
        std::vector<_Float16> out((size_t)M * N);
        CK(hipMemcpy(out.data(), d.D, out.size() * 2, hipMemcpyDeviceToHost));
        double num = 0, den = 0; int nbad = 0;
        for (size_t i = 0; i < out.size(); ++i) {
            const double e = (double)(float)out[i] - p.ref[i];
            num += e * e; den += p.ref[i] * p.ref[i];
            if (fabs(e) > 0.02 * sqrt(den / (i + 1)) + 1e-3) ++nbad;
        }
        const double rel = sqrt(num / den);
#ifdef FLUTE_STAMPS
        {   // one stamped launch on a copy the graph replays have pushed out of the caches
            const int nw = grid * W;
            uint64_t* dst; CK(hipMalloc(&dst, (size_t)nw * 128)); CK(hipMemset(dst, 0, (size_t)nw * 128));
            for (int c = 1; c < d.ncopy; ++c) launch(c);
            CK(hipStreamSynchronize(st));
            launch_s(0, dst);
            CK(hipStreamSynchronize(st));
            std::vector<uint64_t> hs((size_t)nw * 16);
            CK(hipMemcpy(hs.data(), dst, hs.size() * 8, hipMemcpyDeviceToHost)); CK(hipFree(dst));
            uint64_t t0 = ~0ull; for (int w = 0; w < nw; ++w) t0 = std::min(t0, hs[(size_t)w * 16]);
            auto qt = [&](std::vector<double> x, double f) { std::sort(x.begin(), x.end()); return x[(size_t)(f * (x.size() - 1))]; };
            std::vector<double> st0, en, ph[11];
            for (int w = 0; w < nw; ++w) {
                const uint64_t* h = &hs[(size_t)w * 16];
                st0.push_back((double)(h[0] - t0) / 100.0); en.push_back((double)(h[13] - t0) / 100.0);
                for (int i = 0; i < 11; ++i) if (h[2 + i]) ph[i].push_back((double)(h[2 + i] - h[1]));     // (helper waves leave at the barrier: no later stamps)
            }
            const char* names[11] = {"issued", "table_word", "table_written", "x_arrived", "x_written", "barrier", "scales_written", "pieces_done", "wave_reduced", "stored", "store_acked"};
            printf("{\"stamps\": \"%s\", \"start_us\": [%.2f, %.2f, %.2f], \"end_us\": [%.2f, %.2f, %.2f]", name, qt(st0, .5), qt(st0, .9), qt(st0, 1.), qt(en, .5), qt(en, .9), qt(en, 1.));
            for (int i = 0; i < 11; ++i) if (!ph[i].empty()) printf(", \"%s\": [%.0f, %.0f, %.0f]", names[i], qt(ph[i], .1), qt(ph[i], .5), qt(ph[i], .9));
            // the late half of the grid (workgroups that start after the median): their phases decide the end of the launch
            std::vector<double> late[11];
            const double med = qt(st0, .5);
            for (int w = 0; w < nw; ++w) if (st0[w] > med) for (int i = 0; i < 11; ++i) if (hs[(size_t)w * 16 + 2 + i]) late[i].push_back((double)(hs[(size_t)w * 16 + 2 + i] - hs[(size_t)w * 16 + 1]));
            if (!late[0].empty()) { printf(", \"late_half_p50\": ["); for (int i = 0; i < 11; ++i) printf("%s%.0f", i ? ", " : "", late[i].empty() ? -1.0 : qt(late[i], .5)); printf("]"); }
            printf("}\n");
        }
#endif
        const double us = time_graph(st, d.ncopy, 8, launch);
        printf("{\"variant\": \"%s\", \"tag\": \"%s\", \"N\": %d, \"K\": %d, \"g\": %d, \"W\": %d, \"kw\": %d, \"pk\": %d, \"grid\": %d, \"lds\": %zu, \"rel_err\": %.3e, \"nbad\": %d, \"us\": %.3f, \"GBps\": %.1f}\n",
               name, tag, N, K, g, W, kw, pk, grid, lds, rel, nbad, us, bytes / us / 1e3);
        fflush(stdout);
    };
#define V(name, D_, X_, OPT_, W_, kw_) Variant{name, (OneKernel)qgemv_oneshot_kernel<F16, 4, 32, 1, D_, X_, false, OPT_>, W_, kw_, 0, D_}
    std::vector<Variant> vs = {V("w4_kw1_d8_pipe_nt_il", 8, 2, 49, 4, 1), V("w4_kw1_d8_nolookup_il", 8, 2, 34, 4, 1)};
    if (run_old) {
        std::vector<Variant> more = {
            V("w4_kw1_d8_pipe_nt", 8, 2, 17, 4, 1), V("w4_kw1_d8_nolookup", 8, 2, 2, 4, 1),
            V("w8_kw2_d4_pipe_nt", 4, 1, 17, 8, 2), V("w8_kw2_d4_pipe_nt_il", 4, 1, 49, 8, 2),
            V("w4_kw2_d4_pipe_nt", 4, 2, 17, 4, 2), V("w4_kw2_d4_pipe_nt_il", 4, 2, 49, 4, 2),
            V("w8_kw1_d8_pipe_nt", 8, 1, 17, 8, 1), V("w8_kw1_d8_pipe_nt_il", 8, 1, 49, 8, 1),
            V("w8_kw4_d2_pipe_nt_il", 2, 1, 49, 8, 4), V("w16_kw4_d4_nt_il", 4, 1, 33, 16, 4),
        };
        vs.insert(vs.end(), more.begin(), more.end());
    }
    if (strcmp(mode, "floors") == 0) vs.clear();
    for (auto& v : vs) {
        const int pk = (npieces + v.kw - 1) / v.kw;
        if (pk > v.depth) { printf("{\"variant\": \"%s\", \"skip\": \"pk %d > depth\"}\n", v.name, pk); continue; }
        const int upw = v.W / v.kw, lkw = (int)log2((double)v.kw);
        const int ipw = (32 + v.W - 1) / v.W;
        const uint32_t geo = OneGeo::pack(lg, lkw, upw, pk, ipw, 0, oneshot_x_in_holes(4, 1, K) ? 1 : 0);
        const int grid = (units + upw - 1) / upw;
        const size_t lds = oneshot_lds_bytes(4, 1, v.depth, lg, K, v.W);
        CK(hipFuncSetAttribute((const void*)v.fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        check_and_time(v.name, v.W, v.kw, pk, grid, lds, [&](int c, uint64_t* stamps) {
            hipLaunchKernelGGL(v.fn, dim3(grid), dim3(v.W * 64), lds, st, d.Q + (size_t)c * d.qwords, (const void*)(d.S + (size_t)c * d.swords),
                               (const void*)d.X, d.T2, K, N, geo, M, (void*)(d.D + (size_t)c * M * N), 1.0f, stamps);
        });
    }
    // round 5: the lean kernel (qgemm_fast.h): K = 512 * D * KW is a compile-time constant
    if (run_fast) {
        typedef void (*FastKernel)(const uint32_t*, const void*, const void*, const uint32_t*, void*, int, int, int, uint64_t*);
        struct FV { const char* name; FastKernel fn; int W, kw, depth, H; bool xh; };
#define FV_(name, W_, KW_, D_, H_, OPT_) FV{name, (FastKernel)qgemv_fast_kernel<F16, 32, W_, KW_, D_, 1, OPT_>, W_, KW_, D_, 0, false}
        const FV fvs[] = {
            FV_("fast_w4_kw1_d8", 4, 1, 8, 0, 0), FV_("fast_w8_kw2_d4", 8, 2, 4, 0, 0), FV_("fast_w4_kw2_d4", 4, 2, 4, 0, 0),
            FV_("fast_w4_kw1_d8_hipcc_order", 4, 1, 8, 0, 32), FV_("fast_w4_kw1_d8_nolookup", 4, 1, 8, 0, 2),
            FV_("fast_w8_kw2_d8", 8, 2, 8, 0, 0), FV_("fast_w16_kw4_d4", 16, 4, 4, 0, 0),
            FV_("fast_w4_kw1_d4", 4, 1, 4, 0, 0), FV_("fast_w8_kw2_d2", 8, 2, 2, 0, 0),
        };
        for (const FV& v : fvs) {
            if (512 * v.depth * v.kw != K) continue;
            const int upw = v.W / v.kw;
            if (units % upw) continue;
            const int grid = units / upw;
            const size_t lds = fast_lds_bytes(v.W, v.kw, v.depth, lg, 1);
            CK(hipFuncSetAttribute((const void*)v.fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            check_and_time(v.name, v.W + v.H, v.kw, v.depth, grid, lds, [&](int c, uint64_t* stamps) {
                hipLaunchKernelGGL(v.fn, dim3(grid), dim3((v.W + v.H) * 64), lds, st, d.Q + (size_t)c * d.qwords, (const void*)(d.S + (size_t)c * d.swords),
                                   (const void*)d.X, d.T2, (void*)(d.D + (size_t)c * M * N), N, lg, M, stamps);
            });
        }
    }
    // round-2 one-shot kernel (qgemm_stream.h), the plan the round-2 tuner took for the headline: 8 waves, kw 2
    if (run_old && units % 4 == 0 && units / 4 <= 512 && npieces <= 8) {
        StreamArgs sa;
        memset(&sa, 0, sizeof(sa));
        sa.M = M; sa.N = N; sa.K = K; sa.G = G; sa.lg = lg; sa.units = units; sa.upw = 4; sa.kw = 2; sa.lkw = 1;
        sa.ngroups = units / 4; sa.nwg = sa.ngroups; sa.vis_q = 1; sa.vis_r = 0; sa.splitk = 1; sa.k_per_split = npieces * 512;
        sa.kc = npieces * 512; sa.nchunks = 1; sa.kx = npieces * 512; sa.x_off = 65536; sa.s_off = sa.x_off + sa.kx * 2;
        const int pk = (npieces + 1) / 2, ngran = (pk * (512 >> lg) + 7) / 8;
        sa.s_wave_bytes = ngran * 4 * 16; sa.red_off = sa.s_off + 8 * sa.s_wave_bytes; sa.s_fast = (G % 8 == 0) ? 1 : 0;
        const size_t lds = sa.red_off + 64 + 2 * 8 * 4 * 4;
        auto fn = qgemv_stream_kernel<F16, 4, 32, 1, 4, true>;
        CK(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        auto launch = [&](int c) {
            StreamArgs a = sa;
            a.A = d.X; a.Q = d.Q + (size_t)c * d.qwords; a.D = d.D + (size_t)c * M * N; a.S = d.S + (size_t)c * d.swords; a.QM2 = d.T2;
            hipLaunchKernelGGL(fn, dim3(sa.nwg), dim3(512), lds, st, a);
        };
        CK(hipMemsetAsync(d.D, 0xff, (size_t)M * N * 2, st));
        launch(0);
        CK(hipStreamSynchronize(st)); CK(hipGetLastError());
        std::vector<_Float16> out((size_t)M * N);
        CK(hipMemcpy(out.data(), d.D, out.size() * 2, hipMemcpyDeviceToHost));
        double num = 0, den = 0;
        for (size_t i = 0; i < out.size(); ++i) { const double e = (double)(float)out[i] - p.ref[i]; num += e * e; den += p.ref[i] * p.ref[i]; }
        const double us = time_graph(st, d.ncopy, 8, launch);
        printf("{\"variant\": \"r02_stream_oneshot_w8_kw2\", \"tag\": \"%s\", \"N\": %d, \"K\": %d, \"grid\": %d, \"lds\": %zu, \"rel_err\": %.3e, \"us\": %.3f, \"GBps\": %.1f}\n",
               tag, N, K, sa.nwg, lds, sqrt(num / den), us, bytes / us / 1e3);
        fflush(stdout);
    }
    if (run_old) {   // dispatch ramp: when does each wave of an (otherwise empty) grid start?
        const int cfg[][3] = {{256, 256, 0}, {256, 512, 0}, {256, 512, 76 * 1024}, {512, 256, 0}, {1024, 256, 0}, {256, 1024, 0}, {512, 512, 0}, {128, 512, 0}, {2048, 64, 0}, {1024, 128, 0}};
        CK(hipFuncSetAttribute((const void*)ramp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        for (auto& c : cfg) {
            const int nw = c[0] * c[1] / 64;
            uint64_t* dst; CK(hipMalloc(&dst, (size_t)nw * 8));
            std::vector<double> best;
            for (int rep = 0; rep < 5; ++rep) {
                hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(512), 0, st, d.Q, 1);
                hipLaunchKernelGGL(ramp_kernel, dim3(c[0]), dim3(c[1]), (size_t)c[2], st, dst);
                CK(hipStreamSynchronize(st));
                std::vector<uint64_t> h(nw);
                CK(hipMemcpy(h.data(), dst, (size_t)nw * 8, hipMemcpyDeviceToHost));
                std::sort(h.begin(), h.end());
                if (rep == 4) printf("{\"ramp\": [%d, %d, %d], \"waves\": %d, \"start_us_p50_p90_max\": [%.2f, %.2f, %.2f]}\n", c[0], c[1], c[2], nw,
                                     (h[nw / 2] - h[0]) / 100.0, (h[nw * 9 / 10] - h[0]) / 100.0, (h[nw - 1] - h[0]) / 100.0);
            }
            CK(hipFree(dst));
        }
    }
    // floors: empty kernel, pure read in the same geometries
    {
        auto launch = [&](int c) { hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(512), 0, st, d.Q, 1); (void)c; };
        printf("{\"variant\": \"empty_256x512\", \"tag\": \"%s\", \"us\": %.3f}\n", tag, time_graph(st, d.ncopy, 8, launch));
        auto launch2 = [&](int c) { hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(512), 76 * 1024, st, d.Q, 1); (void)c; };
        CK(hipFuncSetAttribute((const void*)empty_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        printf("{\"variant\": \"empty_256x512_lds76k\", \"tag\": \"%s\", \"us\": %.3f}\n", tag, time_graph(st, d.ncopy, 8, launch2));
    }
    struct RV { const char* name; int W, kw, depth; bool nt; size_t lds; };
    const RV rvs[] = {{"read_w8_kw2_d4", 8, 2, 4, false, 0}, {"read_w8_kw2_d4_nt", 8, 2, 4, true, 0}, {"read_w8_kw2_d4_lds76k", 8, 2, 4, false, 76 * 1024},
                      {"read_w16_kw4_d2", 16, 4, 2, false, 0}, {"read_w8_kw4_d2", 8, 4, 2, false, 0}, {"read_w8_kw4_d2_lds76k", 8, 4, 2, false, 76 * 1024},
                      {"read_w4_kw2_d4", 4, 2, 4, false, 0}, {"read_w4_kw4_d2", 4, 4, 2, false, 0}, {"read_w4_kw4_d2_nt", 4, 4, 2, true, 0},
                      // round 6: the lean decode kernel's own launch shape (qgemm_fast.h: 256 workgroups x 4 waves x 8 pieces, nt loads, 72.6 KB of LDS)
                      {"read_w4_kw1_d8", 4, 1, 8, false, 0}, {"read_w4_kw1_d8_nt", 4, 1, 8, true, 0}, {"read_w4_kw1_d8_nt_lds73k", 4, 1, 8, true, 73 * 1024},
                      {"read_w8_kw2_d4_nt_lds73k", 8, 2, 4, true, 73 * 1024}};
    for (const RV& r : rvs) {
        const int pk = (npieces + r.kw - 1) / r.kw;
        if (pk > r.depth) continue;
        const int upw = r.W / r.kw, lkw = (int)log2((double)r.kw);
        const uint32_t geo = OneGeo::pack(lg, lkw, upw, pk, 0, 0, 0);
        const int grid = units / upw;
        void (*fn)(const uint32_t*, uint32_t*, int, uint32_t) =
            r.depth == 8 ? (r.nt ? read_kernel<8, true> : read_kernel<8, false>)
                         : (r.depth == 4 ? (r.nt ? read_kernel<4, true> : read_kernel<4, false>) : (r.nt ? read_kernel<2, true> : read_kernel<2, false>));
        CK(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        auto launch = [&](int c) { hipLaunchKernelGGL(fn, dim3(grid), dim3(r.W * 64), r.lds, st, d.Q + (size_t)c * d.qwords, d.sink, K, geo); };
        printf("{\"variant\": \"%s\", \"tag\": \"%s\", \"N\": %d, \"K\": %d, \"grid\": %d, \"us\": %.3f}\n", r.name, tag, N, K, grid, time_graph(st, d.ncopy, 8, launch));
        fflush(stdout);
    }
    return 0;
}
