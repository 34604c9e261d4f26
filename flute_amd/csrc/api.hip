// Host side of the C ABI (include/flute_amd.h): template table, launch planning
// and dispatch.  Mirrors the role of flute/csrc/qgemm.cpp:39-83 (qgemm_raw) +
// qgemm_kernel_raw_generated.cu:15-768 (_qgemm_raw's template switch) +
// qgemm_kernel.hpp:824-939 (qgemm_host), re-thought for gfx950: instead of one
// Stream-K kernel with 36/144 tile variants there are two kernel families
// (streaming decode for M <= 4 - 3-bit: M <= 2 -, MFMA above) whose launch geometry is derived
// from the template's knobs and the problem shape.
#include <hip/hip_runtime.h>
#include <math.h>
#include <mutex>
#include <stdint.h>
#include <string.h>

#include "../../include/flute_amd.h"
#include "kernels.h"
#include "qgemm_decode.h"
#include "mfma.h"
#include "qgemm_tile.h"

using namespace flute_amd;

namespace {

struct Overrides { int family, m_block, waves, kw, splitk, lut_copies, prescale; };
Overrides g_ovr = {-1, -1, -1, -1, -1, -1, -1};

constexpr int kMaxLds = 160 * 1024;

int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }
int floor_pow2(int v) { int p = 1; while (p * 2 <= v) p *= 2; return p; }
int ceil_div(int a, int b) { return (a + b - 1) / b; }
int round_up(int a, int b) { return ceil_div(a, b) * b; }

// Template id -> knobs.  Enumeration order is the reference's
// (flute/codegen_utils.py:110-152): SMs_Multiple {1,2,4} x tile {0,1,2} x
// stages {2,3,4,5} x (b=4 only) QuantMapMode {Vectorized,_32,_16,_8}; tile 0
// has TileP 64, tiles 1,2 TileP 32, so id -> TileP is identical to the
// reference table and reference-packed weights keep their id.
bool decode_template(int bits, int id, flute_template_info* t) {
    if (bits != 2 && bits != 3 && bits != 4) return false;
    const int nq = (bits == 4) ? 4 : 1;
    const int total = 3 * 3 * 4 * nq;
    if (id < 0 || id >= total) return false;
    const int q = id % nq;
    const int st = (id / nq) % 4;
    const int tile = (id / (nq * 4)) % 3;
    const int mult = id / (nq * 12);
    static const int kMult[3] = {1, 2, 4};
    static const int kThreads[3] = {1024, 1024, 512};
    static const int kTileM[3] = {64, 64, 16};
    static const int kTileP[3] = {64, 32, 32};
    static const int kCopies[4] = {1, 32, 16, 8};
    t->num_bits = bits;
    t->template_id = id;
    t->sms_multiple = kMult[mult];
    t->threads = kThreads[tile];
    t->tile_m = kTileM[tile];
    t->tile_k = 64;
    t->tile_p = kTileP[tile];
    t->stages = 2 + st;
    // b=2/3 pair tables are 64 B / 256 B: always replicate them 32x
    t->lut_copies = (bits == 4) ? kCopies[q] : 32;
    return true;
}

int make_plan(int dtype, int bits, int group, int M, int N, int K, int template_id, int num_sms,
              size_t workspace_bytes, flute_plan* p, flute_template_info* tinfo) {
    if (dtype != 0 && dtype != 1) return FLUTE_ERR_DTYPE;
    if (bits != 2 && bits != 3 && bits != 4) return FLUTE_ERR_NUM_BITS;
    if (group != 32 && group != 64 && group != 128 && group != 256) return FLUTE_ERR_GROUP_SIZE;
    flute_template_info t;
    if (!decode_template(bits, template_id, &t)) return FLUTE_ERR_TEMPLATE_ID;
    if (bits == 3 && t.tile_p != 32) return FLUTE_ERR_TEMPLATE_ID;   // utils.py:137-139
    if (tinfo) *tinfo = t;
    const int J = (bits == 3) ? 16 : 16 / bits;
    if (M < 1 || N < 1 || K < 1) return FLUTE_ERR_SHAPE;
    if (N % (J * t.tile_p) || K % 64 || K % group) return FLUTE_ERR_SHAPE;
    if (num_sms < 1) num_sms = 256;
    const int lg = ilog2(group);
    const int units = N / J;
    const int lines = K / 64;

    memset(p, 0, sizeof(*p));

    const int dec_max = (bits == 3) ? 2 : 4;
    int family = (M <= dec_max) ? 0 : 2;
    if (g_ovr.family == 0 && M <= dec_max) family = 0;
    if (g_ovr.family >= 1) family = 2;            // any M may be forced through the MFMA kernel
    p->family = family;

    if (family == 0) {
        int mb = 1; while (mb < M) mb <<= 1;
        if (g_ovr.m_block > 0 && g_ovr.m_block >= M && g_ovr.m_block <= dec_max) mb = g_ovr.m_block;
        int waves = t.threads / 64;
        if (g_ovr.waves > 0) waves = floor_pow2(g_ovr.waves);      // the kernels shift by log2(waves), log2(kw)
        if (waves > dec_max_threads(bits, mb) / 64) waves = dec_max_threads(bits, mb) / 64;
        if (waves < 1) waves = 1;
        // K split so that the chip holds >= num_sms * mult workgroups' worth of waves; every
        // octet of a wave keeps at least one 64-k line
        const long target_waves = (long)num_sms * t.sms_multiple * waves;
        int f = 1;
        while ((long)units * f < target_waves && lines / (f * 2) >= 8) f *= 2;
        int kw = f < waves ? f : waves;
        int splitk = f / kw;
        // Stages (2..5) has no pipeline to size here: the tuner uses it to try the neighbouring
        // K splits (same, half, double, quadruple)
        if (t.stages == 3 && kw > 1) kw >>= 1;
        if (t.stages == 4) kw <<= 1;
        if (t.stages == 5) kw <<= 2;
        if (g_ovr.kw > 0) kw = floor_pow2(g_ovr.kw);
        if (g_ovr.splitk > 0) splitk = g_ovr.splitk;
        if (kw > waves) kw = waves;
        while (waves % kw) kw >>= 1;
        while ((units % (waves / kw)) && kw < waves) kw <<= 1;
        int kps = round_up(ceil_div(K, splitk), 512);
        splitk = ceil_div(K, kps);
        while (splitk > 1 && (size_t)splitk * M * N * 4 > workspace_bytes) {
            splitk >>= 1;
            kps = round_up(ceil_div(K, splitk), 512);
            splitk = ceil_div(K, kps);
        }
        if (splitk == 1) kps = K;
        const DecodeGeom geo = decode_geom(bits, mb, lg, waves, kw, kps, kMaxLds);
        const int ngroups = units / (waves / kw);
        int occ = (int)(kMaxLds / geo.total);
        if (occ > 2048 / (waves * 64)) occ = 2048 / (waves * 64);
        if (occ < 1) occ = 1;
        long nwg = (long)num_sms * occ;
        if (nwg > ngroups) nwg = ngroups;
        p->m_block = mb; p->waves = waves; p->kw = kw; p->splitk = splitk; p->k_per_split = kps;
        p->grid = (unsigned)(nwg * splitk);
        p->block = (unsigned)(waves * 64);
        p->lds_bytes = geo.total;
        p->lut_copies = (bits == 4) ? 64 : 32;
    } else {
        // M > decode range: MFMA kernel (qgemm_tile.h).  MT 16-row tiles per wave (1 for M <= 16),
        // R lanes share a unit: pick the smallest R whose slab x row-tile count fills the chip; the
        // rest of the parallelism is the in-workgroup K split, a grid-level split only for very
        // narrow layers.
        int mt = (M <= 16) ? 1 : (M <= 32 ? 2 : 4);
        const int mt_cap = (bits == 3) ? 2 : 4;
        if (mt > mt_cap) mt = mt_cap;
        if (t.tile_m / 16 < mt && M > 16) mt = t.tile_m / 16 >= 2 ? t.tile_m / 16 : mt;
        // SMs_Multiple = "more, smaller workgroups": halves / quarters the row tiles per wave (and,
        // below, raises the slab count the choice of R aims for)
        for (int m2 = t.sms_multiple; m2 > 1 && mt > 1; m2 >>= 1) mt >>= 1;
        if (g_ovr.lut_copies == 1 || g_ovr.lut_copies == 2 || g_ovr.lut_copies == 4) mt = g_ovr.lut_copies;
        if (mt > mt_cap) mt = mt_cap;
        // instantiated (R, MT): (J/R)*MT <= 16 accumulator tiles, R in {1,2,4}, MT > 1 needs R <= 2
        auto combo_ok = [&](int r, int m) {
            if (bits == 3) return r == 1 && m == 1;
            return (J / r) * m <= 16 && r <= 4 && (m == 1 || r <= 2);
        };
        while (mt > 1 && !combo_ok(1, mt) && !combo_ok(2, mt)) mt >>= 1;
        const int mtiles = ceil_div(M, mt * 16);
        int R = 1;
        while (!combo_ok(R, mt)) R *= 2;
        while (combo_ok(R * 2, mt) && (long)units * R / 16 * mtiles < (long)num_sms * t.sms_multiple) R *= 2;
        if (g_ovr.m_block > 0 && combo_ok(g_ovr.m_block, mt)) R = g_ovr.m_block;
        // SW = 2 slabs per wave (4-bit, no lane sharing, fp16 up to MT = 4 / bf16 up to MT = 2: the bf16 path
        // keeps a second accumulator set): every activation fragment then serves 8 column tiles and the
        // texture-path traffic per MFMA drops by 40 %.  Worth it once halving the slab count still leaves a
        // workgroup for every CU; QuantMapMode (the last template digit) lets the tuner force either.
        const bool sw_ok = bits == 4 && R == 1 && mt >= 2 && (dtype == 0 || mt == 2) && (units / 16) % 2 == 0;
        int sw = 1;
        if (sw_ok && (long)(units / 32) * mtiles >= (long)num_sms) sw = 2;
        if (sw_ok && bits == 4 && (template_id % 4) == 3) sw = 2;
        if (bits == 4 && (template_id % 4) == 2) sw = 1;
        if (sw_ok && g_ovr.prescale == 2) sw = 2;
        if (g_ovr.prescale == 1) sw = 1;
        const int slabs = units * R / 16 / sw;                    // wave-sized column groups
        int nw = (t.threads >= 1024) ? 8 : 4;                     // Threads 1024 / 512 templates
        if (g_ovr.waves > 0 && g_ovr.waves <= 8) nw = floor_pow2(g_ovr.waves);
        while (nw > 1 && tile_geom(bits, R, mt, sw, nw, kMaxLds).depth < 2) nw >>= 1;   // ring of >= 2 slots per wave
        int kw = nw;
        while (kw > 1 && K / kw < 256) kw >>= 1;
        // enough workgroups already: keep more of K per wave (fewer partial tiles to reduce)
        while (kw > 1 && (long)slabs * mtiles / (nw / kw) >= 2L * num_sms * t.sms_multiple && K / kw < 1024) kw >>= 1;
        if (t.stages == 3 && kw > 1) kw >>= 1;                    // the tuner's handle on the K split (see decode)
        if (t.stages == 4 && kw < nw) kw <<= 1;
        if (t.stages == 5 && kw > 2) kw >>= 2;
        if (g_ovr.kw > 0 && g_ovr.kw <= nw) kw = floor_pow2(g_ovr.kw);
        while (nw % kw) kw >>= 1;
        while (slabs % (nw / kw)) kw <<= 1;
        const long wgs = (long)slabs / (nw / kw) * mtiles;
        int splitk = 1;
        while (wgs * splitk * 2 <= (long)num_sms && K / (splitk * 2 * kw) >= 256) splitk *= 2;
        if (g_ovr.splitk > 0) splitk = g_ovr.splitk;
        int kps = round_up(ceil_div(K, splitk), 32 * kw);
        splitk = ceil_div(K, kps);
        while (splitk > 1 && (size_t)splitk * M * N * 4 > workspace_bytes) {
            splitk >>= 1;
            kps = round_up(ceil_div(K, splitk), 32 * kw);
            splitk = ceil_div(K, kps);
        }
        if (splitk == 1) kps = K;
        p->m_block = R; p->m_tiles = mt; p->slabs_per_wave = sw; p->waves = nw; p->kw = kw; p->splitk = splitk;
        p->k_per_split = kps;
        p->grid = (unsigned)(wgs * splitk);
        p->block = (unsigned)(nw * 64);
        p->lds_bytes = (size_t)tile_geom(bits, R, mt, sw, nw, kMaxLds).total;
        p->lut_copies = 32;
    }
    p->workspace_needed = p->splitk > 1 ? (size_t)p->splitk * M * N * 4 : 0;
    if (p->lds_bytes > (size_t)kMaxLds) return FLUTE_ERR_SHAPE;
    return FLUTE_OK;
}

QGemmKernel pick_kernel(int family, int bits, int dtype, int tile_p, int mblk, int mtiles, int sw) {
    if (family == 0) {
        const int pre = (g_ovr.prescale > 0) ? g_ovr.prescale : 0;
        if (bits == 4) return decode_kernel_b4(dtype, tile_p, mblk, pre);
        if (bits == 3) return decode_kernel_b3(dtype, tile_p, mblk, pre);
        return decode_kernel_b2(dtype, tile_p, mblk, pre);
    }
    if (bits == 4) return tile_kernel_b4(dtype, tile_p, mblk, mtiles, sw);
    if (bits == 3) return tile_kernel_b3(dtype, tile_p, mblk, mtiles);
    return tile_kernel_b2(dtype, tile_p, mblk, mtiles);
}

// (device, kernel) pairs already granted > 64 KB of dynamic LDS: the attribute is per device, and one
// process may drive several GPUs (accelerate-style sharded inference)
struct BigLds { int dev; const void* fn; };
BigLds g_big_lds[256];
int g_big_lds_n = 0;
std::mutex g_big_lds_mu;                       // flute_qgemm may be called from several host threads

int ensure_lds(const void* fn, size_t bytes) {
    if (bytes <= 65536) return 0;
    std::lock_guard<std::mutex> lock(g_big_lds_mu);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return FLUTE_ERR_LAUNCH; }
    for (int i = 0; i < g_big_lds_n; ++i)
        if (g_big_lds[i].fn == fn && g_big_lds[i].dev == dev) return 0;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds) != hipSuccess) {
        (void)hipGetLastError();
        return FLUTE_ERR_LAUNCH;
    }
    if (g_big_lds_n < 256) g_big_lds[g_big_lds_n++] = BigLds{dev, fn};
    return 0;
}

}  // namespace

extern "C" {

int flute_abi_version(void) { return FLUTE_AMD_ABI_VERSION; }

const char* flute_strerror(int status) {
    switch (status) {
        case FLUTE_OK: return "ok";
        case FLUTE_ERR_NUM_BITS: return "Unsupported num_bits value";
        case FLUTE_ERR_GROUP_SIZE: return "Unsupported group_size value";
        case FLUTE_ERR_TEMPLATE_ID: return "Unsupported template_id value";
        case FLUTE_ERR_SHAPE: return "Unsupported shape: need N % (16/num_bits*TileP) == 0 (N % 512 for 3 bits), K % 64 == 0, K % group_size == 0";
        case FLUTE_ERR_WORKSPACE: return "workspace too small";
        case FLUTE_ERR_LAUNCH: return "HIP error: kernel launch failed (invalid argument)";
        case FLUTE_ERR_DTYPE: return "Unsupported dtype (fp16 / bf16 only)";
        case FLUTE_ERR_HADAMARD_SIZE: return "Only power of two Hadamard sizes up to 2^15 are supported";
        case FLUTE_ERR_NULL: return "null pointer argument";
        default: return "unknown flute_amd status";
    }
}

void flute_set_overrides(int family, int m_block, int waves, int kw, int splitk, int lut_copies,
                         int prescale) {
    g_ovr = Overrides{family, m_block, waves, kw, splitk, lut_copies, prescale};
}

int flute_num_templates(int num_bits) {
    if (num_bits == 4) return 144;
    if (num_bits == 2 || num_bits == 3) return 36;
    return 0;
}

int flute_get_template_info(int num_bits, int template_id, flute_template_info* out) {
    if (!out) return FLUTE_ERR_NULL;
    if (num_bits != 2 && num_bits != 3 && num_bits != 4) return FLUTE_ERR_NUM_BITS;
    return decode_template(num_bits, template_id, out) ? FLUTE_OK : FLUTE_ERR_TEMPLATE_ID;
}

int flute_qgemm_plan(int dtype, int num_bits, int group_size, int M, int N, int K,
                     int template_id, int num_sms, size_t workspace_bytes, flute_plan* out) {
    if (!out) return FLUTE_ERR_NULL;
    return make_plan(dtype, num_bits, group_size, M, N, K, template_id, num_sms, workspace_bytes,
                     out, nullptr);
}

int flute_qgemm(int dtype, int num_bits, int group_size, int M, int N, int K, int P,
                const void* A, const void* Q, void* D, const void* S, const void* QM,
                const void* QM2, void* workspace, size_t workspace_bytes, int template_id,
                int num_sms, void* stream) {
    return flute_qgemm_hadamard(dtype, num_bits, group_size, 0, M, N, K, P, A, Q, D, S, QM, QM2, nullptr,
                                workspace, workspace_bytes, template_id, num_sms, stream);
}

int flute_qgemm_hadamard_fused(int dtype, int num_bits, int group_size, int hadamard_size, int M,
                               int N, int K, int template_id, int num_sms, size_t workspace_bytes) {
    flute_plan p;
    if (make_plan(dtype, num_bits, group_size, M, N, K, template_id, num_sms, workspace_bytes, &p,
                  nullptr))
        return 0;
    return (p.family == 0 && hadamard_size >= 2 && hadamard_size <= 512 &&
            (hadamard_size & (hadamard_size - 1)) == 0 && K % hadamard_size == 0) ? 1 : 0;
}

int flute_qgemm_hadamard(int dtype, int num_bits, int group_size, int hadamard_size, int M, int N,
                         int K, int P, const void* A, const void* Q, void* D, const void* S,
                         const void* QM, const void* QM2, void* x_scratch, void* workspace,
                         size_t workspace_bytes, int template_id, int num_sms, void* stream) {
    (void)QM;   // single-code table: unused, the kernel reads only the pair table (as the reference)
    if (M == 0) return FLUTE_OK;
    int had_log = 0;
    if (hadamard_size > 1) {
        if (hadamard_size & (hadamard_size - 1)) return FLUTE_ERR_HADAMARD_SIZE;
        if (flute_qgemm_hadamard_fused(dtype, num_bits, group_size, hadamard_size, M, N, K, template_id,
                                       num_sms, workspace ? workspace_bytes : 0)) {
            had_log = ilog2(hadamard_size);              // rotated inside the decode kernel's staging
        } else {
            // two launches (qgemm.cpp:201-244): rotate into the caller's scratch, then the plain product
            if (!A || !x_scratch) return FLUTE_ERR_NULL;
            const int rc = hadamard_dispatch(dtype, A, x_scratch, (size_t)M * K, (uint32_t)hadamard_size,
                                             reinterpret_cast<hipStream_t>(stream));
            if (rc) return rc;
            A = x_scratch;
        }
    }
    flute_plan p;
    flute_template_info t;
    if (!workspace) workspace_bytes = 0;
    const int rc = make_plan(dtype, num_bits, group_size, M, N, K, template_id, num_sms,
                             workspace_bytes, &p, &t);
    if (rc) return rc;
    if (P != num_bits * (N / 16)) return FLUTE_ERR_SHAPE;
    if (!A || !Q || !D || !S || !QM2) return FLUTE_ERR_NULL;

    QGemmArgs a;
    a.A = A; a.Q = reinterpret_cast<const uint32_t*>(Q); a.D = D; a.S = S;
    a.QM2 = reinterpret_cast<const uint32_t*>(QM2);
    a.partial = reinterpret_cast<float*>(workspace);
    a.M = M; a.N = N; a.K = K; a.G = K / group_size;
    a.lg = ilog2(group_size);
    a.units = N / ((num_bits == 3) ? 16 : 16 / num_bits);
    a.splitk = p.splitk; a.k_per_split = p.k_per_split; a.kw = p.kw; a.m0 = 0;
    a.lut_shift = 0;
    a.lds_budget = kMaxLds;
    a.lkw = ilog2(p.kw);
    a.had_log = had_log;
    a.had_scale = 1.0f / sqrtf((float)(1 << had_log));       // as flute_hadamard: bit-identical results
    for (int i = 0; i < 10; ++i) a.geo[i] = 0;
    if (p.family == 0) {
        const DecodeGeom g = decode_geom(num_bits, p.m_block, a.lg, p.waves, p.kw, p.k_per_split, kMaxLds);
        a.geo[0] = g.kc; a.geo[1] = g.nbuf; a.geo[2] = g.gcap; a.geo[3] = ilog2(g.upw);
        a.geo[4] = (int)g.x_off; a.geo[5] = (int)g.s_off; a.geo[6] = (int)g.red_off; a.geo[7] = ilog2(g.kc);
        const int ngroups = a.units / g.upw, nwg = (int)p.grid / p.splitk;
        a.geo[8] = ngroups / nwg; a.geo[9] = ngroups % nwg;
    } else {
        const TileGeom g = tile_geom(num_bits, p.m_block, p.m_tiles, p.slabs_per_wave, p.waves, kMaxLds);
        a.geo[0] = g.depth; a.geo[1] = g.scale_bytes; a.geo[2] = g.slot_bytes; a.geo[3] = g.wave_bytes;
        a.geo[4] = ceil_div(M, p.m_tiles * 16);
        // slab groups a multiple of the 8 XCDs: row tiles of one slab share an XCD (qgemm_tile.h) - as long
        // as the activations (which every XCD then reads in full) are the smaller operand
        a.geo[5] = (((int)p.grid / p.splitk / a.geo[4]) % 8 == 0 && (long)M * 32 <= (long)num_bits * N) ? 1 : 0;
    }

    QGemmKernel fn = pick_kernel(p.family, num_bits, dtype, t.tile_p, p.m_block, p.m_tiles, p.slabs_per_wave);
    if (!fn) return FLUTE_ERR_TEMPLATE_ID;
    if (ensure_lds(reinterpret_cast<const void*>(fn), p.lds_bytes)) return FLUTE_ERR_LAUNCH;

    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    void* kargs[] = {&a};
    if (hipLaunchKernel(reinterpret_cast<const void*>(fn), dim3(p.grid), dim3(p.block), kargs,
                        p.lds_bytes, st) != hipSuccess) {
        (void)hipGetLastError();
        return FLUTE_ERR_LAUNCH;
    }
    if (p.splitk > 1)
        return splitk_reduce_dispatch(dtype, a.partial, D, (size_t)M * N, p.splitk, st);
    return FLUTE_OK;
}

int flute_hadamard(int dtype, const void* in, void* out, uint32_t numel, uint32_t had_size,
                   void* stream) {
    if (!in || !out) return numel == 0 ? FLUTE_OK : FLUTE_ERR_NULL;
    return hadamard_dispatch(dtype, in, out, numel, had_size, reinterpret_cast<hipStream_t>(stream));
}

int flute_unpack(int num_bits, int template_id, int N, int K, const void* Q, void* W,
                 void* stream) {
    flute_template_info t;
    if (num_bits != 2 && num_bits != 3 && num_bits != 4) return FLUTE_ERR_NUM_BITS;
    if (!decode_template(num_bits, template_id, &t)) return FLUTE_ERR_TEMPLATE_ID;
    if (num_bits == 3 && t.tile_p != 32) return FLUTE_ERR_TEMPLATE_ID;
    const int J = (num_bits == 3) ? 16 : 16 / num_bits;
    if (N < 1 || K < 2 || N % (J * t.tile_p) || K % 2) return FLUTE_ERR_SHAPE;
    if (!Q || !W) return FLUTE_ERR_NULL;
    return unpack_dispatch(num_bits, t.tile_p, N, K, Q, W, reinterpret_cast<hipStream_t>(stream));
}

int flute_debug_stream_read(const void* src, void* sink, size_t bytes, int bytes_per_wave,
                            int grid, int block, void* stream) {
    if (!src || !sink || bytes_per_wave < 8192 || bytes_per_wave % 8192) return FLUTE_ERR_SHAPE;
    return stream_read_dispatch(src, sink, bytes, bytes_per_wave, grid, block,
                                reinterpret_cast<hipStream_t>(stream));
}

}  // extern "C"
