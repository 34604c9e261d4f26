/* flute_amd C ABI - the drop-in boundary of the MI355X (gfx950) qgemm path.
 *
 * Plain pointers and sizes only (no torch types).  Every entry point is what a
 * binding for the reference's hot path would call; the reference interface it
 * replaces is cited on each declaration (paths relative to HanGuo97/flute
 * v0.4.2).  All functions return 0 on success or a negative flute_status;
 * flute_strerror() maps a status to the message prefix the reference raises
 * (the reference's tuner string-matches those, flute/tune.py:160-167).
 *
 * Device pointers must be valid on the current HIP device, contiguous and 16-B
 * aligned (torch allocations are).  Nothing here allocates, synchronises or
 * touches the host after enqueue: every call is stream-ordered on `stream`
 * (a hipStream_t) and hipGraph-capturable, like the reference's launch on
 * at::cuda::getCurrentCUDAStream() (flute/csrc/qgemm.cpp:101-105).
 */
#ifndef FLUTE_AMD_H
#define FLUTE_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 8 (round 6, late): same structs; family 8 = persistent MFMA decode kernel (qgemm_persistm.h) in flute_plan.family / flute_overrides.family -
 *    slabs_per_wave = column groups per set (1 .. 3), visits = sets per workgroup (override: m_tiles), k_chunks = activation requests per macro-step
 * 7 (round 6): same structs; flute_plan.kw / m_block of family 6 = K parts per workgroup (2 / 4) / row tiles per XCD group, flute_plan.slabs_per_wave
 *    of family 7 = column groups per workgroup (1 .. 3), and the overrides of the same names select them; family 6 refuses waves = 8
 * 6 (round 5): flute_plan.one_shot / flute_overrides.one_shot value 4 (lean decode kernel, qgemm_fast.h), flute_debug_timestamp */
#define FLUTE_AMD_ABI_VERSION 8

enum flute_dtype { FLUTE_F16 = 0, FLUTE_BF16 = 1 };

enum flute_status {
    FLUTE_OK = 0,
    FLUTE_ERR_NUM_BITS = -1,      /* "Unsupported num_bits value"    qgemm.cpp:171 */
    FLUTE_ERR_GROUP_SIZE = -2,    /* "Unsupported group_size value"  qgemm.cpp:153 */
    FLUTE_ERR_TEMPLATE_ID = -3,   /* "Unsupported template_id value" qgemm_kernel_raw_generated.cu:205 */
    FLUTE_ERR_SHAPE = -4,         /* shape / divisibility precondition (ops.py:40-49) */
    FLUTE_ERR_WORKSPACE = -5,     /* split-K needs more workspace than given */
    FLUTE_ERR_LAUNCH = -6,        /* HIP launch failed: message starts "CUDA error: invalid argument" (tune.py:160) */
    FLUTE_ERR_DTYPE = -7,
    FLUTE_ERR_HADAMARD_SIZE = -8, /* hadamard_transform.cpp:23-25 */
    FLUTE_ERR_NULL = -9
};

/* One row of the gfx950 template table.  Same fields as the reference's
 * TEMPLATE_CONFIGS entries (flute/codegen_utils.py:110-152,
 * data/qgemm_kernel_raw_generated_configs.pth) and the same id -> TileP map, so
 * weights packed by the reference for template id X decode correctly here under
 * the same id.  Meaning of the knobs on gfx950:
 *   sms_multiple  decode: weight-ring depth (1: 4 pieces, 2/4: 2 pieces in flight per wave); MFMA kernel:
 *                 more, smaller workgroups
 *   threads       upper bound of the decode kernel's workgroup size (1024 / 512); MFMA kernel: 8 / 4 waves
 *   tile_m        rows per wave of the MFMA kernel (16/32/64)
 *   tile_k        64 (granularity of K)
 *   tile_p        packed-layout parameter (32/64) - fixes the wire format
 *   stages        decode: which of the planner's ranked (waves, K split) shapes to launch (2 = best,
 *                 3/4/5 = the next ones); MFMA kernel: the neighbouring in-workgroup K splits
 *   lut_copies    the reference's QuantMapMode slot (1/32/16/8): MFMA kernel, 4-bit: automatic / no lane sharing
 *                 above M = 16 (the grid K split fills the chip instead) / one / two slabs per wave (above M = 16 the
 *                 last one also means no lane sharing: two slabs per wave x the grid K split).  The kernels
 *                 always replicate the pair table 32x in LDS.
 * The decode kernel applies the group scale in fp32 to an 8-k partial sum: exact on
 * one-hot inputs, within 2^-11 relative per term of the reference's round_T(lut * s) otherwise. */
typedef struct flute_template_info {
    int num_bits, template_id;
    int sms_multiple, threads, tile_m, tile_k, tile_p, stages, lut_copies;
} flute_template_info;

/* Launch plan chosen for a problem (host logic only, no GPU needed). */
typedef struct flute_plan {
    int family;          /* 0 = decode (GEMV kernels, M<=4; 3 bits: M<=2; see one_shot), 2 = MFMA kernel with
                            LDS-DMA staged operands (every larger M), 3 = block-tiled prefill kernel
                            (enough 128 / 256 x 256 output blocks to fill the chip; 3-bit layers from M = 65 also 128- or
                            64-row blocks x splitk K slices - m_block 5 / 12; 128-row blocks x 2 / 4 slices meet inside the launch (splitk_mode 1), the others through fp32 slabs + the reduce pass), 5 = skinny MFMA kernel
                            (qgemm_skinny.h: 4-bit, 3 <= M <= 16, K = 32 x ring_depth x waves, layers whose 64-column
                            slabs fill 55..100 % of the CUs; weights and activations straight to registers),
                            6 = split-K block kernel (qgemm_splitk.h: 2- / 4-bit, m_tiles x 16 rows (128 or 64) x 256 / kw columns
                            (128; 64 with kw = 4 K parts per workgroup, 64-row tiles only - round 6) output tiles x splitk K slices,
                            one workgroup of 8 compute + 4 loader waves each, partial tiles combined inside the launch; automatic
                            from M = 33 (4 bits) / 65 (2 bits) where its modelled time is 8 % under the other MFMA kernels' (64-column
                            tiles that fill half the chip: under the per-wave kernel's time + 4 us) and one round of workgroups
                            covers the output),
                            7 = lean MFMA decode kernel (qgemm_fastm.h, round 5: 4 bits, 5 <= M <= 16, K in {2048, 4096}, a
                            workgroup = 4 unit rows x all of K, N / 16 workgroups of 8 waves between half a round and one
                            round of the CUs, 32 KB + 32 copies x 4 KB of LDS = 160 KB; what it cannot take falls back),
                            8 = persistent MFMA decode kernel (qgemm_persistm.h, round 6: 4 or 2 bits, 3 <= M <= 16, K % 128 == 0, group size
                            64 / 128; `grid` workgroups of 8 waves stream `visits` sets of slabs_per_wave column groups (16 columns each) x
                            all of K, k_chunks = 1 / 2 / 4 activation requests per 128-k macro-step for M <= 4 / 8 / 16; automatic under the
                            ids that leave the choice to the planner for layers above 16 M weights with K >= 6144, K >= 3584 at M <= 8 or
                            where K is neither 2048 nor 4096) */
    int m_block;         /* decode: rows per pass (1/2/4); family 2: R (lanes sharing a unit); family 3: block shape
                            (4 / 5: 256- / 128-row blocks; 8 + rt: 3-bit blocks of rt = 1, 2, 4 row tiles); family 6: row tiles of a
                            column tile that run as consecutive blocks of ONE XCD when K is not split (1 = natural order, 2, 4, 8);
                            families 7, 8: 16 */
    int m_tiles;         /* family 2: 16-row tiles per wave (1/2/4); family 6: row tiles per output tile (8 / 4) */
    int slabs_per_wave;  /* family 2: 16-unit column slabs per wave (1/2); family 7: column groups per workgroup; family 8: per set (1 .. 3) */
    int waves;           /* waves per workgroup (decode: any count up to 16, not only powers of two) */
    int kw;              /* waves of a workgroup sharing one unit (in-workgroup K split); family 6: K parts per workgroup (2 / 4) */
    int splitk;          /* grid-level K split (fp32 slabs in the workspace; see splitk_mode) */
    int k_per_split;
    int lut_copies;
    unsigned grid, block;
    size_t lds_bytes;
    size_t workspace_needed;
    int ring_depth;      /* decode: 1-KiB weight pieces in flight per wave (ring kernel 2/4; one-shot kernels: pieces per wave
                            4/8, 3 bits 2/4; persistent one-shot kernel: pieces per segment); skinny MFMA kernel: k-steps per wave */
    int visits;          /* decode: unit groups the busiest workgroup streams; family 8: sets the busiest workgroup streams */
    int k_chunks;        /* decode: passes over K when the activations do not fit in LDS at once; family 8: activation requests per macro-step */
    int one_shot;        /* family 8: 1 = the activations resident in LDS (4 k_chunks rows x K within 64 KB), 0 = through the wave-private rings;
                            decode: 0 = persistent ring kernel (qgemm_stream.h); 1 = one-shot kernel (qgemm_oneshot.h:
                            non-persistent workgroups, every request issued by the prologue, ring_depth = pieces per
                            wave), 2 = the same with the software-pipelined piece loop, 3 = persistent one-shot kernel
                            (qgemm_persist.h: table / activations staged once, every wave walks `visits` units of
                            `k_chunks` segments of ring_depth pieces, the next segment requested ahead), 4 = lean decode
                            kernel (qgemm_fast.h, round 5: 4 bits, M <= 4, K = 512 * ring_depth * kw in {2048, 3584, 4096, 8192}
                            a compile-time constant, m_block rows per pass) */
    int splitk_mode;     /* splitk > 1: 0 = fp32 slabs in the workspace + a second (reduce) launch, 1 = combined inside the
                            launch (csrc/xwg.h: write-through slabs + one arrival word per output tile) */
} flute_plan;

/* Per-call launch-plan overrides for the offline tuner, the sweeps and the tests; every field -1 (or a
 * NULL pointer) = automatic.  Plain data passed with the call: there is no process-global tuning state.
 *   family          5 skinny MFMA kernel (4-bit, M <= 16; waves 4 / 8 picks the in-workgroup K split);
 *                   6 split-K block kernel (splitk picks the K slices per tile; 1 = none; m_tiles 8 / 4: 128- / 64-row
 *                   tiles; kw 2 / 4: K parts per workgroup = 128- / 64-column tiles (4 with 64-row tiles only); m_block 1 / 2 / 4 / 8:
 *                   row tiles per XCD group of the block order; waves must be 12 or automatic - the variant without loader
 *                   waves was dropped in round 6);
 *                   7 lean MFMA decode kernel (4 bits, 5 <= M <= 16, K in {2048, 4096}; falls back where it does not apply);
 *                   8 persistent MFMA decode kernel (4 / 2 bits, M <= 16, K % 128 == 0, K >= 1024, group size 64 / 128; slabs_per_wave 1 .. 3:
 *                   column groups per set, m_tiles: sets per workgroup, one_shot 0: activation rings also where the activations could be resident;
 *                   refused - FLUTE_ERR_SHAPE - where it does not apply);
 *                   0 decode kernels also at M = 3, 4 (2- / 4-bit; automatic: M <= 2, and M <= 4 for
 *                   small layers called with a Hadamard size, to keep the rotation fused), 1 / 2 per-wave MFMA
 *                   kernel, 3 block-tiled prefill kernel (m_tiles 8 / 4: 256 / 128-row block); 4 and > 8 are rejected
 *                   (FLUTE_ERR_SHAPE).  m_tiles / waves / kw / splitk / slabs_per_wave given WITHOUT family = 6 belong to the
 *                   per-wave kernel: such a call never takes the split-K block kernel automatically
 *   m_block         decode: rows per pass; MFMA: R (lanes sharing a unit)
 *   waves, kw       waves per workgroup / in-workgroup K split
 *   splitk          grid-level K split
 *   m_tiles, slabs_per_wave   MFMA kernel: 16-row tiles per wave (1/2/4), column slabs per wave (1/2)
 *   ring_depth      decode: pieces in flight per wave (ring kernel 2/4; one-shot kernel 4/8, 3-bit 2/4); without
 *                   one_shot = 1 a given depth selects the ring kernel
 *   one_shot        decode: 1 one-shot kernel, 0 persistent ring kernel, 3 persistent one-shot kernel (M <= 2) - the code
 *                   flute_plan.one_shot reports for it; 2, ABI v4's value for the same request, is still accepted; 4 lean decode
 *                   kernel (4 bits, M <= 4, K in {2048, 3584, 4096, 8192}; `waves` 4 / 8 picks its shape; what it cannot take falls back) */
typedef struct flute_overrides {
    int family, m_block, waves, kw, splitk, m_tiles, slabs_per_wave, ring_depth, one_shot;
} flute_overrides;

/* D[M,N] = A[M,K] @ (table2-lookup(Q) * S)   fused LUT-dequant GEMM.
 * Replaces _qgemm_raw<T,TQ,T2,NumBits,GroupSize>  (flute/csrc/qgemm.cpp:15-36,
 * body flute/csrc/qgemm_kernel_raw_generated.cu:15-768 -> qgemm_host,
 * qgemm_kernel.hpp:824-939).
 *   A  [M,K] T row-major          Q  [P,K] int16, P = num_bits*N/16 (flute/utils.py:59-253)
 *   S  [N,K/group_size] T         QM [2^b] T (unused by the kernel, as in the reference)
 *   QM2 [2^b,2^b] pairs of T in one 32-bit word (flute/utils.py:15-33)
 *   workspace: caller-owned scratch, used for split-K slabs; may be NULL when
 *   the plan has splitk == 1 (flute/utils.py:36-56 over-allocates it). */
int flute_qgemm(int dtype, int num_bits, int group_size, int M, int N, int K, int P,
                const void* A, const void* Q, void* D, const void* S, const void* QM,
                const void* QM2, void* workspace, size_t workspace_bytes, int template_id,
                int num_sms, void* stream);

/* flute.qgemm_hadamard (flute/__init__.py:32-50; apply_hadamard + qgemm_raw_simple_hadamard,
 * flute/csrc/qgemm.cpp:201-244): D = (A.reshape(-1, hadamard_size) @ H/sqrt(hadamard_size)).reshape(M,K)
 * @ dequant(Q).  When the launch plan is a decode kernel, hadamard_size <= 512 divides K and M * K <= 8192 elements
 * (every workgroup rotates all M rows for itself: beyond that a separate flute_hadamard launch is cheaper - measured),
 * the rotation is fused into that kernel's activation staging (one launch, no round trip of the
 * rotated activations through HBM; same fp32 butterflies and single rounding as flute_hadamard, so the
 * result is bit-identical to the two-launch form).  Otherwise the rotated activations go to
 * x_scratch ([M,K] T, caller-owned) with flute_hadamard and the plain product follows.
 * flute_qgemm_hadamard_fused returns 1 when x_scratch will not be touched (may then be NULL). */
int flute_qgemm_hadamard(int dtype, int num_bits, int group_size, int hadamard_size, int M, int N,
                         int K, int P, const void* A, const void* Q, void* D, const void* S,
                         const void* QM, const void* QM2, void* x_scratch, void* workspace,
                         size_t workspace_bytes, int template_id, int num_sms, void* stream);
int flute_qgemm_hadamard_fused(int dtype, int num_bits, int group_size, int hadamard_size, int M,
                               int N, int K, int template_id, int num_sms, size_t workspace_bytes);

/* flute_qgemm_hadamard with a per-call plan override (ovr may be NULL): what the offline tuner and the
 * development sweeps call.  hadamard_size 0 = plain flute_qgemm. */
int flute_qgemm_ex(int dtype, int num_bits, int group_size, int hadamard_size, int M, int N, int K, int P,
                   const void* A, const void* Q, void* D, const void* S, const void* QM,
                   const void* QM2, void* x_scratch, void* workspace, size_t workspace_bytes,
                   int template_id, int num_sms, const flute_overrides* ovr, void* stream);

/* The plan flute_qgemm would use (exposed for tests and the offline tuner). */
int flute_qgemm_plan(int dtype, int num_bits, int group_size, int M, int N, int K,
                     int template_id, int num_sms, size_t workspace_bytes, flute_plan* out);
int flute_qgemm_plan_ex(int dtype, int num_bits, int group_size, int M, int N, int K,
                        int template_id, int num_sms, size_t workspace_bytes,
                        const flute_overrides* ovr, flute_plan* out);

/* out = in.reshape(-1, had_size) @ (H/sqrt(had_size)), Sylvester order.
 * Replaces run_fht<dtype>(a, out, numel, had_size, stream)
 * (flute/csrc/hadamard_transform_cuda.cu:701-748; wrapper hadamard_transform.cpp:17-56;
 * used by apply_hadamard, qgemm.cpp:201-211).  in == out is allowed. */
int flute_hadamard(int dtype, const void* in, void* out, size_t numel, uint32_t had_size,
                   void* stream);

/* Q[P,K] -> integer codes W[K,N] uint8 on the device.  Native replacement for
 * flute.utils.unpack, which runs qgemm on an identity matrix
 * (flute/utils.py:347-407). */
int flute_unpack(int num_bits, int template_id, int N, int K, const void* Q, void* W,
                 void* stream);

/* Template table (replaces data/qgemm_kernel_raw_generated_configs.pth +
 * the generated switch, qgemm_kernel_raw_generated.cu:92-767). */
int flute_num_templates(int num_bits);
int flute_get_template_info(int num_bits, int template_id, flute_template_info* out);

/* Calibration only: stream `bytes` from `src` with the decode kernel's access shape and
 * no arithmetic (what the HBM path alone costs for a given byte count). */
int flute_debug_stream_read(const void* src, void* sink, size_t bytes, int bytes_per_wave,
                            int grid, int block, void* stream);

/* Measurement only: enqueue a one-lane kernel that writes the chip-wide 100 MHz clock (ticks of 10 ns) to the
 * 8 bytes at `dst` (device memory).  Capturable into a hipGraph: two of them bracket exactly the launches between. */
int flute_debug_timestamp(void* dst, void* stream);

const char* flute_strerror(int status);
int flute_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* FLUTE_AMD_H */
