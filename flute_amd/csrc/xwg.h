// Cross-workgroup reduction of split-K partial tiles INSIDE one launch (round 4).
//
// gfx950 counterpart of the reference's Stream-K fix-up (flute/csrc/tile_scheduler_utils.hpp:58-211: FixupHelper -
// partial accumulators in the workspace, flag barriers, the workspace left clean for the next call, :196).  Until
// round 3 a grid-level K split here meant fp32 slabs + a SECOND launch (>= 2 us of kernel boundary, the slabs dirty in
// L2 behind it); the one in-launch attempt used agent-scope release fences (an L2 write-back per workgroup on this
// 8-die part: 51 us at four splits).  This header is the cheap form MI355X_MICROARCH.md prices (rows handoff-flag,
// publish-large, splitk-seam):
//   * partial tiles are stored WRITE-THROUGH (`buffer_store_dwordx4 ... sc1`): no release fence, no L2 write-back;
//   * every storing wave drains its own stores (`s_waitcnt vmcnt(0)`), the workgroup meets at a barrier, ONE lane
//     arrives with a relaxed agent-scope atomic on the tile's state word;
//   * whoever combines reads the other slices' partials with `sc1` loads (they bypass the reader's L1; the lines were
//     never left in the writers' L2) - no acquire fence;
//   * nothing depends on dispatch order or on where a workgroup runs, and nobody ever waits for a workgroup that has not
//     ARRIVED (arrived = running): the bounded poll below may give up, it cannot deadlock;
//   * the state words are zero again when the launch ends (the last arriver resets them): a hipGraph replays the launch
//     without a memset node, the caller's zero-filled workspace (flute/utils.py:36-45) stays clean.
//
// Two forms, chosen by the kernel:
//   L ("last arriver combines"): every slice publishes its whole partial tile; the workgroup whose arrival completes the
//     count adds the others to its registers and writes D.  For small tiles (M <= 16 slabs: 4 KB per slice).
//   E ("every slice combines its share"): the tile is cut into `nsh` = splitk shares, slice s owns share s, publishes
//     the OTHER shares only, waits (bounded) until all slices have arrived, claims its share, adds the others' partials
//     of it to its own registers and writes that part of D.  A slice whose wait runs out publishes its own share too and
//     marks it abandoned; the last arriver - which by construction finds every partial in place - combines abandoned
//     shares after its own.  For large tiles (128 x 128 fp32 = 64 KB per slice: one round trip of 48 KB per workgroup
//     instead of 192 KB through the last arriver).
//
// State per output tile, two words in the FIRST kXwgFlagBytes of the caller's workspace (the slabs start behind them):
//   w0: arrivals;   w1: bit q = share q claimed by its owner, bit 16 + q = share q abandoned (published in full).
#pragma once
#include <utility>

#include "common.h"

namespace flute_amd {

constexpr size_t kXwgFlagBytes = 64 * 1024;
constexpr int kXwgMaxTiles = (int)(kXwgFlagBytes / 8);
#ifndef FLUTE_XWG_SPIN_LIMIT
#define FLUTE_XWG_SPIN_LIMIT 4096     // development builds set 0 / 1: every owner that is not last gives up at once - the abandon / sweep path under test
#endif
constexpr unsigned kXwgSpinLimit = FLUTE_XWG_SPIN_LIMIT;   // polls of ~0.3-1 us each before an owner gives its share up

typedef uint32_t xwg_u32x4 __attribute__((ext_vector_type(4)));
// state words are addressed as GLOBAL memory (global_atomic_*, not flat_*: MI355X_MICROARCH.md, Guideline 16)
typedef __attribute__((address_space(1))) uint32_t xwg_word;
__device__ __forceinline__ xwg_word* xwg_state(uint32_t* p) { return (xwg_word*)p; }

__device__ __forceinline__ __amdgpu_buffer_rsrc_t xwg_rsrc(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
// 16-B write-through store / L1-bypassing load at byte offset `off` of the slab region (aux 16 = sc1)
__device__ __forceinline__ void xwg_store(f32x4_t v, __amdgpu_buffer_rsrc_t r, uint32_t off) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(xwg_u32x4, v), r, (int)off, 0, 16);
}
__device__ __forceinline__ f32x4_t xwg_load(__amdgpu_buffer_rsrc_t r, uint32_t off) {
    return __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 16));
}

// Publish point.  EVERY wave of the workgroup calls this after its last partial store: drains the wave's stores, meets
// the others, lane 0 of the workgroup arrives; returns the number of slices that had arrived before (wave-uniform).
// `bcast` = a free LDS dword (byte address): the barrier inside also retires every earlier LDS use of the workgroup.
__device__ __forceinline__ uint32_t xwg_arrive(xwg_word* st, uint32_t bcast, int tid) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        const uint32_t before = __hip_atomic_fetch_add(st, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *(__attribute__((address_space(3))) uint32_t*)(uintptr_t)bcast = before;
    }
    __syncthreads();
    return __builtin_amdgcn_readfirstlane(lds_ld32(bcast));
}

// E form, an owner that was not last: one lane polls the arrival count (relaxed, with s_sleep); the other waves park at
// the barrier.  Returns 1 when all `nsl` slices have arrived, 0 when the poll ran out.
__device__ __forceinline__ uint32_t xwg_wait_all(xwg_word* st, uint32_t nsl, uint32_t bcast, int tid) {
    if (tid == 0) {
        uint32_t ok = 0;
        for (unsigned spin = 0; spin < kXwgSpinLimit; ++spin) {
            if ((__hip_atomic_load(st, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 0xffu) >= nsl) { ok = 1; break; }
            __builtin_amdgcn_s_sleep(8);
        }
        *(__attribute__((address_space(3))) uint32_t*)(uintptr_t)bcast = ok;
    }
    __syncthreads();
    const uint32_t ok = __builtin_amdgcn_readfirstlane(lds_ld32(bcast));
    __syncthreads();                              // `bcast` may be reused
    return ok;
}
__device__ __forceinline__ void xwg_claim(xwg_word* st, uint32_t share, int tid) {
    if (tid == 0) (void)__hip_atomic_fetch_or(st + 1, 1u << share, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// after the abandoned share's own partial has been stored by every wave
__device__ __forceinline__ void xwg_abandon(xwg_word* st, uint32_t share, int tid) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) (void)__hip_atomic_fetch_or(st + 1, 0x10000u << share, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// E form, the last arriver after its own share: waits until every other share is claimed or abandoned (their owners
// have all arrived, i.e. are running a bounded poll: this terminates) and returns the abandoned set (bit q).
__device__ __forceinline__ uint32_t xwg_sweep(xwg_word* st, uint32_t nsl, uint32_t mine, uint32_t bcast, int tid) {
    if (tid == 0) {
        const uint32_t want = ((1u << nsl) - 1u) & ~(1u << mine);
        uint32_t w1 = 0;
        // every owner's poll is bounded, so this loop ends by itself; the bound only guards against a hung chip.  Expiry is FATAL
        // (the launch traps and the stream reports an error): carrying on with an incomplete abandoned set would skip shares nobody
        // combined and reset state words an owner may still OR into - silent, persistent corruption of the shared workspace.
        bool done = false;
        // (~16 M polls, tens of seconds: an owner that was merely preempted - a debugger, a time-sliced GPU - is waited for)
        for (unsigned spin = 0; spin < ((kXwgSpinLimit > 4096u ? kXwgSpinLimit : 4096u) << 12); ++spin) {
            w1 = __hip_atomic_load(st + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((((w1 | (w1 >> 16)) & want) == want)) { done = true; break; }
            __builtin_amdgcn_s_sleep(8);
        }
        if (!done) __builtin_trap();
        *(__attribute__((address_space(3))) uint32_t*)(uintptr_t)bcast = (w1 >> 16) & want;
    }
    __syncthreads();
    const uint32_t ab = __builtin_amdgcn_readfirstlane(lds_ld32(bcast));
    __syncthreads();
    return ab;
}
// the last arriver, when nobody can touch the tile's state any more: leave it clean for the next launch
__device__ __forceinline__ void xwg_reset(xwg_word* st, int tid) {
    if (tid == 0) {
        __hip_atomic_store(st, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(st + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// The whole seam of one output tile for a kernel whose waves hold the tile as NR x NC accumulator fragments (f32x4 per lane:
// 1 KB per fragment and wave), NW waves per workgroup - round 5: the form qgemm_splitk.h's epilogue spells out, as a function,
// for the 3-bit block kernel's K slices (qgemm_block3.h; until round 4 fp32 [M][N] slabs + a reduce LAUNCH).
// Slabs in FRAGMENT order, [slice][tile][wave][i][t][lane] x 16 B: every store / load instruction of a wave moves one contiguous
// KB and the same lane of the same wave of another slice finds its counterpart at the same place (the reference's
// BlockStripedReduce does the same, tile_scheduler_utils.hpp:80-83).  E form when the slice count (2 or 4) divides NR - share q =
// fragment rows q, q + nsl, ...; L form otherwise.  Sums in a fixed order (E: owner first, then the other slices ascending;
// L: all slices ascending): the result does not depend on who arrives when.  store_d(i, t, sum) writes fragment (i, t) of D.
// LDS dword 0 is the broadcast word: whatever the kernel kept there must be dead.  Every wave of the workgroup calls this.
template <int NR, int NC, int NW, typename StoreD>
__device__ __forceinline__ void xwg_seam(const f32x4_t (&own)[NR][NC], int nsl, int split, uint32_t tile, uint32_t ntiles,
                                         int wave, int lane, int tid, float* partial, uint32_t* state, StoreD&& store_d) {
    constexpr uint32_t TILE_SLAB = (uint32_t)NW * NR * NC * 1024u;
    const __amdgpu_buffer_rsrc_t slab = xwg_rsrc(partial, (uint32_t)nsl * ntiles * TILE_SLAB);
    const uint32_t slab_lane = tile * TILE_SLAB + (uint32_t)wave * (NR * NC * 1024u) + (uint32_t)lane * 16u;
    auto slab_off = [&](int slice, int i, int t) {
        return (uint32_t)slice * (ntiles * TILE_SLAB) + slab_lane + (uint32_t)(i * NC + t) * 1024u;
    };
    xwg_word* st = xwg_state(state + 2 * tile);
    const uint32_t bcast = 0;

    auto e_form = [&]<int NSH>(std::integral_constant<int, NSH>) {
        constexpr int PER = NR / NSH;
        const int me = split;
        f32x4_t mine[PER][NC];
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            if (i % NSH == me) {                                   // wave-uniform
#pragma unroll
                for (int t = 0; t < NC; ++t) mine[i / NSH][t] = own[i][t];
            } else {
#pragma unroll
                for (int t = 0; t < NC; ++t) xwg_store(own[i][t], slab, slab_off(me, i, t));
            }
        }
        // share `sh`: first + the other slices' partials in ascending slice order -> D
        auto combine = [&](int sh, const f32x4_t (&first)[PER][NC]) {
            f32x4_t ld[PER][NC][NSH - 1];
#pragma unroll
            for (int jj = 0; jj < PER; ++jj)
#pragma unroll
                for (int t = 0; t < NC; ++t)
#pragma unroll
                    for (int o = 0; o < NSH - 1; ++o) {
                        const int s2 = o + (o >= sh ? 1 : 0);      // the o-th slice other than the owner
                        ld[jj][t][o] = xwg_load(slab, slab_off(s2, sh + jj * NSH, t));
                    }
#pragma unroll
            for (int jj = 0; jj < PER; ++jj)
#pragma unroll
                for (int t = 0; t < NC; ++t) {
                    f32x4_t s = first[jj][t];
#pragma unroll
                    for (int o = 0; o < NSH - 1; ++o) s += ld[jj][t][o];
                    store_d(sh + jj * NSH, t, s);
                }
        };
        const uint32_t before = xwg_arrive(st, bcast, tid);
        if (before == (uint32_t)(NSH - 1)) {
            combine(me, mine);
            const uint32_t ab = xwg_sweep(st, NSH, me, bcast, tid);
            for (int q = 0; q < NSH; ++q) {
                if (!((ab >> q) & 1u)) continue;                   // abandoned by its owner: every slice's partial of it is in place
                f32x4_t first[PER][NC];
#pragma unroll
                for (int jj = 0; jj < PER; ++jj)
#pragma unroll
                    for (int t = 0; t < NC; ++t) first[jj][t] = xwg_load(slab, slab_off(q, q + jj * NSH, t));
                combine(q, first);
            }
            xwg_reset(st, tid);
        } else if (xwg_wait_all(st, NSH, bcast, tid)) {
            xwg_claim(st, me, tid);
            combine(me, mine);
        } else {
#pragma unroll
            for (int jj = 0; jj < PER; ++jj)
#pragma unroll
                for (int t = 0; t < NC; ++t) xwg_store(mine[jj][t], slab, slab_off(me, me + jj * NSH, t));
            xwg_abandon(st, me, tid);
        }
    };

    if (nsl == 4 && NR % 4 == 0) {
        if constexpr (NR % 4 == 0) e_form(std::integral_constant<int, 4>{});
    } else if (nsl == 2 && NR % 2 == 0) {
        if constexpr (NR % 2 == 0) e_form(std::integral_constant<int, 2>{});
    } else {
        // L form: every slice publishes its whole partial; the last arriver sums ALL slices in ascending order (its own from
        // the slab as well: one order whoever is last)
#pragma unroll
        for (int i = 0; i < NR; ++i)
#pragma unroll
            for (int t = 0; t < NC; ++t) xwg_store(own[i][t], slab, slab_off(split, i, t));
        const uint32_t before = xwg_arrive(st, bcast, tid);
        if (before == (uint32_t)(nsl - 1)) {
            f32x4_t s[NR][NC];
#pragma unroll
            for (int i = 0; i < NR; ++i)
#pragma unroll
                for (int t = 0; t < NC; ++t) s[i][t] = xwg_load(slab, slab_off(0, i, t));
            for (int s2 = 1; s2 < nsl; ++s2) {
                f32x4_t ld[NR][NC];
#pragma unroll
                for (int i = 0; i < NR; ++i)
#pragma unroll
                    for (int t = 0; t < NC; ++t) ld[i][t] = xwg_load(slab, slab_off(s2, i, t));
#pragma unroll
                for (int i = 0; i < NR; ++i)
#pragma unroll
                    for (int t = 0; t < NC; ++t) s[i][t] += ld[i][t];
            }
#pragma unroll
            for (int i = 0; i < NR; ++i)
#pragma unroll
                for (int t = 0; t < NC; ++t) store_d(i, t, s[i][t]);
            xwg_reset(st, tid);
        }
    }
}

}  // namespace flute_amd
