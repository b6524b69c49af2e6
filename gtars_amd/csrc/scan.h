// scan.h -- wave/workgroup scans and the chained ("decoupled look-back")
// cross-workgroup prefix sum shared by the fused enumerate kernels.
#pragma once
#include "common.h"

namespace gtars {

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains
// every outstanding global load/store of the wave (s_waitcnt vmcnt(0)), which
// would put the latency of in-flight result stores and prefetches on the
// critical path of every tile; the kernels here hand data between waves of a
// workgroup through LDS only.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Inclusive scan across the wave's 64 lanes on the VALU's data-parallel-primitive path: four shifts inside each row
// of 16 lanes, then the row totals handed on with row_bcast:15 / row_bcast:31 -- six v_add_u32_dpp and no LDS
// traffic (the __shfl_up form is six ds_bpermute round trips in a dependent chain).
__device__ __forceinline__ u32 wave_inclusive_scan_u32(u32 x, int lane) {
    (void)lane;
    // lanes without a source (row_shr past the row start, rows masked out) read `old` = 0
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);  // row_shr:1
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);  // row_shr:2
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);  // row_shr:4
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);  // row_shr:8
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1, 3
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2, 3
    return x;
}

// Inclusive running maximum across the wave for values >= 0 (0 is the identity the DPP fills in), same DPP steps.
__device__ __forceinline__ i32 wave_inclusive_max_nonneg(i32 x) {
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false));
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false));
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false));
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false));
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false));
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false));
    return x;
}

// Sum of one value < 2^48 per lane (the chained scan's granule payloads), every lane gets it: two 24-bit halves
// summed by DPP scans (64 * 2^24 fits 32 bits), no LDS round trips in the look-back's critical path.
__device__ __forceinline__ u64 wave_reduce_sum_u48(u64 x) {
    const u32 lo = wave_inclusive_scan_u32((u32)x & 0xFFFFFFu, 0);
    const u32 hi = wave_inclusive_scan_u32((u32)(x >> 24) & 0xFFFFFFu, 0);
    return (u64)__builtin_amdgcn_readlane(lo, 63) + ((u64)__builtin_amdgcn_readlane(hi, 63) << 24);
}

__device__ __forceinline__ u64 wave_reduce_sum_u64(u64 x) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) x += __shfl_xor(x, d, 64);
    return x;
}

// Exclusive scan of one u32 per thread across a TPB-thread workgroup; returns
// the exclusive prefix and the workgroup total.  lds: >= TPB/64 u32.
template <int TPB>
__device__ __forceinline__ u32 block_exclusive_scan(u32 x, u32 *lds, u32 &total) {
    constexpr int NW = TPB / 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u32 inc = wave_inclusive_scan_u32(x, lane);
    if (lane == 63) lds[wave] = inc;
    __syncthreads();
    u32 base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        const u32 v = lds[w];
        if (w < wave) base += v;
        tot += v;
    }
    total = tot;
    __syncthreads();
    return base + inc - x;
}

// ---- chained scan state ------------------------------------------------------
// One 8-byte granule per tile: status in the top two bits, value below.  Flag
// and payload travel in the same naturally aligned 64-bit word, written and
// read with relaxed agent-scope atomics (L1-bypassing), so no fence is needed
// and nothing depends on dispatch order or XCD placement.  Tiles are handed
// out by a ticket counter, so a tile only waits on tiles that already started.
//
// Bits 61..48 carry the launch's epoch: a granule written by an earlier launch
// reads as "not published", so a workspace that is reused launch after launch
// never has to be cleared (saves a memset kernel per call); bits 47..0 the value.
constexpr u64 ST_SHIFT = 62;
constexpr u64 ST_AGG = 1ull << ST_SHIFT;
constexpr u64 ST_INC = 2ull << ST_SHIFT;
constexpr u64 ST_MASK = 3ull << ST_SHIFT;
constexpr u64 EP_SHIFT = 48;
constexpr u32 EP_MAX = 0x3FFFu;
constexpr u64 VAL_MASK = (1ull << EP_SHIFT) - 1ull;
constexpr u32 LOOKBACK_SPIN_LIMIT = 1u << 22;

struct ScanWs {
    u32 ticket;
    u32 err;
    u64 total;
    u64 state[1];  // [num_tiles]
};

inline size_t scan_ws_bytes_for_tiles(u64 tiles) { return sizeof(u64) * (tiles + 3 + 16); }  // +16: debug words

__device__ __forceinline__ u64 ld_state(const u64 *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_state(u64 *p, u64 v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Executed by wave 0 (all 64 lanes); returns the tile's exclusive global prefix.
// Every round reads LB_W * 64 predecessor granules with independent loads (one
// memory round trip).  Measured on MI355X (tools/ablate.sh, EXTRA=-DGTARS_LB_W=n):
// with the cross-tile software pipeline of k_tok_lds the nearest inclusive prefix is
// usually inside the first window, and wider rounds only add loads of cold lines:
// LB_W = 1 / 2 / 4 / 8 / 16 -> 952 / 964 / 1003 / 1076 / 1194 us for 64M queries.
#ifndef GTARS_LB_W
#define GTARS_LB_W 1
#endif
constexpr int LB_W = GTARS_LB_W;

// publish a tile's aggregate (tile 0: its inclusive prefix) -- one lane
__device__ __forceinline__ void publish_aggregate(u64 *state, u32 tile, u64 agg, u32 epoch = 0) {
    st_state(&state[tile], (tile == 0 ? ST_INC : ST_AGG) | ((u64)epoch << EP_SHIFT) | agg);
}

// resolve the exclusive prefix of `tile` (whose aggregate is already
// published) and publish its inclusive prefix.  W = windows of 64 granules read per round
// (1 measured best both for software-pipelined tiles and for one-tile-per-workgroup launches:
// 4 windows cost 1M-query launches 19.9 us instead of 18.9).
template <int W = LB_W>
__device__ __forceinline__ u64 resolve_prefix(u64 *state, u32 tile, u64 agg, int lane, u32 *err, u32 epoch = 0) {
    if (tile == 0) return 0;
    u64 excl = 0;
    i64 pred = (i64)tile - 1;
    u32 spins = 0;
    bool done = false;
    while (!done) {
        u64 val[W];
#pragma unroll
        for (int w = 0; w < W; ++w) {
            const i64 idx = pred - (i64)w * 64 - lane;
            u64 v = ST_INC;  // before tile 0: inclusive 0
            if (idx >= 0) {
                v = ld_state(&state[idx]);
                if ((u32)((v >> EP_SHIFT) & EP_MAX) != epoch) v = 0;  // left over from an earlier launch
            }
            val[w] = v;
        }
        // consume the windows nearest-first; stop at the first window that is
        // not fully published up to its first inclusive entry
        int consumed = 0;
#pragma unroll
        for (int w = 0; w < W; ++w) {
            if (done || consumed != w) continue;
            const u64 status = val[w] & ST_MASK;
            const unsigned long long b_inc = __ballot(status == ST_INC);
            const unsigned long long b_inv = __ballot(status == 0);
            const int first_inc = b_inc ? __ffsll((long long)b_inc) - 1 : 64;
            const unsigned long long need = first_inc >= 63 ? ~0ull : ((1ull << (first_inc + 1)) - 1ull);
#if defined(GTARS_ABLATE) && (GTARS_ABLATE & 64)
            (void)b_inv; (void)need;  // timing experiment: never wait (results wrong)
#else
            if (b_inv & need) continue;  // not ready: re-read from this window on
#endif
            const u64 contrib = (lane <= first_inc) ? (val[w] & VAL_MASK) : 0ull;
            excl += wave_reduce_sum_u48(contrib);
            consumed = w + 1;
            if (first_inc < 64) done = true;
        }
        pred -= (i64)consumed * 64;
        if (!done && consumed == 0) {
            if (++spins > LOOKBACK_SPIN_LIMIT) {
                if (lane == 0) atomicOr(err, 1u);
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    if (lane == 0) st_state(&state[tile], ST_INC | ((u64)epoch << EP_SHIFT) | (excl + agg));
    return excl;
}

__device__ __forceinline__ u64 lookback(u64 *state, u32 tile, u64 agg, int lane, u32 *err) {
    if (lane == 0) publish_aggregate(state, tile, agg);
    return resolve_prefix(state, tile, agg, lane, err);
}

}  // namespace gtars
