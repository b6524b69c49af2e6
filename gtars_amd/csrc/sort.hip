// sort.hip -- K1: stable LSD radix sort of (u32 key, u32 payload) pairs on the device, used to build
// the indexes in the reference's stable orders:
//   Bits::build     sort by (start, end), ties in input order      (gtars-overlaprs/src/bits.rs:105)
//   Igd::finalize   sort by start, ties in insertion order         (gtars-igd/src/igd.rs:157-167)
//   per-chromosome bucketing = one more stable pass on the chromosome id
//                                                                  (gtars-tokenizers/src/utils/mod.rs:55-87)
// i.e. a per-chromosome segmented sort realised as stable passes from the least significant key up:
// end (4 x 8 bits), start (4 x 8 bits), chrom (ceil(bits/8)).  The payload is the original index, so
// stability == "ties in input order" by construction.
//
// Per 8-bit pass: (1) per-tile LDS histogram -> digit-major table, (2) exclusive scan of the table,
// (3) scatter with stable in-tile ranks: a wave finds its equal-digit peers with 8 ballots
// (64-wide "match-any"), waves are ordered through a small LDS table, rounds through a running base;
// the tile is reordered by digit in LDS first, so that it leaves as a few contiguous runs per wave.
// Bound: HBM, 16 B moved per element per pass (8 B in, 8 B out) + one 4-B random gather per key word.
#include <algorithm>
#include <mutex>

#include "common.h"
#include "scan.h"

namespace gtars {

constexpr int RS_TPB = 256;
constexpr int RS_ITEMS = 8;
constexpr int RS_TILE = RS_TPB * RS_ITEMS;
constexpr int RS_NW = RS_TPB / 64;

__global__ void __launch_bounds__(RS_TPB)
k_radix_hist(const u32 *__restrict__ keys, u32 n, int shift, u32 *__restrict__ table, u32 n_tiles) {
    __shared__ u32 h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const u32 base = blockIdx.x * RS_TILE;
#pragma unroll
    for (int r = 0; r < RS_ITEMS; ++r) {
        const u32 i = base + r * RS_TPB + threadIdx.x;
        if (i < n) atomicAdd(&h[(keys[i] >> shift) & 0xFFu], 1u);
    }
    __syncthreads();
    table[(size_t)threadIdx.x * n_tiles + blockIdx.x] = h[threadIdx.x];  // digit-major
}

__global__ void __launch_bounds__(RS_TPB)
k_radix_scatter(const u32 *__restrict__ keys, const u32 *__restrict__ vals, u32 n, int shift,
                const u64 *__restrict__ table_off, u32 n_tiles, u32 *__restrict__ okeys,
                u32 *__restrict__ ovals) {
    __shared__ u32 wc[RS_NW][256];  // per-wave digit counts of the current round
    __shared__ u32 run[256];        // elements of each digit already placed by earlier rounds
    __shared__ u32 toff[256];       // first slot of each digit inside the tile (tile-local exclusive scan)
    __shared__ u64 gbase[256];      // first global position of each digit's run of this tile
    __shared__ u32 s_k[RS_TILE], s_v[RS_TILE];  // the tile, reordered by digit before it is written out
    __shared__ u32 s_scan[RS_NW];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    run[threadIdx.x] = 0;
    {
        // table_off is the exclusive scan of the digit-major count table, so consecutive entries give the
        // count of (digit, tile) back; their block-wide exclusive scan is the tile-local layout
        const size_t idx = (size_t)threadIdx.x * n_tiles + blockIdx.x;
        const u64 o0 = table_off[idx], o1 = table_off[idx + 1];
        gbase[threadIdx.x] = o0;
        u32 total;
        toff[threadIdx.x] = block_exclusive_scan<RS_TPB>((u32)(o1 - o0), s_scan, total);
    }
    const u32 base = blockIdx.x * RS_TILE;
    for (int r = 0; r < RS_ITEMS; ++r) {
#pragma unroll
        for (int w = 0; w < RS_NW; ++w) wc[w][threadIdx.x] = 0;
        __syncthreads();
        const u32 i = base + r * RS_TPB + threadIdx.x;
        const bool ok = i < n;
        const u32 k = ok ? keys[i] : 0u;
        const u32 v = ok ? vals[i] : 0u;
        const u32 d = (k >> shift) & 0xFFu;
        // peers = lanes of this wave holding the same digit (and in range)
        unsigned long long peers = __ballot(ok);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const unsigned long long bal = __ballot((d >> b) & 1u);
            peers &= ((d >> b) & 1u) ? bal : ~bal;
        }
        const u32 rank_in_wave = __popcll(peers & ((1ull << lane) - 1ull));
        if (ok && rank_in_wave == 0) wc[wave][d] = __popcll(peers);  // one lane per digit writes the count
        __syncthreads();
        if (ok) {
            u32 before = run[d];
#pragma unroll
            for (int w = 0; w < RS_NW; ++w)
                if (w < wave) before += wc[w][d];
            const u32 slot = toff[d] + before + rank_in_wave;
            s_k[slot] = k;
            s_v[slot] = v;
        }
        __syncthreads();
        {
            u32 add = 0;
#pragma unroll
            for (int w = 0; w < RS_NW; ++w) add += wc[w][threadIdx.x];
            run[threadIdx.x] += add;
        }
        __syncthreads();
    }
    // write the tile out slot by slot: equal digits are adjacent, so a wave writes a few contiguous runs
    const u32 tile_n = min((u32)RS_TILE, n - base);
    for (u32 j = threadIdx.x; j < tile_n; j += RS_TPB) {
        const u32 k = s_k[j];
        const u32 d = (k >> shift) & 0xFFu;
        const u64 pos = gbase[d] + (j - toff[d]);
        okeys[pos] = k;
        ovals[pos] = s_v[j];
    }
}

// ---------------------------------------------------------------------------------------------- multisplit
// Partition by a small key in ONE pass over the data instead of one radix pass per 8 key bits: the IGD sweep only needs
// its queries grouped by owning tile (<= MS_MAX_BINS bins), in any order inside a group.  Every workgroup counts a
// contiguous chunk into LDS bins (one LDS atomic per element), a column scan turns the per-workgroup counts into
// per-workgroup cursors, and the second pass places every element with one more LDS atomic -- 10M queries into 24k
// bins: ~0.3 ms instead of 1.25 ms for five 8-bit passes (profiles/r02).  Not stable (atomics decide the order inside a
// bin); nothing downstream depends on it.
// What bounds k_ms_scatter (10M elements into 24k bins: 0.215 ms by rocprofv3): the scattered 8-byte stores -- with the same
// stores made contiguous the kernel takes a fifth of the time, without stores an eighth.  Random 8-byte accesses to an 80 MB
// region run at ~5e10 per second chip-wide, reads and writes alike; a first pass into <= 256 coarse bins with the same
// element-wise stores does not help (1.2x the time for both passes): a wave's 64 stores still go to ~55 different lines.
// Large splits therefore go through k_split_pass below (tiles reordered by bin in LDS, stores leave as runs): 0.12 ms.
constexpr int MS_TPB = 1024;

// Several query SETS in one partitioned batch (set k = rows [bound[k - 1], bound[k]) of the input, at most 4): the set of a row
// travels in the top bits of its (a, b) pair -- bit 31 of b = set & 1, bit 31 of a = set >> 1 -- which the values themselves
// never use (IGD coordinates are non-negative i32: igd.rs:514-517 rejects or clamps everything else before this point).
struct SetTags {
    u32 b1, b2, b3;  // first row of set 1, 2, 3 (0xFFFFFFFF: no such set)
    __device__ __forceinline__ void apply(u32 row, u32 &a, u32 &b) const {
        if (b1 == 0xFFFFFFFFu) return;  // (uniform) one set: nothing to tag
        const u32 set = (row >= b1 ? 1u : 0u) + (row >= b2 ? 1u : 0u) + (row >= b3 ? 1u : 0u);
        a |= (set >> 1) << 31;
        b |= (set & 1u) << 31;
    }
};

// per bin: exclusive prefix over the workgroups (in place) and the bin total; 16 independent loads per step
__global__ void k_ms_colscan(u32 *__restrict__ table, u32 n_wg, u32 n_bins, u32 *__restrict__ tot, const u32 *__restrict__ run_if) {
    if (run_if && *run_if == 0) return;  // the caller found nothing to partition (decided on the device)
    const u32 b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_bins) return;
    u32 run = 0;
    for (u32 w0 = 0; w0 < n_wg; w0 += 16) {
        u32 v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = w0 + k < n_wg ? table[(size_t)(w0 + k) * n_bins + b] : 0u;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (w0 + k < n_wg) table[(size_t)(w0 + k) * n_bins + b] = run;
            run += v[k];
        }
    }
    tot[b] = run;
}

// bin_off[0 .. n_bins] = exclusive scan of tot (one workgroup); bins far heavier than the average are also listed in pieces
// (HeavyBins, common.h: the consumer may hand the pieces of such a bin to several workgroups); for the two-level split (cur_a != null) also the cursors its
// passes reserve runs from: cur_b[i] = bin_off[i], cur_a[j] = bin_off[j << shift] (was a launch of its own)
template <u32 MAXS>  // bins per thread at most: 36 for the common sizes (<= MS_MAX_BINS), 64 up to MS_MAX_BINS_2L + 1
__global__ void __launch_bounds__(MS_TPB)
k_ms_binscan(const u32 *__restrict__ tot, u32 n_bins, u32 *__restrict__ bin_off, u32 shift, u32 *__restrict__ cur_a,
             u32 *__restrict__ cur_b, const u32 *__restrict__ run_if, HeavyBins heavy) {
    if (run_if && *run_if == 0) return;  // the caller found nothing to partition (decided on the device)
    // Every wave takes a contiguous run of S * 64 bins (S <= 64 for MS_MAX_BINS_2L + 1 bins): S coalesced loads in flight together, S wave
    // scans on the DPP path with a running carry, ONE barrier for the waves' totals (a round-per-1024-bins loop with two
    // barriers per round took 15 us for 24k bins).
    constexpr u32 NW = MS_TPB / 64;
    __shared__ u32 s_wtot[NW];
    const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u32 S = (n_bins + MS_TPB - 1) / MS_TPB, first = wave * S * 64u;
    u32 v[MAXS], carry = 0;
#pragma unroll
    for (u32 k = 0; k < MAXS; ++k) {
        const u32 b = first + k * 64u + lane;
        v[k] = k < S && b < n_bins ? tot[b] : 0u;
    }
#pragma unroll
    for (u32 k = 0; k < MAXS; ++k) {
        if (k < S) {  // (uniform)
            heavy.note(first + k * 64u + lane, v[k]);  // (the bin's own total, while it is still in the register)
            const u32 inc = wave_inclusive_scan_u32(v[k], (int)lane);
            v[k] = carry + inc - v[k];  // exclusive, inside the wave's run
            carry += __builtin_amdgcn_readlane(inc, 63);
        }
    }
    if (lane == 0) s_wtot[wave] = carry;
    __syncthreads();
    u32 base = 0, total = 0;
#pragma unroll
    for (u32 w = 0; w < NW; ++w) {
        const u32 x = s_wtot[w];
        base += w < wave ? x : 0u;
        total += x;
    }
#pragma unroll
    for (u32 k = 0; k < MAXS; ++k) {
        const u32 b = first + k * 64u + lane;
        if (k < S && b < n_bins) {
            const u32 o = base + v[k];
            bin_off[b] = o;
            if (cur_a) {
                cur_b[b] = o;
                if ((b & ((1u << shift) - 1u)) == 0) cur_a[b >> shift] = o;
            }
        }
    }
    if (threadIdx.x == 0) bin_off[n_bins] = total;
}

template <class KeyT, bool CLAMP>
__global__ void __launch_bounds__(MS_TPB)
k_ms_scatter(const KeyT *__restrict__ key, const u32 *__restrict__ a, const u32 *__restrict__ b, u32 n, u32 n_bins, u32 chunk,
             const u32 *__restrict__ table, const u32 *__restrict__ bin_off, u32 drop_bin, uint2 *__restrict__ out_ab, const u32 *__restrict__ run_if,
             SetTags tags) {
    if (run_if && *run_if == 0) return;  // the caller found nothing to partition (decided on the device)
    extern __shared__ u32 ms_bins[];
    const u32 *row = table + (size_t)blockIdx.x * n_bins;
    for (u32 k = threadIdx.x; k < n_bins; k += MS_TPB) ms_bins[k] = bin_off[k] + row[k];
    __syncthreads();
    const u32 lo = blockIdx.x * chunk, hi = min(n, lo + chunk);
    // 8 elements per thread and step: their loads, LDS atomics and stores overlap (one element per step is a chain of
    // a global load, an LDS round trip and two stores: 0.44 ms per 10M elements, latency-bound)
    constexpr int U = 8;
    for (u32 base = lo; base < hi; base += MS_TPB * U) {
        u32 k[U], va[U], vb[U], pos[U];
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const u32 i = base + (u32)j * MS_TPB + threadIdx.x;
            const bool ok = i < hi;
            k[j] = ok ? (u32)key[i] : drop_bin;
            va[j] = ok ? a[i] : 0u;
            vb[j] = ok ? b[i] : 0u;
            if (CLAMP) va[j] = (i32)va[j] < 0 ? 0u : va[j];
            tags.apply(i, va[j], vb[j]);
        }
#pragma unroll
        for (int j = 0; j < U; ++j) pos[j] = k[j] != drop_bin ? atomicAdd(&ms_bins[k[j]], 1u) : 0u;
#pragma unroll
        for (int j = 0; j < U; ++j) {
            if (k[j] != drop_bin) out_ab[pos[j]] = make_uint2(va[j], vb[j]);  // one scattered 8-byte store per element
        }
    }
}

u32 multisplit_workgroups(u32 n) {
    // one persistent-sized grid: chunks of at least 16k elements, at most 256 workgroups -- unless a chunk would then exceed what
    // a counting caller's 16-bit counters hold (65532 elements: batches beyond 16.7M elements take more workgroups, which queue)
    const u32 by_size = (n + 16383) / 16384;
    return std::max<u32>(std::max<u32>(1, std::min<u32>(256, by_size)), (u32)(((u64)n + 65531u) / 65532u));
}

// ---- two-level split with tiles reordered in LDS -------------------------------------------------------------
// k_ms_scatter stores every element on its own: 64 lanes, ~55 different lines, and random 8-byte accesses to a large
// region run at ~5e10 per second chip-wide.  Here a workgroup takes a tile of 8192 elements, ranks them by bin with LDS
// atomics, reserves a run per bin with ONE global atomic on the bin's cursor, reorders the tile by bin in LDS and writes
// it out slot by slot -- neighbouring lanes then store neighbouring elements of the same run.  Two passes keep the runs
// long: pass A splits by coarse bin (key >> shift, <= 256 bins: runs of ~40 elements), pass B splits each coarse segment by
// the full key -- a tile of pass A's output spans only a few coarse bins, i.e. <= SP_BINS neighbouring keys; an element
// outside that window (tiny coarse segments) is placed with its own global atomic.  Not stable.
// Geometry.  Pass A: 1024 threads, one workgroup per CU, tiles of 8192 elements (runs of ~32 elements per coarse bin: 256-byte
// stores; its prologue -- the column sums of the routing workgroups' rows -- wants the threads).  Pass B: 512 threads, tiles of 4096,
// TWO workgroups per CU in different phases of different tiles: every tile is a chain of load / rank / reserve / reorder / store
// round trips behind five barriers, which one workgroup per CU goes through with all its waves together (round 6, same-box A/B:
// pass B 48.7 -> 42.1 us per 10M pairs; pass A with 512 threads 53.7 -> 81.0, so it keeps 1024; profiles/r06/igd_split_geometry_ab.txt).
#ifndef GTARS_SP_TPB_A
#define GTARS_SP_TPB_A 1024
#endif
#ifndef GTARS_SP_TPB_B
#define GTARS_SP_TPB_B 512
#endif
constexpr int SP_TPB_A = GTARS_SP_TPB_A, SP_TPB_B = GTARS_SP_TPB_B;
#ifndef SP_STAMPS
#define SP_STAMPS 0  // diagnostic build: per-phase shader-clock totals of wave 0 of every workgroup (tools/r06_split_stamps.py)
#endif
#if SP_STAMPS
__device__ unsigned long long g_split_stamps[2][16];
#define SPSTAMP(k)                                   \
    do {                                             \
        const u64 _t = __builtin_amdgcn_s_memtime(); \
        sp_acc[k] += _t - sp_last;                   \
        sp_last = _t;                                \
    } while (0)
#else
#define SPSTAMP(k) \
    do {           \
    } while (0)
#endif
static_assert((SP_TPB_A == 1024 || SP_TPB_A == 512) && (SP_TPB_B == 1024 || SP_TPB_B == 512), "split-pass workgroups of 1024 or 512 threads");
constexpr int SP_ITEMS = 8;

// KeyT: u32, or unsigned short when every bin fits 16 bits (2 bytes per element less to read and write in both passes).
// CLAMP (first pass on raw query columns): a = max((i32)a, 0) -- the start clamp of Igd::count_overlaps (igd.rs:517) applied
// on the way, so that no prepared copy of the columns is ever written.
// The scan of the bin totals is part of the passes (no k_ms_binscan launch: 12 us + a launch for 24k bins).  The cursors hold
// RELATIVE counts (zeroed by the caller), and
//   pass A: every workgroup scans the <= 256 coarse totals (left by the caller's counting kernel) for itself: a base per coarse
//           bin in LDS -- a run is reserved at base + atomicAdd(relative cursor); workgroup w additionally sums the counting
//           workgroups' rows over the fine bins of coarse bin w, w + grid, ..., scans them and writes the fine offsets bin_off[]
//           (what pass B and the consumer read), and notes the heavy bins;
//   pass B: a run is reserved at bin_off[bin] + atomicAdd(relative cursor).
template <int TPB, bool FINE, class KeyT, bool CLAMP>
__global__ void __launch_bounds__(TPB, 4)
k_split_pass(const KeyT *__restrict__ key, const u32 *__restrict__ a, const u32 *__restrict__ b, const uint2 *__restrict__ ab_in, u32 n,
             u32 shift, u32 drop_bin, u32 *__restrict__ cursor, KeyT *__restrict__ out_key, uint2 *__restrict__ out_ab, const u32 *__restrict__ run_if,
             SetTags tags, const u32 *__restrict__ rows = nullptr, u32 n_bins = 0, u32 *__restrict__ bin_off = nullptr, HeavyBins heavy = HeavyBins{},
             const u32 *__restrict__ ctot = nullptr, u32 n_rows = 0, uint2 *__restrict__ trash = nullptr) {
    if (run_if && *run_if == 0) return;  // the caller found nothing to partition (decided on the device)
    constexpr int TILE_ = TPB * SP_ITEMS, BINS_ = TPB;  // one thread per tile-local bin (counter reset, layout scan, run reservation)
    extern __shared__ u32 sp_lds[];
    u32 *s_k = sp_lds, *s_a = s_k + TILE_, *s_b = s_a + TILE_;
    u32 *cnt = s_b + TILE_, *toff = cnt + BINS_, *gbase = toff + BINS_;
    __shared__ u32 s_scan[TPB / 64];
    __shared__ u32 s_cbase[!FINE ? 257 : 1];  // exclusive offsets of the coarse bins (+ the grand total)
    const u32 n_tiles = (n + TILE_ - 1) / TILE_;
    // One workgroup per CU (its tile fills the LDS), so nothing else hides a tile's memory latencies: the NEXT tile's elements
    // are requested into a second set of registers before the current tile is ranked (in flight during ranking, reservation,
    // LDS reorder and write-out), and a bin's run reservation -- a global atomic WITH return -- is only waited for after the
    // tile has been reordered in LDS (the reorder needs the tile-local layout, not the global base).  Before: every tile paid a
    // load round trip and an atomic round trip back to back, 13 us per 8192-element tile (164 KB moved: 12.6 GB/s per CU).
    u32 nk[SP_ITEMS], na[SP_ITEMS], nb[SP_ITEMS], nfirst = 0;
    auto request = [&](u32 tile) {
        const u32 base = tile * TILE_;
        if (FINE) nfirst = (u32)key[base];  // (every lane the same address: the tile's first element, see bin0 below)
#pragma unroll
        for (int j = 0; j < SP_ITEMS; ++j) {
            const u32 i = base + (u32)j * TPB + threadIdx.x;
            const u32 ic = i < n ? i : n - 1u;  // (unconditional loads: the wait counts stay exact; the copy is ignored)
            nk[j] = (u32)key[ic];
            if (FINE) {
                const uint2 p = ab_in[ic];
                na[j] = p.x;
                nb[j] = p.y;
            } else {
                na[j] = a[ic];
                nb[j] = b[ic];
            }
        }
    };
#if SP_STAMPS
    u64 sp_acc[12] = {}, sp_last = __builtin_amdgcn_s_memtime();
#endif
    if (blockIdx.x < n_tiles) request(blockIdx.x);
    if constexpr (!FINE) {
        // the coarse totals were left by the caller's counting kernel (multisplit_coarse_totals)
        const u32 n_coarse = ((n_bins - 1u) >> shift) + 1u;
        u32 grand;
        const u32 mine = threadIdx.x < n_coarse ? ctot[threadIdx.x] : 0u;  // coarse bin threadIdx.x
        const u32 ex = block_exclusive_scan<TPB>(mine, s_scan, grand);
        if (threadIdx.x < 256u) s_cbase[threadIdx.x] = ex;
        if (threadIdx.x == 0) s_cbase[256] = grand;
        __syncthreads();
        // fine offsets of this workgroup's coarse bins (2^shift <= 256 fine bins each) + the heavy-bin list
        for (u32 jc = blockIdx.x; jc < n_coarse; jc += gridDim.x) {
            // the fine totals of coarse bin jc = the column sums of the counting workgroups' rows (16-bit counts, two per word):
            // 2^shift / 8 16-byte vectors per row, as many rows side by side as the workgroup has threads for, partial sums
            // combined in LDS
            const u32 vpr = (1u << shift) >> 3, rw = multisplit_row_words(n_bins), w0 = (jc << shift) >> 1;  // (shift >= 3)
            cnt[threadIdx.x] = 0;  // (2^shift <= 256 <= BINS_ counters)
            __syncthreads();
            const u32 vec = threadIdx.x % vpr, wv = w0 + 4u * vec;
            for (u32 r0 = 0; r0 < n_rows; r0 += 4u * (TPB / vpr)) {
                uint4 x[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const u32 r = r0 + (u32)q * (TPB / vpr) + threadIdx.x / vpr;
                    x[q] = r < n_rows && wv < rw ? *reinterpret_cast<const uint4 *>(rows + (size_t)r * rw + wv) : make_uint4(0, 0, 0, 0);
                }
                u32 lo[4] = {0, 0, 0, 0}, hi[4] = {0, 0, 0, 0};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    lo[0] += x[q].x & 0xFFFFu, hi[0] += x[q].x >> 16;
                    lo[1] += x[q].y & 0xFFFFu, hi[1] += x[q].y >> 16;
                    lo[2] += x[q].z & 0xFFFFu, hi[2] += x[q].z >> 16;
                    lo[3] += x[q].w & 0xFFFFu, hi[3] += x[q].w >> 16;
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (lo[k]) atomicAdd(&cnt[8u * vec + 2u * k], lo[k]);
                    if (hi[k]) atomicAdd(&cnt[8u * vec + 2u * k + 1u], hi[k]);
                }
            }
            __syncthreads();
            const u32 fb = (jc << shift) + threadIdx.x;
            const bool in = threadIdx.x < (1u << shift) && fb < n_bins;
            const u32 v = in ? cnt[threadIdx.x] : 0u;
            u32 tt;
            const u32 fe = block_exclusive_scan<TPB>(v, s_scan, tt);
            if (in) {
                bin_off[fb] = s_cbase[jc] + fe;
                heavy.note(fb, v);
            }
        }
        if (blockIdx.x == 0 && threadIdx.x == 0) bin_off[n_bins] = s_cbase[256];
        __syncthreads();  // cnt is the tile loop's again
    }
    SPSTAMP(0);  // prologue
    for (u32 tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const u32 base = tile * TILE_;
        cnt[threadIdx.x] = 0;  // BINS_ == TPB
        u32 k[SP_ITEMS], va[SP_ITEMS], vb[SP_ITEMS], lb[SP_ITEMS], rank[SP_ITEMS];
        // FINE: the tile's window of keys starts at its smallest coarse bin -- the coarse bin of its FIRST element: the first pass
        // left the elements grouped by coarse bin, ascending.  (Round 6: the minimum over the tile's keys -- a wave reduction, an LDS
        // atomic and two barriers -- was a sixth of this pass by the in-kernel stamps.  Elements of a coarse bin below the first
        // element's cannot exist; a first element that is dropped only makes the window start no later than it must.)
        const u32 bin0 = FINE ? (nfirst >> shift) << shift : 0u;
#pragma unroll
        for (int j = 0; j < SP_ITEMS; ++j) {
            const u32 i = base + (u32)j * TPB + threadIdx.x;
            const bool ok = i < n;
            k[j] = ok ? nk[j] : 0xFFFFFFFFu;
            va[j] = ok ? na[j] : 0u;
            vb[j] = ok ? nb[j] : 0u;
            if (FINE) {
                // no owner: dropped HERE, not in the first pass -- that pass must fill all n slots of its output (the drop
                // bin's coarse segment included), or this one would read whatever an earlier call left behind the kept ones
                if (k[j] == drop_bin) k[j] = 0xFFFFFFFFu;
            } else {
                if (CLAMP) va[j] = (i32)va[j] < 0 ? 0u : va[j];
                tags.apply(i, va[j], vb[j]);  // (the second pass carries the pairs as they are)
            }
        }
        SPSTAMP(1);  // the tile's elements (wait for the prefetch)
        if (tile + gridDim.x < n_tiles) request(tile + gridDim.x);
        __syncthreads();  // cnt zeroed
#pragma unroll
        for (int j = 0; j < SP_ITEMS; ++j) {
            lb[j] = 0xFFFFFFFFu;
            rank[j] = 0;
            if (k[j] == 0xFFFFFFFFu) continue;
            const u32 x = FINE ? k[j] - bin0 : k[j] >> shift;
            if (!FINE || x < (u32)BINS_) {  // (pass A: <= 256 coarse bins)
                lb[j] = x;
                rank[j] = atomicAdd(&cnt[x], 1u);
            } else {
                // outside the tile's window (FINE only): its own slot from the bin's cursor
                out_ab[bin_off[k[j]] + atomicAdd(&cursor[k[j]], 1u)] = make_uint2(va[j], vb[j]);
            }
        }
        SPSTAMP(2);  // window + rank (LDS atomics)
        __syncthreads();
        SPSTAMP(3);  // barrier
        u32 my_cnt, my_rel0 = 0, my_got = 0;
        {
            // tile-local layout + one reserved run per non-empty bin.  The reservation's answer is picked up after the reorder: the
            // sum base + answer is formed THERE -- formed inside this branch, the compiler waited for the atomic's round trip here
            // (round 6, from the listing: global_atomic_add / s_waitcnt vmcnt(0) back to back).
            my_cnt = cnt[threadIdx.x];
            u32 total;
            toff[threadIdx.x] = block_exclusive_scan<TPB>(my_cnt, s_scan, total);
            if (my_cnt) {
                const u32 bin = FINE ? bin0 + threadIdx.x : threadIdx.x;
                if constexpr (FINE)
                    my_rel0 = bin_off[bin];
                else
                    my_rel0 = s_cbase[bin];
                my_got = atomicAdd(&cursor[bin], my_cnt);
            }
        }
        SPSTAMP(4);  // layout scan + reservation issued
        __syncthreads();
        {
            // (the layout's offsets of all eight elements first, at clamped addresses: behind `if (lb[j] == ~0) continue` every
            // element's read / write pair waited for the one before)
            u32 tf[SP_ITEMS];
#pragma unroll
            for (int j = 0; j < SP_ITEMS; ++j) tf[j] = toff[lb[j] != 0xFFFFFFFFu ? lb[j] : 0u];
#pragma unroll
            for (int j = 0; j < SP_ITEMS; ++j) {
                if (lb[j] == 0xFFFFFFFFu) continue;
                const u32 slot = tf[j] + rank[j];
                s_k[slot] = k[j];
                s_a[slot] = va[j];
                s_b[slot] = vb[j];
            }
        }
        SPSTAMP(5);  // reorder in LDS
        gbase[threadIdx.x] = my_rel0 + my_got;
        SPSTAMP(6);  // the reservation's answer
        __syncthreads();
        SPSTAMP(7);  // barrier
        const u32 staged = toff[BINS_ - 1] + cnt[BINS_ - 1];
        // (a fixed number of rounds: the compiler then knows how many stores are in flight; the LDS reads of ALL rounds at clamped
        // addresses and in flight together -- with `if (j < staged) { read; read; store }` per round every round was two dependent
        // LDS round trips behind the previous round's: 16 in a chain per tile)
        {
            u32 kk[SP_ITEMS], va_[SP_ITEMS], vb_[SP_ITEMS], gb_[SP_ITEMS], to_[SP_ITEMS];
#pragma unroll
            for (int r = 0; r < SP_ITEMS; ++r) {
                const u32 j = threadIdx.x + (u32)r * TPB, jj = j < staged ? j : 0u;
                kk[r] = s_k[jj];
                va_[r] = s_a[jj];
                vb_[r] = s_b[jj];
            }
#pragma unroll
            for (int r = 0; r < SP_ITEMS; ++r) {
                const u32 j = threadIdx.x + (u32)r * TPB;
                u32 x = FINE ? kk[r] - bin0 : kk[r] >> shift;
                x = j < staged && x < (u32)BINS_ ? x : 0u;
                gb_[r] = gbase[x];
                to_[r] = toff[x];
            }
            // (EVERY lane stores in every round -- a lane without an element into a trash slot of the workspace: with the stores
            // behind `if (j < staged)` the compiler cannot count them, and its wait for the NEXT tile's prefetched elements at the top
            // of the loop became s_waitcnt vmcnt(0): a wait for this tile's stores to be acknowledged, on every tile)
            uint2 *const trash_ab = trash + (threadIdx.x & 63u);
            KeyT *const trash_key = reinterpret_cast<KeyT *>(trash + 64) + (threadIdx.x & 63u);
#pragma unroll
            for (int r = 0; r < SP_ITEMS; ++r) {
                const u32 j = threadIdx.x + (u32)r * TPB;
                const u32 pos = gb_[r] + (j - to_[r]);
                *(j < staged ? out_ab + pos : trash_ab) = make_uint2(va_[r], vb_[r]);
                if (!FINE) *(j < staged ? out_key + pos : trash_key) = (KeyT)kk[r];
            }
        }
        SPSTAMP(8);  // write-out
        // LDS is reused by the next tile: order the LDS accesses only -- a full barrier would also drain this tile's stores
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        SPSTAMP(9);  // closing barrier
    }
#if SP_STAMPS
    if (threadIdx.x == 0) {
        for (int k = 0; k < 10; ++k) atomicAdd(&g_split_stamps[FINE ? 1 : 0][k], sp_acc[k]);
        atomicAdd(&g_split_stamps[FINE ? 1 : 0][15], 1ull);
    }
#endif
}

#if SP_STAMPS
}  // namespace gtars
extern "C" int gtars_debug_split_stamps(unsigned long long *out, int reset) {  // out[2][16]: pass A, pass B
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(gtars::g_split_stamps), 256) != hipSuccess) return 1;
    if (reset) {
        unsigned long long z[32] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(gtars::g_split_stamps), z, 256) != hipSuccess) return 1;
    }
    return 0;
}
namespace gtars {
#endif
// table [max(256, n / 65532 + 2)][n_bins] | tot [n_bins] | cursors (coarse [1024], fine [n_bins]) | first-level output: keys [n], pairs [n]
// (the table region also holds a table_ready caller's rows of packed 16-bit counts: one row per counting workgroup -- at most
// max(512, n / 65532 + 2) of them, a workgroup counts <= 65535 elements -- of multisplit_row_words(n_bins) words)
static size_t multisplit_table_words(u32 n_bins, u32 n) {
    return std::max<size_t>((size_t)std::max<u32>(256u, n / 65532u + 2u) * n_bins,
                            (size_t)std::max<u32>(512u, n / 65532u + 2u) * multisplit_row_words(n_bins));
}
size_t multisplit_ws_bytes(u32 n_bins, u32 n) {
    return (multisplit_table_words(n_bins, n) + 2 * (size_t)n_bins + 1024 + 256) * 4 + (size_t)n * 12 + 256 + 1024;  // (+ the passes' trash slots)
}

// elements per workgroup of the counting / scatter grid: a multiple of 4 (callers that count the keys themselves take four
// consecutive elements per lane)
u32 multisplit_chunk(u32 n) {
    const u32 n_wg = multisplit_workgroups(n);
    return ((n + n_wg - 1) / n_wg + 3u) & ~3u;
}
u32 *multisplit_table(void *ws) { return (u32 *)ws; }
static bool multisplit_two_level(u32 n_bins, u32 n) {
    const bool one_level = cfg_flag("GTARS_MS_ONE_LEVEL");  // A/B
    if (n_bins > MS_MAX_BINS) return true;  // beyond the one-level split's LDS counters (databases of 75M+ records)
    return n_bins > 1024 && n >= (1u << 20) && !one_level;
}
// where a table_ready caller of the two-level split leaves the bin TOTALS (zeroed by the caller, filled with atomics) instead of
// per-workgroup rows; null when the split of (n_bins, n) is one-level and wants the table
u32 *multisplit_totals(void *ws, u32 n_bins, u32 n) { return multisplit_two_level(n_bins, n) ? (u32 *)ws + multisplit_table_words(n_bins, n) : nullptr; }
// words from multisplit_totals() on that such a caller zeroes before it counts: the totals and, behind them, the passes'
// relative cursors (coarse [1024], fine [n_bins])
size_t multisplit_zeroed_words(u32 n_bins) { return (size_t)n_bins * 2 + 1024; }
u32 multisplit_coarse_shift(u32 n_bins) {
    u32 shift = 0;
    while (((n_bins - 1) >> shift) >= 256u) ++shift;
    return shift;
}
// (the coarse cursors use the first 256 of their 1024 words: the coarse totals live in the next 256)
u32 *multisplit_coarse_totals(void *ws, u32 n_bins, u32 n) {
    u32 *tot = multisplit_totals(ws, n_bins, n);
    return tot ? tot + n_bins + 256 : nullptr;
}

// The caller's own kernel (the IGD routing kernel, in the pass that computes the keys) has already COUNTED the keys: for a
// one-level split one row of n_bins u32 counters per workgroup of a multisplit_workgroups(n) x MS_TPB grid whose workgroup w
// covers elements [w * chunk, (w + 1) * chunk) at multisplit_table(ws); for a two-level split (multisplit_totals(ws) != null)
// n_count_rows rows of packed 16-bit counts there, the coarse totals at multisplit_coarse_totals(ws), and
// multisplit_zeroed_words(n_bins) zeroed words from multisplit_totals(ws) on.  Keys are 16 bits wide (n_bins <= 65535); the start
// clamp of Igd::count_overlaps (igd.rs:517) is applied to `a` on the way.
gtars_status multisplit_pairs(const unsigned short *key, const u32 *a, const u32 *b, u32 n, u32 n_bins, u32 drop_bin, uint2 *out_ab,
                              u32 *bin_off, void *ws, size_t ws_bytes, hipStream_t st, const u32 *run_if, const u32 *set_bounds,
                              const HeavyBins *heavy, u32 n_count_rows) {
    typedef unsigned short KeyT;
    constexpr bool CLAMP = true;
    const SetTags tags = set_bounds ? SetTags{set_bounds[0], set_bounds[1], set_bounds[2]} : SetTags{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    if (n_bins == 0 || n_bins > MS_MAX_BINS_2L + 1) return fail(GTARS_ERR_INTERNAL, "multisplit: too many bins");
    if (ws_bytes < multisplit_ws_bytes(n_bins, n)) return fail(GTARS_ERR_INTERNAL, "multisplit workspace too small");
    const u32 n_wg = multisplit_workgroups(n);
    const u32 chunk = multisplit_chunk(n);
    u32 *table = (u32 *)ws, *tot = table + multisplit_table_words(n_bins, n);
    u32 *cur_a = tot + n_bins, *cur_b = cur_a + 1024;
    KeyT *tmp_key = (KeyT *)(((uintptr_t)(cur_b + n_bins) + 63) & ~(uintptr_t)63);
    uint2 *tmp_ab = reinterpret_cast<uint2 *>(reinterpret_cast<u32 *>(tmp_key) + (((size_t)n + 15) & ~(size_t)15));
    uint2 *trash = tmp_ab + n;  // 64 pairs + 64 keys nobody reads (k_split_pass: lanes without an element store there)
    const size_t lds = (size_t)n_bins * 4;  // (one-level kernel only: n_bins <= MS_MAX_BINS there)
    constexpr size_t sp_lds_a = ((size_t)SP_TPB_A * SP_ITEMS * 3 + (size_t)SP_TPB_A * 3) * 4, sp_lds_b = ((size_t)SP_TPB_B * SP_ITEMS * 3 + (size_t)SP_TPB_B * 3) * 4;
    // the dynamic-LDS limits belong to the functions (per device); a failed attempt is retried by the next call
    static std::mutex mu;
    static bool done[16] = {};
    int dev = 0;
    GT_HIP(hipGetDevice(&dev));
    {
        std::lock_guard<std::mutex> lock(mu);
        if (!done[dev & 15]) {
            GT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_ms_scatter<KeyT, CLAMP>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)(MS_MAX_BINS * 4)));
            GT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_split_pass<SP_TPB_A, false, KeyT, CLAMP>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)sp_lds_a));
            GT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_split_pass<SP_TPB_B, true, KeyT, false>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)sp_lds_b));
            done[dev & 15] = true;
        }
    }
    const bool two_level = multisplit_two_level(n_bins, n);
    if (two_level) {
        // the bin scan is part of the two passes (k_split_pass: relative cursors, column sums of the counting workgroups' rows)
        const u32 shift = multisplit_coarse_shift(n_bins);
        const u32 tiles_a = (n + SP_TPB_A * SP_ITEMS - 1) / (SP_TPB_A * SP_ITEMS), tiles_b = (n + SP_TPB_B * SP_ITEMS - 1) / (SP_TPB_B * SP_ITEMS);
        const unsigned grid_a = std::min<u32>(256 * (1024 / SP_TPB_A), tiles_a), grid_b = std::min<u32>(256 * (1024 / SP_TPB_B), tiles_b);
        ProfScope p("k_split_pass", st);
        const SetTags no_tags{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
        hipLaunchKernelGGL((k_split_pass<SP_TPB_A, false, KeyT, CLAMP>), dim3(grid_a), dim3(SP_TPB_A), sp_lds_a, st, key, a, b, (const uint2 *)nullptr, n,
                           shift, drop_bin, cur_a, tmp_key, tmp_ab, run_if, tags, (const u32 *)table, n_bins, bin_off, heavy ? *heavy : HeavyBins{},
                           (const u32 *)(cur_a + 256), n_count_rows, trash);
        hipLaunchKernelGGL((k_split_pass<SP_TPB_B, true, KeyT, false>), dim3(grid_b), dim3(SP_TPB_B), sp_lds_b, st, (const KeyT *)tmp_key,
                           (const u32 *)nullptr, (const u32 *)nullptr, tmp_ab, n, shift, drop_bin, cur_b, (KeyT *)nullptr, out_ab, run_if, no_tags,
                           (const u32 *)nullptr, n_bins, bin_off, HeavyBins{}, (const u32 *)nullptr, 0u, trash);
    } else {
        {
            ProfScope p("k_ms_scan", st);
            hipLaunchKernelGGL(k_ms_colscan, dim3((n_bins + 255) / 256), dim3(256), 0, st, table, n_wg, n_bins, tot, run_if);
            constexpr u32 S_SMALL = (MS_MAX_BINS + MS_TPB - 1) / MS_TPB;
            hipLaunchKernelGGL(k_ms_binscan<S_SMALL>, dim3(1), dim3(MS_TPB), 0, st, tot, n_bins, bin_off, 0u, (u32 *)nullptr, cur_b, run_if,
                               heavy ? *heavy : HeavyBins{});
        }
        ProfScope p("k_ms_scatter", st);
        hipLaunchKernelGGL((k_ms_scatter<KeyT, CLAMP>), dim3(n_wg), dim3(MS_TPB), lds, st, key, a, b, n, n_bins, chunk, table, bin_off, drop_bin,
                           out_ab, run_if, tags);
    }
    GT_HIP(hipGetLastError());
    return GTARS_OK;
}

__global__ void k_iota(u32 *__restrict__ p, u32 n) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = i;
}

__global__ void k_gather_u32(const u32 *__restrict__ src, const u32 *__restrict__ idx, u32 n, u32 *__restrict__ dst) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[idx[i]];
}

gtars_status device_gather_u32(const u32 *src, const u32 *idx, u32 n, u32 *dst, hipStream_t st) {
    if (n == 0) return GTARS_OK;
    ProfScope p("k_gather_u32", st);
    hipLaunchKernelGGL(k_gather_u32, dim3((n + 255) / 256), dim3(256), 0, st, src, idx, n, dst);
    GT_HIP(hipGetLastError());
    return GTARS_OK;
}

size_t radix_sort_ws_bytes(u32 n) {
    const u32 tiles = (n + RS_TILE - 1) / RS_TILE;
    const size_t table = (size_t)256 * tiles;
    // table u32 | table offsets u64 (+1) | scan partials
    return table * 4 + (table + 1) * 8 + scan_ws_bytes(table) + 256;
}

// Sort (keys, vals) by bits [begin_bit, end_bit) of keys, stable.  Ping-pongs between (k0,v0) and
// (k1,v1); returns which pair holds the result through *in_first (0: k0/v0, 1: k1/v1).
gtars_status radix_sort_pairs(u32 *k0, u32 *v0, u32 *k1, u32 *v1, u32 n, int begin_bit, int end_bit, void *ws,
                              size_t ws_bytes, int *result_in, hipStream_t st) {
    *result_in = 0;
    if (n == 0) return GTARS_OK;
    const u32 tiles = (n + RS_TILE - 1) / RS_TILE;
    const size_t table = (size_t)256 * tiles;
    if (ws_bytes < radix_sort_ws_bytes(n)) return fail(GTARS_ERR_INTERNAL, "radix sort workspace too small");
    u32 *d_table = (u32 *)ws;
    u64 *d_off = (u64 *)((char *)ws + ((table * 4 + 15) & ~(size_t)15));
    void *d_scan = (char *)d_off + (((table + 1) * 8 + 15) & ~(size_t)15);
    const size_t scan_bytes = ws_bytes - (size_t)((char *)d_scan - (char *)ws);
    u32 *ki = k0, *vi = v0, *ko = k1, *vo = v1;
    int cur = 0;
    for (int shift = begin_bit; shift < end_bit; shift += 8) {
        {
            ProfScope p("k_radix_hist", st);
            hipLaunchKernelGGL(k_radix_hist, dim3(tiles), dim3(RS_TPB), 0, st, ki, n, shift, d_table, tiles);
        }
        gtars_status s = launch_scan_u32_to_u64(d_table, table, d_off, d_scan, scan_bytes, st);
        if (s) return s;
        {
            ProfScope p("k_radix_scatter", st);
            hipLaunchKernelGGL(k_radix_scatter, dim3(tiles), dim3(RS_TPB), 0, st, ki, vi, n, shift, d_off, tiles, ko, vo);
        }
        std::swap(ki, ko);
        std::swap(vi, vo);
        cur ^= 1;
    }
    GT_HIP(hipGetLastError());
    *result_in = cur;
    return GTARS_OK;
}

// Order of the reference for one index: perm such that
//   (chrom, k1, k2, input order) ascending, k2 optional (Bits: k1 = start, k2 = end; IGD: k1 = start).
// d_perm (n u32) receives the permutation.  d_chrom/d_k1/d_k2 are the unsorted device columns.
size_t device_sort_perm_ws_bytes(u32 n) { return (size_t)n * 4 * 3 + radix_sort_ws_bytes(n) + 64; }

// asynchronous form: caller provides the scratch buffer, nothing is synchronised
gtars_status device_sort_perm_ws(const u32 *d_chrom, const u32 *d_k1, const u32 *d_k2, u32 n, u32 n_chrom, u32 *d_perm,
                                 void *scratch, size_t scratch_bytes, hipStream_t st) {
    if (n == 0) return GTARS_OK;
    if (scratch_bytes < device_sort_perm_ws_bytes(n)) return fail(GTARS_ERR_INTERNAL, "sort scratch too small");
    u32 *buf = (u32 *)scratch;
    const size_t wsb = radix_sort_ws_bytes(n);
    u32 *kA = buf, *kB = buf + n, *vB = buf + 2 * (size_t)n;
    void *ws = (void *)(buf + 3 * (size_t)n);
    u32 *vA = d_perm;
    const unsigned g = (n + 255) / 256;
    hipLaunchKernelGGL(k_iota, dim3(g), dim3(256), 0, st, vA, n);
    gtars_status s = GTARS_OK;
    u32 *kc = kA, *vc = vA, *ko = kB, *vo = vB;
    auto pass = [&](const u32 *col, int bits) -> gtars_status {
        hipLaunchKernelGGL(k_gather_u32, dim3(g), dim3(256), 0, st, col, vc, n, kc);
        int res = 0;
        gtars_status r = radix_sort_pairs(kc, vc, ko, vo, n, 0, bits, ws, wsb, &res, st);
        if (r) return r;
        if (res) {
            std::swap(kc, ko);
            std::swap(vc, vo);
        }
        return GTARS_OK;
    };
    if (d_k2) s = pass(d_k2, 32);
    if (!s) s = pass(d_k1, 32);
    if (!s && n_chrom > 1) {
        int bits = 0;
        while ((1u << bits) < n_chrom) ++bits;
        bits = (bits + 7) & ~7;
        s = pass(d_chrom, bits);
    }
    if (!s && vc != d_perm) GT_HIP(hipMemcpyAsync(d_perm, vc, (size_t)n * 4, hipMemcpyDeviceToDevice, st));
    return s;
}

gtars_status device_sort_perm(const u32 *d_chrom, const u32 *d_k1, const u32 *d_k2, u32 n, u32 n_chrom, u32 *d_perm,
                              hipStream_t st) {
    if (n == 0) return GTARS_OK;
    void *buf = nullptr;
    const size_t bytes = device_sort_perm_ws_bytes(n);
    GT_HIP(hipMalloc(&buf, bytes));
    gtars_status s = device_sort_perm_ws(d_chrom, d_k1, d_k2, n, n_chrom, d_perm, buf, bytes, st);
    hipError_t e = hipStreamSynchronize(st);
    (void)hipFree(buf);
    if (s) return s;
    GT_HIP(e);
    return GTARS_OK;
}

}  // namespace gtars
