// igd_sweep.hip -- K5, batch form: Igd::count_set_overlaps / count_region_hits (gtars-igd/src/igd.rs:504-590, tile walk
// :753-847) for a large query batch as ONE streaming pass over the database.
//
// The per-query kernel in kernels.hip searches a 5e7-record array for every query (random HBM accesses).  Here the roles are
// swapped.  The database -- one record per stored interval, sorted by (chromosome, start) -- is cut into tiles of IGD_TILE
// records; a query is OWNED by the tile that holds the first record that can overlap it (first tile whose bound, last start +
// longest record + 1, is > q.start), so every record is read from HBM once per batch and the scan of a query's candidates runs
// out of LDS (the tile + a halo of IGD_HALO records; beyond: the query's own lane, from global memory).  Queries need not be
// sorted, only PARTITIONED by owner tile.  One call is five launches:
//   k_igd_begin      result vector, the split's cursors, and the order check: a batch already in (chromosome, start) order is
//                    swept as it lies (its tiles' query ranges by binary search in the routing launch);
//   k_igd_route      validity rules (igd.rs:514-517), owner tile of every query through static tables (fine buckets with about
//                    one tile boundary each), a u16 key per query, per-workgroup rows of 16-bit counts for the split;
//   k_split_pass x2  (sort.hip) two-level counting split of the (start, end) pairs by key, bin scan included;
//   the sweep        per tile: stage what the tile's queries need, count, add per-file counts into an LDS histogram, flush once.
// Three forms of the sweep, all applying exactly the reference's hit rule  min(r.end, qe) - max(r.start, qs) >= min_overlap
// (igd.rs:792-795), each stored interval once per query whatever the reference's tile replication (igd.rs:812-817):
//   k_igd_sweep_rank   pairwise counts, min_overlap == 1 (round 5): no candidate walk at all -- two ranks per query (among the
//                      tile's starts, among its SORTED ends) into two LDS histograms, prefix sums, one add per record; the tile's
//                      arrays arrive by LDS-DMA a tile ahead (see the comment at the kernel);
//   k_igd_sweep<0/2>   the candidate walk by quads out of LDS: pairwise counts for other min_overlap values and pieces views,
//                      binary counts (LOLA support) for min_overlap == 1 through pme_file -- a record is the FIRST hit of its
//                      file for a query iff it overlaps and no earlier record of the file ends after q.start, a per-record
//                      constant of the database -- also for up to four query sets in one pass: set tags in the pairs' top bits
//                      when the batch was partitioned, or (sets that are each in order, k_igd_begin<.., MS>: a LOLA call's universe
//                      and user sets) the batch as it arrived with a row of tile ranges per set (the SIO instantiations);
//   k_igd_sweep<1>     binary counts with a per-query list of credited files (other min_overlap values).
// Everything a tile's queries need that depends on the database only (search tables, u16 file ids, prefix maxima, sorted ends and
// their ranks, routing tables) is built once with the index (k_igd_tile_tables, k_igd_tile_tables_rank, api.hip finish_tiles).
//
// Bound: HBM.  Algorithmic bytes 12*Nq + 16*Ndb + 8*F; no MFMA (integer compare / scan / histogram).
#include <algorithm>
#include <mutex>
#include <type_traits>

#include "common.h"
#include "scan.h"

namespace gtars {

#ifndef GTARS_IGD_ABLATE
#define GTARS_IGD_ABLATE 0  // timing experiments (results wrong): 1 one record per query, 2 no histogram atomics, 4 no prefix-max scan, 8 no queries, 16 no searches, 32 (rank form) no prefix sums / adds
#endif
constexpr int IGD_TILE = (int)IGD_TILE_RECORDS;
constexpr int SW_TPB = 512;

constexpr int IGD_HALO = GTARS_IGD_HALO;  // (common.h)
static_assert(IGD_TILE_BLOCK == (u32)(IGD_TILE + IGD_HALO + 4) && IGD_TILE_RECORDS == 2048, "common.h states the block size");
#ifndef IGD_STAMPS
#define IGD_STAMPS 0  // diagnostic build: per-phase shader-clock totals of wave 0 of every workgroup (tools/r03_sweep_stamps.py)
#endif
#if IGD_STAMPS
__device__ unsigned long long g_sweep_stamps[8];
__device__ unsigned long long g_route_stamps[8];
#define STAMP(k)                                                  \
    do {                                                          \
        const u64 _t = __builtin_amdgcn_s_memtime();              \
        st_acc[k] += _t - st_last;                                \
        st_last = _t;                                             \
    } while (0)
#else
#define STAMP(k) \
    do {         \
    } while (0)
#endif
constexpr int LUT_NB_S = (int)IGD_LUT_S_NB, LUT_NB_P = (int)IGD_LUT_P_NB;  // buckets of the per-tile search tables
// smallest shift with (span >> shift) < NB (NB a power of two)
template <int NB>
__device__ __forceinline__ u32 lut_shift(u32 span) {
    constexpr int K = 31 - __builtin_clz((unsigned)NB);
    const int bits = 32 - __clz((int)(span | 1u));  // significant bits of span
    return bits > K ? (u32)(bits - K) : 0u;
}
// first index i in [0, n) with key[i] >= x, through the table: the answer lies in [lut[b], lut[b + 1]] for b = bucket(x) -- the
// two entries come by two 2-byte LDS reads
__device__ __forceinline__ void lut_range(const unsigned short *lut, i32 x, i32 k0, i32 k1, u32 sh, u32 n, u32 &l, u32 &h) {
    if (x <= k0) {
        l = h = 0;
    } else if (x > k1) {
        l = h = n;
    } else {
        const u32 b = (u32)(x - k0) >> sh;
#ifdef GTARS_IGD_LUT_B32
        typedef unsigned short us2 __attribute__((ext_vector_type(2)));
        typedef us2 us2_a2 __attribute__((aligned(2)));
        typedef const __attribute__((address_space(3))) us2_a2 *lds_us2;
        const us2 p = *(lds_us2)(uintptr_t)(lut + b);
        l = p.x;
        h = p.y;
#else
        // (volatile: left to itself the compiler fuses the two into one 4-byte read at a 2-byte-aligned address, and a misaligned
        // LDS read is no fast path -- round 6, measured on k_igd_route)
        typedef const volatile __attribute__((address_space(3))) unsigned short *lds_vu16;
        const lds_vu16 pp = (lds_vu16)(uintptr_t)(lut + b);
        l = pp[0];
        h = pp[1];
#endif
    }
}
// lanes that walk one query's candidate records together (pair loop of the sweep; measured 4 / 8 / 16: 0.263 / 0.278 / 0.318 ms):
// a QUAD, so that a query's state reaches its group through DPP quad_perm moves (VALU) instead of ds_bpermute (the LDS pipe,
// which is what the sweep is bound by)
constexpr int IGD_GROUP_LANES = 4;
template <int K>
__device__ __forceinline__ int quad_bcast(int v) {  // every lane of a quad receives the value of the quad's lane K
    return __builtin_amdgcn_mov_dpp(v, K * 0x55, 0xf, 0xf, true);
}
constexpr int IGD_SEEN = 32;   // per-thread list of credited files (binary counting)  // records after the tile kept in LDS too (a query's scan may run past its tile)

// ---- query preparation: validity rules of Igd::count_overlaps (igd.rs:514-517) ------------------
// (values in, values out, selects instead of conditional stores: with reference parameters into per-lane arrays the compiler
// merged the stores of the two branches through a SELECTED ADDRESS, which forced the arrays -- the routing kernel's s[] and e[] --
// into scratch memory: 48 bytes per lane written and read back through the vector-memory pipe on every step)
constexpr int PREP_TPB = 1024;
__device__ __forceinline__ void igd_prep_one(u32 c_in, u32 s_in, u32 e_in, u32 n_chrom, u32 &c, i32 &s, i32 &e) {
    const i32 s0 = (i32)s_in, e0 = (i32)e_in;  // `as i32` (igd.rs:549-550)
    const bool bad = s0 >= e0 || e0 <= 0 || c_in >= n_chrom;
    c = bad ? n_chrom : c_in;        // rejected: sorts behind every real chromosome; never served
    s = bad ? 0 : (s0 < 0 ? 0 : s0);  // clamp (igd.rs:517)
    e = bad ? 0 : e0;
}

// Prepared copies of the three columns + "is the batch out of (chromosome, start) order?" -- the FULL-SORT path only: databases
// beyond 65534 tiles (134M records: 16-bit owner keys do not reach) and GTARS_IGD_FULL_SORT (tests); everything smaller is routed
// on the raw columns by k_igd_route.
__global__ void __launch_bounds__(PREP_TPB)
k_igd_prep_queries(const u32 *__restrict__ qc, const u32 *__restrict__ qs, const u32 *__restrict__ qe, u32 nq, u32 n_chrom, u32 chunk,
                   u32 *__restrict__ kc, u32 *__restrict__ ks, u32 *__restrict__ ke, u32 *__restrict__ unsorted) {
    const u32 lo = blockIdx.x * chunk, hi = min(nq, lo + chunk);
    const int lane = threadIdx.x & 63;
    bool bad = false;
    for (u32 base = lo; base < hi; base += PREP_TPB) {
        const u32 i = base + threadIdx.x;
        const bool ok = i < hi;
        u32 c = n_chrom;
        i32 s = 0, e = 0;
        if (ok) igd_prep_one(qc[i], qs[i], qe[i], n_chrom, c, s, e);
        // already in (chromosome, start) order?  then the sort is skipped (BED inputs usually are)
        u32 pc = __shfl_up(c, 1, 64);
        u32 ps = __shfl_up((u32)s, 1, 64);
        if (lane == 0 && ok && i > 0) {
            i32 s1, e1;
            igd_prep_one(qc[i - 1], qs[i - 1], qe[i - 1], n_chrom, pc, s1, e1);
            ps = (u32)s1;
        }
        if (ok && i > 0 && (pc > c || (pc == c && ps > (u32)s))) bad = true;
        if (ok) {
            kc[i] = c;
            ks[i] = (u32)s;
            ke[i] = (u32)e;
        }
    }
    if (__any(bad) && lane == 0) *unsorted = 1u;
}

// first index in [lo, hi) with a[i] >= key; CLAMP: a holds raw i32 starts, negative ones count as 0 (igd.rs:517)
template <bool CLAMP>
__device__ __forceinline__ u32 lb_u32(const u32 *__restrict__ a, u32 lo, u32 hi, u64 key) {
    while (lo < hi) {
        const u32 mid = lo + ((hi - lo) >> 1);
        const u32 x = a[mid];
        if ((u64)(CLAMP && (i32)x < 0 ? 0u : x) < key)
            lo = mid + 1;
        else
            hi = mid;
    }
    return lo;
}

// Every query is OWNED by exactly one tile: the tile that holds its lower_bound position
// p = first record of the chromosome with start >= key, key = max(q.start - max_len, 0).
// key is monotone in q.start, so the owned queries of a tile are a contiguous range [ql, qh) of the
// sorted queries:   last_start(previous tile) < key <= last_start(this tile).
template <bool CLAMP>
__device__ __forceinline__ void igd_tile_range_one(const IgdView &v, const u32 *__restrict__ tile_first, const u32 *__restrict__ tile_cnt,
                                                   const u32 *__restrict__ tile_chrom, u32 t, const u32 *__restrict__ sorted_qs,
                                                   const u32 *__restrict__ cq_off, u32 *__restrict__ ql, u32 *__restrict__ qh,
                                                   const HeavyBins &heavy) {
    const u32 c = tile_chrom[t], p0 = tile_first[t], cnt = tile_cnt[t];
    const u64 max_len = (u64)v.chrom_maxlen[c];
    const u32 lo = cq_off[c], hi = cq_off[c + 1];
    const bool first_of_chrom = p0 == v.chrom_off[c];
    // key > prev_last  <=>  q.start >= prev_last + max_len + 1   (keys clamped to 0 belong to the first tile)
    // The two bounds are searched in LOCKSTEP: each search is a chain of ~23 dependent loads for a 10M-query batch, and the
    // launch that runs this is nothing but such chains (17 us per call of an in-order batch when they ran one after the other).
    const u64 ka = first_of_chrom ? 0ull : (u64)v.starts[p0 - 1] + max_len + 1, kb = (u64)v.starts[p0 + cnt - 1] + max_len + 1;
    u32 la = lo, ha = first_of_chrom ? lo : hi, lb = lo, hb = hi;
    while (la < ha || lb < hb) {
        const u32 ma = la + ((ha - la) >> 1), mb = lb + ((hb - lb) >> 1);
        const u32 xa = sorted_qs[la < ha ? ma : lo], xb = sorted_qs[lb < hb ? mb : lo];  // (both loads in flight together)
        if (la < ha) {
            if ((u64)(CLAMP && (i32)xa < 0 ? 0u : xa) < ka)
                la = ma + 1;
            else
                ha = ma;
        }
        if (lb < hb) {
            if ((u64)(CLAMP && (i32)xb < 0 ? 0u : xb) < kb)
                lb = mb + 1;
            else
                hb = mb;
        }
    }
    const u32 a = la, b = lb;
    ql[t] = a;
    qh[t] = b;
    heavy.note(t, b - a);  // a tile far heavier than the average is served in parts (k_igd_sweep)
}

// ---- routing: owner tile of every query + the partition's histogram, in ONE pass over the raw batch ------------------------
// Round 2 prepared the batch in one kernel (four columns written: clamped chromosome / start / end and the owner tile, 160 MB
// for config 3) and counted the owner tiles in a second one.  Here the raw columns are read once, the owner tile leaves as a
// u16 key (2 bytes per query) and the per-workgroup histogram the partition needs is counted in the same pass (16-bit
// counters packed two per LDS word: a workgroup's chunk is at most 65535 queries); the start clamp (igd.rs:517) is applied
// later, where the columns are read anyway (multisplit_pairs clamp_a, the sweep's own loads).  The owner search goes through a
// static table over (chromosome, start >> shift) built with the index (IgdTiles::route_*): 2 + ~3 LDS round trips instead of
// a 10-step binary search of the chromosome's tile bounds.
constexpr int RT_TPB = 1024;
constexpr int RT_U = 4;  // queries per thread and step (their loads and searches overlap)
// Two forms of the owner search:
//   FINE (databases whose fine tables fit the LDS next to the counters: up to ~36k tiles = 75M records): IgdTiles::route_f* --
//     buckets of 2^route_fshift positions with about one tile boundary each, the boundaries as 16-bit offsets inside their
//     bucket: the pair of table entries by one 4-byte LDS read, then one or two 2-byte probes (round 3 kept the 4-byte bounds
//     themselves in LDS behind a 4096-entry table and halved ~6 candidates: 2 + ~4 dependent LDS round trips);
//   otherwise the 4096-entry table and a binary search of the bounds in global memory (L2-resident: 4 bytes per tile), which
//     carries the fused routing to 65534 tiles = 134M records.
// LDS image of the routing kernel: {table base, last bound} per chromosome | table (u16) | [FINE: boundary offsets (u16)] |
// counters (u16), every section on a 16-byte boundary (the tables are copied by 16-byte loads)
__host__ __device__ inline u32 rt_up4(u32 words) { return (words + 3u) & ~3u; }
size_t igd_route_lds_bytes(u32 n_tiles, u32 n_chrom, u32 n_lut) {  // (bounds in global memory)
    return ((size_t)rt_up4(2 * (n_chrom + 1)) + rt_up4((n_lut + 2) / 2) + (n_tiles + 2) / 2) * 4;
}
size_t igd_route_fine_lds_bytes(u32 n_tiles, u32 n_chrom, u64 n_fine) {
    if (n_fine > 0x7FFFFFF0ull) return ~(size_t)0;
    return ((size_t)rt_up4(2 * (n_chrom + 1)) + rt_up4(((u32)n_fine + 2) / 2) + rt_up4((n_tiles + 2) / 2) + (n_tiles + 2) / 2) * 4;
}
template <bool VEC, bool FINE>
__global__ void __launch_bounds__(RT_TPB)
k_igd_route(const u32 *__restrict__ qc, const u32 *__restrict__ qs, const u32 *__restrict__ qe, u32 nq, u32 n_chrom,
            const u32 *__restrict__ bnd, const u32 *__restrict__ chrom_tile_off, const u32 *__restrict__ route_base,
            const u32 *__restrict__ route_len, const u32 *__restrict__ route_lut, u32 n_lut, u32 route_shift, u32 n_tiles, u32 chunk,
            unsigned short *__restrict__ key, u32 *__restrict__ table, u32 *__restrict__ tot, const u32 *__restrict__ run_if,
            u32 *__restrict__ ctot, u32 cshift, const u32 *__restrict__ route_kq, IgdView v, const u32 *__restrict__ tile_first,
            const u32 *__restrict__ tile_cnt, const u32 *__restrict__ tile_chrom, const u32 *__restrict__ cq_off, u32 *__restrict__ ql,
            u32 *__restrict__ qh, HeavyBins heavy, u32 n_sets_io) {
    if (run_if && *run_if == 0) {
        // the batch is in owner order (the order check in front of this kernel): nothing to route, the sweep takes it as it lies --
        // and this launch computes the tiles' query ranges instead (what k_igd_tile_ranges did in a launch of its own)
        // (tile t goes to workgroup t mod grid: the binary searches are chains of dependent loads, so they want to be spread
        // over every CU's memory path rather than packed into the first few workgroups)
        // (several sets in order: one row of ranges per set -- stride n_tiles -- from that set's chromosome offsets; no heavy-tile
        // parts there: a part is a range of ONE list of queries)
        if (ql) {
            if (n_sets_io <= 1) {
                for (u32 t = threadIdx.x * gridDim.x + blockIdx.x; t < n_tiles; t += gridDim.x * RT_TPB)
                    igd_tile_range_one<true>(v, tile_first, tile_cnt, tile_chrom, t, qs, cq_off, ql, qh, heavy);
            } else {
                // (a thread per (tile, set): the searches are chains of dependent loads -- the sets side by side, not one after the other)
                for (u32 x = threadIdx.x * gridDim.x + blockIdx.x; x < n_tiles * n_sets_io; x += gridDim.x * RT_TPB) {
                    const u32 k = x / n_tiles, t = x - k * n_tiles;
                    igd_tile_range_one<true>(v, tile_first, tile_cnt, tile_chrom, t, qs, cq_off + k * (n_chrom + 2u), ql + (size_t)k * n_tiles,
                                             qh + (size_t)k * n_tiles, HeavyBins{});
                }
            }
        }
        return;
    }
    extern __shared__ u32 rt_lds[];
    // FINE: route_base / route_lut / n_lut / route_shift are the FINE tables' (IgdTiles::route_fbase / route_flut / route_fn /
    // route_fshift); {base, len} pairs per chromosome | table | [boundary offsets] | counters
    uint2 *s_bl = reinterpret_cast<uint2 *>(rt_lds);
    u32 *s_lutw = rt_lds + rt_up4(2 * (n_chrom + 1));
    u32 *s_kqw = s_lutw + rt_up4((n_lut + 2) / 2);
    u32 *bins = s_kqw + (FINE ? rt_up4((n_tiles + 2) / 2) : 0u);  // (n_tiles + 2) / 2 words: bin b in half (b & 1) of word b >> 1
    const unsigned short *s_lut = reinterpret_cast<const unsigned short *>(s_lutw);
    const unsigned short *s_kq = reinterpret_cast<const unsigned short *>(s_kqw);
#if IGD_STAMPS
    u64 st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_last = __builtin_amdgcn_s_memtime();
#endif
    // (the first full step's columns are requested HERE, in front of the 97-KB table copy: their round trip to HBM overlaps it)
    typedef u32 v4u __attribute__((ext_vector_type(4)));
    const u32 lo_q = blockIdx.x * chunk, hi_q = min(nq, lo_q + chunk);
    v4u pc0 = v4u{0, 0, 0, 0}, ps0 = pc0, pe0 = pc0;
    if (VEC && lo_q + RT_TPB * RT_U <= hi_q) {
        const u32 f_first = lo_q + threadIdx.x * RT_U;
        pc0 = *reinterpret_cast<const v4u *>(qc + f_first), ps0 = *reinterpret_cast<const v4u *>(qs + f_first),
        pe0 = *reinterpret_cast<const v4u *>(qe + f_first);
    }
    for (u32 c = threadIdx.x; c <= n_chrom; c += RT_TPB) s_bl[c] = make_uint2(route_base[c], c < n_chrom ? route_len[c] : 0u);
    {
        // the tables: 16-byte loads, four in flight per thread (a dword-per-step copy loop is a chain of L2 round trips)
        auto copy = [&](const u32 *__restrict__ src, u32 *dst, u32 n_words) {  // (both 16-byte aligned; n_words any)
            const u32 n4 = n_words >> 2;
            const uint4 *s4 = reinterpret_cast<const uint4 *>(src);
            uint4 *d4 = reinterpret_cast<uint4 *>(dst);
            for (u32 i0 = threadIdx.x; i0 < n4; i0 += RT_TPB * 4) {
                uint4 x[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) x[k] = i0 + (u32)k * RT_TPB < n4 ? s4[i0 + (u32)k * RT_TPB] : make_uint4(0, 0, 0, 0);
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (i0 + (u32)k * RT_TPB < n4) d4[i0 + (u32)k * RT_TPB] = x[k];
            }
            for (u32 w = (n4 << 2) + threadIdx.x; w < n_words; w += RT_TPB) dst[w] = src[w];
        };
        copy(route_lut, s_lutw, (n_lut + 1) / 2);
        if (FINE) copy(route_kq, s_kqw, (n_tiles + 1) / 2);
    }
    for (u32 w = threadIdx.x; w < (n_tiles + 2) / 2; w += RT_TPB) bins[w] = 0;
    __syncthreads();
    STAMP(0);
    // A lane takes RT_U = 4 CONSECUTIVE queries per step.  VEC (16-byte aligned columns, chunk a multiple of 4): three 16-byte
    // loads and one 8-byte key store per step instead of twelve loads and four stores (the loop was instruction-bound: 17.8k
    // cycles per step by the in-kernel stamps, ~500 VALU instructions per wave and step).  The raw columns of the NEXT step
    // are loaded while the current one is searched (one workgroup per CU: nothing else covers the HBM latency).  Whether the
    // batch is in order is decided in front of this kernel (k_igd_begin).
    // (three plain vectors, not arrays filled through a lambda: those stayed in scratch memory -- 48 bytes per lane written and
    // read back through the vector-memory pipe on every step)
    static_assert(RT_U == 4, "a lane's four queries travel as one vector per column");
    v4u nc, ns, ne;
#define RT_FETCH(base_)                                                                                     \
    do {                                                                                                    \
        const u32 f0 = (base_) + threadIdx.x * RT_U;                                                        \
        if (VEC && f0 + RT_U <= hi_q) {                                                                     \
            nc = *reinterpret_cast<const v4u *>(qc + f0);                                                   \
            ns = *reinterpret_cast<const v4u *>(qs + f0);                                                   \
            ne = *reinterpret_cast<const v4u *>(qe + f0);                                                   \
        } else {                                                                                            \
            nc = v4u{f0 + 0 < hi_q ? qc[f0 + 0] : GTARS_UNKNOWN_CHROM, f0 + 1 < hi_q ? qc[f0 + 1] : GTARS_UNKNOWN_CHROM, \
                     f0 + 2 < hi_q ? qc[f0 + 2] : GTARS_UNKNOWN_CHROM, f0 + 3 < hi_q ? qc[f0 + 3] : GTARS_UNKNOWN_CHROM}; \
            ns = v4u{f0 + 0 < hi_q ? qs[f0 + 0] : 0u, f0 + 1 < hi_q ? qs[f0 + 1] : 0u, f0 + 2 < hi_q ? qs[f0 + 2] : 0u, \
                     f0 + 3 < hi_q ? qs[f0 + 3] : 0u};                                                      \
            ne = v4u{f0 + 0 < hi_q ? qe[f0 + 0] : 0u, f0 + 1 < hi_q ? qe[f0 + 1] : 0u, f0 + 2 < hi_q ? qe[f0 + 2] : 0u, \
                     f0 + 3 < hi_q ? qe[f0 + 3] : 0u};                                                      \
        }                                                                                                   \
    } while (0)
    // the owner tile of a lane's four prepared queries (n_tiles: none -- invalid, unknown chromosome, past every bound)
    auto owners = [&](const u32 (&c)[RT_U], const i32 (&s)[RT_U], u32 (&tt)[RT_U]) {
        u32 l[RT_U], h[RT_U];
        bool owned[RT_U];
        // The owner: first tile of the chromosome whose bound is > start, bracketed by the static table.  Written so that a lane's
        // four queries have their LDS reads in flight TOGETHER and no read sits behind a branch -- reads at clamped addresses,
        // selects instead of conditions (round 6).  The form `if (owned) { read; ... }` per query came out as one guarded read
        // after the other, each with its own s_waitcnt: ~18 dependent LDS round trips per step, which is what the kernel's
        // "70 % of the wave cycles waiting" were (found on k_tok_sweep with in-kernel stamps, then seen in this kernel's listing).
        {
            uint2 bl[RT_U];
#pragma unroll
            for (int u = 0; u < RT_U; ++u) bl[u] = s_bl[min(c[u], n_chrom)];  // {table base, last bound} of the chromosome
            u32 j[RT_U];
#pragma unroll
            for (int u = 0; u < RT_U; ++u) {
                owned[u] = (c[u] < n_chrom) & ((u32)s[u] < bl[u].y);  // otherwise: invalid, unknown chromosome or past every bound
                j[u] = owned[u] ? bl[u].x + ((u32)s[u] >> route_shift) : 0u;
            }
            // the bucket's pair of entries by TWO 2-byte reads (inline assembly, or the compiler fuses them): as ONE 4-byte read at a
            // 2-byte-aligned address -- every other bucket -- they cost 7 of the kernel's 47.6 us; misaligned LDS reads are no fast path
            u32 pl[RT_U], ph[RT_U], pa[RT_U];
#pragma unroll
            for (int u = 0; u < RT_U; ++u) pa[u] = (u32)(uintptr_t)(s_lut + j[u]);
            asm volatile("ds_read_u16 %0, %8\n\tds_read_u16 %1, %8 offset:2\n\tds_read_u16 %2, %9\n\tds_read_u16 %3, %9 offset:2\n\t"
                         "ds_read_u16 %4, %10\n\tds_read_u16 %5, %10 offset:2\n\tds_read_u16 %6, %11\n\tds_read_u16 %7, %11 offset:2\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&v"(pl[0]), "=&v"(ph[0]), "=&v"(pl[1]), "=&v"(ph[1]), "=&v"(pl[2]), "=&v"(ph[2]), "=&v"(pl[3]), "=&v"(ph[3])
                         : "v"(pa[0]), "v"(pa[1]), "v"(pa[2]), "v"(pa[3])
                         : "memory");
#pragma unroll
            for (int u = 0; u < RT_U; ++u) {
                l[u] = owned[u] ? pl[u] : 0u;
                h[u] = owned[u] ? ph[u] : 0u;
            }
        }
        if constexpr (FINE) {
            // tiles [l, h) have their boundary inside the query's bucket: the first one whose 16-bit offset is > the query's --
            // or equal and, with buckets wider than 2^16, really >= (the exact bound, global memory: one query in 2^16)
            const u32 qsh = route_shift - 16u, msk = (1u << route_shift) - 1u;
            bool more = true;
            while (__ballot(more)) {
                u32 kq[RT_U];
#pragma unroll
                for (int u = 0; u < RT_U; ++u) kq[u] = s_kq[l[u] < n_tiles ? l[u] : 0u];  // (an address inside the table whatever the state)
                more = false;
#pragma unroll
                for (int u = 0; u < RT_U; ++u) {
                    const bool act = l[u] < h[u];
                    const u32 sq = ((u32)s[u] & msk) >> qsh;
                    bool below = kq[u] < sq;
                    if (act && kq[u] == sq && qsh) below = bnd[l[u]] <= (u32)s[u];  // (rare: the exact bound)
                    const u32 nl = l[u] + 1u;
                    h[u] = (act & !below) ? l[u] : h[u];
                    l[u] = (act & below) ? nl : l[u];
                    more = more | (act & below & (l[u] < h[u]));
                }
            }
        } else {
            bool more = true;
            while (more) {
                more = false;
#pragma unroll
                for (int u = 0; u < RT_U; ++u) {
                    if (l[u] < h[u]) {
                        const u32 mid = l[u] + ((h[u] - l[u]) >> 1);
                        if (bnd[mid] <= (u32)s[u])
                            l[u] = mid + 1;
                        else
                            h[u] = mid;
                        more = more || l[u] < h[u];
                    }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < RT_U; ++u) tt[u] = owned[u] ? l[u] : n_tiles;
    };
    const u32 STEP = RT_TPB * RT_U;
    u32 base = lo_q;
    if constexpr (VEC) {
        // FULL steps (every lane's four queries are in range): no branch around any memory operation, so that the compiler's wait
        // for the prefetched columns is a COUNTED one.  With the range checks inside the loop (the general form below) every wait
        // came out as s_waitcnt vmcnt(0), which also waits for the step's own key store, issued a few instructions earlier: a
        // store's round trip on the critical path of every step (round 6, from the listing).
        if (base + STEP <= hi_q) {
            v4u pc = pc0, ps = ps0, pe = pe0;  // (requested in front of the table copy)
            // (a step's keys are stored at the START of the next step, in front of that step's loads: a store issued BEHIND the
            // prefetch is the youngest operation when the loop comes round, and the wait for the prefetched columns -- which the
            // compiler cannot count past the loop's entry, where there is no store yet -- would be a wait for that store as well)
            uint2 k_prev = make_uint2(0u, 0u);
            u32 i_prev = 0;
            bool have_prev = false;
            for (; base + STEP <= hi_q; base += STEP) {
                const u32 i0 = base + threadIdx.x * RT_U;
                const u32 rc[RT_U] = {pc.x, pc.y, pc.z, pc.w}, rs[RT_U] = {ps.x, ps.y, ps.z, ps.w}, re[RT_U] = {pe.x, pe.y, pe.z, pe.w};
                u32 c[RT_U], tt[RT_U];
                i32 s[RT_U], e[RT_U];
#pragma unroll
                for (int u = 0; u < RT_U; ++u) igd_prep_one(rc[u], rs[u], re[u], n_chrom, c[u], s[u], e[u]);
                // (the empty statement ties the store to the columns just waited for, so that it is not scheduled in front of that wait)
                asm volatile("" : "+v"(k_prev.x), "+v"(k_prev.y) : "v"(rc[0]), "v"(rs[0]), "v"(re[0]));
                STAMP(4);  // (diagnostic build) the wait for the columns
                if (have_prev) *reinterpret_cast<uint2 *>(key + i_prev) = k_prev;
                // the next full step's columns (behind the last one: this step's again -- an unconditional load)
                const u32 fn = (base + 2u * STEP <= hi_q ? base + STEP : base) + threadIdx.x * RT_U;
                pc = *reinterpret_cast<const v4u *>(qc + fn);
                ps = *reinterpret_cast<const v4u *>(qs + fn);
                pe = *reinterpret_cast<const v4u *>(qe + fn);
                owners(c, s, tt);
                STAMP(5);  // the owner search
#pragma unroll
                for (int u = 0; u < RT_U; ++u) atomicAdd(&bins[tt[u] >> 1], 1u << ((tt[u] & 1u) * 16u));
                STAMP(6);  // the counters
                k_prev = make_uint2(tt[0] | (tt[1] << 16), tt[2] | (tt[3] << 16));
                i_prev = i0;
                have_prev = true;
            }
            if (have_prev) *reinterpret_cast<uint2 *>(key + i_prev) = k_prev;
        }
    }
    // the rest of the chunk (a partial step; every step of a batch whose columns are not 16-byte aligned)
    if (base < hi_q) RT_FETCH(base);
    for (; base < hi_q; base += STEP) {
        const u32 i0 = base + threadIdx.x * RT_U;
        u32 c[RT_U], tt[RT_U];
        i32 s[RT_U], e[RT_U];
        bool ok[RT_U];
        const u32 rc[RT_U] = {nc.x, nc.y, nc.z, nc.w}, rs[RT_U] = {ns.x, ns.y, ns.z, ns.w}, re[RT_U] = {ne.x, ne.y, ne.z, ne.w};
#pragma unroll
        for (int u = 0; u < RT_U; ++u) {
            ok[u] = i0 + u < hi_q;
            c[u] = n_chrom;
            s[u] = e[u] = 0;
            if (ok[u]) igd_prep_one(rc[u], rs[u], re[u], n_chrom, c[u], s[u], e[u]);
        }
        if (base + STEP < hi_q) RT_FETCH(base + STEP);
        owners(c, s, tt);
#pragma unroll
        for (int u = 0; u < RT_U; ++u)
            if (ok[u]) atomicAdd(&bins[tt[u] >> 1], 1u << ((tt[u] & 1u) * 16u));
        if (VEC && i0 + RT_U <= hi_q) {
            *reinterpret_cast<uint2 *>(key + i0) = make_uint2(tt[0] | (tt[1] << 16), tt[2] | (tt[3] << 16));
        } else {
#pragma unroll
            for (int u = 0; u < RT_U; ++u)
                if (ok[u]) key[i0 + u] = (unsigned short)tt[u];
        }
    }
#undef RT_FETCH
    STAMP(1);
    __syncthreads();
    STAMP(2);
    if (tot) {
        // Two-level split: the workgroup's counters leave as they are -- a ROW of packed 16-bit counts, plain 16-byte stores --
        // and the split's first pass sums the rows of the bins it needs (multisplit_pairs, FOLD).  Round 3 added every non-zero
        // counter to a global total: 20k atomics per workgroup, 5M per call, a third of this kernel by the in-kernel stamps.
        const u32 nb = n_tiles + 1, rw = multisplit_row_words(nb), have = (n_tiles + 2) / 2;
        u32 *row = table + (size_t)blockIdx.x * rw;  // (`tot` only says that the split is two-level)
        for (u32 w4 = threadIdx.x * 4u; w4 < rw; w4 += RT_TPB * 4u) {
            uint4 x;
            x.x = w4 + 0 < have ? bins[w4 + 0] : 0u;
            x.y = w4 + 1 < have ? bins[w4 + 1] : 0u;
            x.z = w4 + 2 < have ? bins[w4 + 2] : 0u;
            x.w = w4 + 3 < have ? bins[w4 + 3] : 0u;
            *reinterpret_cast<uint4 *>(row + w4) = x;
        }
        // ... and the totals of the split's COARSE bins (2^cshift fine bins each, <= 256 of them), which every workgroup of that
        // pass needs: thread t sums quarter (t & 3) of coarse bin t >> 2 from the LDS counters, one atomic per coarse bin
        {
            const u32 n_coarse = ((nb - 1u) >> cshift) + 1u, per = 1u << (cshift - 2u);  // (cshift >= 3: per is even)
            const u32 j = threadIdx.x >> 2;
            u32 sum = 0;
            if (j < n_coarse) {
                const u32 b0 = (j << cshift) + (threadIdx.x & 3u) * per, b1 = min(nb, b0 + per);
                for (u32 w = b0 >> 1; 2u * w < b1; ++w) {
                    const u32 x = bins[w];
                    sum += (x & 0xFFFFu) + (2u * w + 1u < b1 ? x >> 16 : 0u);
                }
            }
            sum += (u32)__shfl_xor((int)sum, 1, 64);
            sum += (u32)__shfl_xor((int)sum, 2, 64);
            if ((threadIdx.x & 3u) == 0 && j < n_coarse && sum) atomicAdd(&ctot[j], sum);
        }
    } else {
        u32 *row = table + (size_t)blockIdx.x * (n_tiles + 1);
        for (u32 b = threadIdx.x; b <= n_tiles; b += RT_TPB) row[b] = (bins[b >> 1] >> ((b & 1u) * 16u)) & 0xFFFFu;
    }
    STAMP(3);
#if IGD_STAMPS
    if (threadIdx.x == 0) {
        for (int k = 0; k < 7; ++k) atomicAdd(&g_route_stamps[k], st_acc[k]);
        atomicAdd(&g_route_stamps[7], 1ull);
    }
#endif
}

// start of a batch call: the result vector, the "arrived out of owner order" flag and (fused routing) the bin totals -- one
// launch instead of four memsets
// Workgroup 0 also PROBES the batch's order (probe_n > 0: its first <= 1024 queries, every thread's loads issued together -- a
// 16-element loop per thread was a chain of 16 memory round trips, 8 us on every call): a shuffled batch is recognised here, and
// the full order check behind this kernel returns at once.
constexpr u32 ORD_PROBE = 1024;
__device__ __forceinline__ bool igd_out_of_order(u32 pc, i32 ps, u32 c, i32 s) { return pc > c || (pc == c && (u32)ps > (u32)s); }
// The flag lives in one of TWO words at the head of the workspace, by the call's parity: a call raises the word of its parity
// (zero when the call starts) and zeroes the other one for the next call -- nobody ever clears a word that somebody else may be
// raising in the same launch (Workspace::reserve zeroes both when it allocates).
__global__ void __launch_bounds__(256)
k_igd_call_init(unsigned long long *__restrict__ hits, u32 n_files, u32 *__restrict__ flag, u32 flag_value, u32 *__restrict__ tot, u32 n_tot,
                const u32 *__restrict__ qc, const u32 *__restrict__ qs, const u32 *__restrict__ qe, u32 probe_n, u32 n_chrom,
                u32 *__restrict__ next_flag, u32 *__restrict__ heavy_count) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_files) hits[i] = 0ull;
    if (i < n_tot) tot[i] = 0u;
    if (blockIdx.x == 0) {
        bool bad = false;
        if (probe_n > 1) {
            constexpr u32 PER = ORD_PROBE / 256;
            const u32 a = threadIdx.x * PER;  // pairs (k - 1, k) for k in (a, a + PER], k < probe_n
            u32 rc[PER + 1], rs[PER + 1], re[PER + 1];
#pragma unroll
            for (u32 k = 0; k <= PER; ++k) {
                const bool in = a + k < probe_n;
                rc[k] = in ? qc[a + k] : 0u;
                rs[k] = in ? qs[a + k] : 0u;
                re[k] = in ? qe[a + k] : 0u;
            }
            u32 pc = 0, c;
            i32 ps = 0, pe, s2, e2;
#pragma unroll
            for (u32 k = 0; k <= PER; ++k) {
                igd_prep_one(rc[k], rs[k], re[k], n_chrom, c, s2, e2);
                if (k > 0 && a + k < probe_n) bad = bad || igd_out_of_order(pc, ps, c, s2);
                pc = c;
                ps = s2;
                (void)pe;
            }
        }
        const int any_bad = __syncthreads_or(bad ? 1 : 0);
        if (threadIdx.x == 0) {
            if (flag_value | (any_bad ? 1u : 0u)) flag[0] = 1u;
            next_flag[0] = 0u;
            heavy_count[0] = 0u;  // number of listed heavy-tile parts (HeavyBins::count)
        }
    }
}

// Row ranges of a batch of several query sets (gtars_igd_count_sets): set k = rows [lo(k), hi(k)).  n = 1: the whole batch.
constexpr u32 IGD_MAX_SETS = 4;
struct SetRows {
    u32 n = 1, b1 = 0xFFFFFFFFu, b2 = 0xFFFFFFFFu, b3 = 0xFFFFFFFFu, nq = 0;
    __host__ __device__ u32 lo(u32 k) const { return k == 0 ? 0u : min(k == 1 ? b1 : k == 2 ? b2 : b3, nq); }
    __host__ __device__ u32 hi(u32 k) const { return k + 1 >= n ? nq : lo(k + 1); }
    __host__ __device__ u32 of(u32 row) const { return (row >= b1 ? 1u : 0u) + (row >= b2 ? 1u : 0u) + (row >= b3 ? 1u : 0u); }
};
constexpr int ORD_TPB = 256;
// Start of a call on the partition path, ONE launch (round 3: k_igd_call_init + k_igd_order_check; a launch costs ~5 us): every thread
// zeroes its words of the result vector and of the split's totals / cursors, workgroup 0 the next call's flag and the heavy-tile
// count; then the order check -- a workgroup leaves as soon as it (or anybody) has seen disorder: no probe launch in front.
template <bool VEC, bool MS = false>  // MS: several query sets, each checked by itself (SetRows)
__global__ void __launch_bounds__(ORD_TPB)
k_igd_begin(unsigned long long *__restrict__ hits, u32 n_files, u32 *__restrict__ tot, u32 n_tot, u32 *__restrict__ flag,
            u32 *__restrict__ next_flag, u32 *__restrict__ heavy_count, u32 flag0, const u32 *__restrict__ qc, const u32 *__restrict__ qs,
            const u32 *__restrict__ qe, u32 nq, u32 n_chrom, u32 *__restrict__ cq_off, SetRows sets) {
    for (u32 i = blockIdx.x * ORD_TPB + threadIdx.x; i < max(n_files, n_tot); i += gridDim.x * ORD_TPB) {
        if (i < n_files) hits[i] = 0ull;
        if (i < n_tot) tot[i] = 0u;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        next_flag[0] = 0u;
        heavy_count[0] = 0u;
        if (flag0) flag[0] = 1u;
    }
    if (flag0) return;  // the partition is certain (forced, or a form of several sets the in-order sweep does not serve): nothing to check
    // Several query sets (round 5): "in order" means every set is in order BY ITSELF -- what a LOLA call's universe and user sets
    // are (sorted region sets) --, and cq_off holds one row of chromosome offsets per set (stride n_chrom + 2).  The sweep then
    // serves the sets one after the other from a tile's staged records, and nothing is partitioned.
    const u32 cq_stride = n_chrom + 2u;
    if (MS && blockIdx.x == 0)
        for (u32 k = 0; k < sets.n; ++k)
            if (sets.lo(k) == sets.hi(k))  // an empty set: no row writes its offsets
                for (u32 x = threadIdx.x; x <= n_chrom; x += ORD_TPB) cq_off[k * cq_stride + x] = sets.lo(k);
    const int lane = threadIdx.x & 63;
    bool bad = false;
    for (u64 base = (u64)blockIdx.x * (ORD_TPB * 4); base < nq; base += (u64)gridDim.x * (ORD_TPB * 4)) {
        if (*reinterpret_cast<volatile const u32 *>(flag)) return;  // (somebody has seen disorder: the batch will be partitioned)
        const u64 i0 = base + (u64)threadIdx.x * 4;
        u32 rc[4], rs[4], re[4];
        if (VEC && i0 + 4 <= nq) {
            const uint4 a = *reinterpret_cast<const uint4 *>(qc + i0), b = *reinterpret_cast<const uint4 *>(qs + i0),
                        d = *reinterpret_cast<const uint4 *>(qe + i0);
            rc[0] = a.x, rc[1] = a.y, rc[2] = a.z, rc[3] = a.w;
            rs[0] = b.x, rs[1] = b.y, rs[2] = b.z, rs[3] = b.w;
            re[0] = d.x, re[1] = d.y, re[2] = d.z, re[3] = d.w;
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool in = i0 + u < nq;
                rc[u] = in ? qc[i0 + u] : GTARS_UNKNOWN_CHROM;  // past the end: "rejected", sorts last
                rs[u] = in ? qs[i0 + u] : 0u;
                re[u] = in ? qe[i0 + u] : 0u;
            }
        }
        // the element in front of this lane's four: the previous lane's last one; lane 0 reads it
        const bool edge = lane == 0 && i0 > 0 && i0 < nq;
        const u32 ec = edge ? qc[i0 - 1] : 0u, es = edge ? qs[i0 - 1] : 0u, ee = edge ? qe[i0 - 1] : 0u;
        u32 c[4];
        i32 s[4], e[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) igd_prep_one(rc[u], rs[u], re[u], n_chrom, c[u], s[u], e[u]);
        u32 pc = __shfl_up(c[3], 1, 64);
        i32 ps = __shfl_up(s[3], 1, 64);
        if (lane == 0) {
            i32 pe;
            igd_prep_one(ec, es, ee, n_chrom, pc, ps, pe);
        }
        const u32 pc0 = pc;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const u32 row = (u32)(i0 + u);
            const bool opens = MS ? row == sets.lo(sets.of(row)) : row == 0;  // (the first row of its set follows nothing)
            if (i0 + u < nq && !opens && igd_out_of_order(pc, ps, c[u], s[u])) bad = true;
            pc = c[u];
            ps = s[u];
        }
        if (__any(bad)) {  // (before anything is written: a shuffled batch must leave here at once)
            if (lane == 0) *flag = 1u;
            return;
        }
        // chromosome boundaries (c[u] <= n_chrom after the validity rules): cq_off[c] = first row whose chromosome is >= c -- the
        // thread that sees a chromosome change at row i writes the entries of every chromosome in between
        pc = pc0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const u64 i = i0 + u;
            if (i < nq) {
                const u32 set = MS ? sets.of((u32)i) : 0u, row0 = MS ? sets.lo(set) : 0u, row1 = MS ? sets.hi(set) : nq;
                u32 *__restrict__ co = cq_off + set * cq_stride;
                const bool opens = (u32)i == row0;
                const u32 from = opens ? 0u : pc + 1u;
                if (opens || c[u] > pc)
                    for (u32 k = from; k <= c[u]; ++k) co[k] = (u32)i;
                if ((u32)i == row1 - 1u)
                    for (u32 k = c[u] + 1u; k <= n_chrom; ++k) co[k] = row1;
            }
            pc = c[u];
        }
    }
}

__global__ void k_igd_tile_bounds(IgdView v, const u32 *__restrict__ tile_first, const u32 *__restrict__ tile_cnt,
                                  const u32 *__restrict__ tile_chrom, u32 n_tiles, u32 *__restrict__ bnd) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_tiles) return;
    const u64 b = (u64)(u32)v.starts[tile_first[t] + tile_cnt[t] - 1] + (u64)(u32)v.chrom_maxlen[tile_chrom[t]] + 1ull;
    bnd[t] = b > 0xFFFFFFFFull ? 0xFFFFFFFFu : (u32)b;
}

gtars_status launch_igd_tile_bounds(const IgdView &v, const u32 *tile_first, const u32 *tile_cnt, const u32 *tile_chrom, u32 n_tiles,
                                    u32 *bnd, hipStream_t st) {
    if (!n_tiles) return GTARS_OK;
    hipLaunchKernelGGL(k_igd_tile_bounds, dim3((n_tiles + 255) / 256), dim3(256), 0, st, v, tile_first, tile_cnt, tile_chrom, n_tiles, bnd);
    GT_HIP(hipGetLastError());
    return GTARS_OK;
}

// per-chromosome segments of the sorted queries
// `perm` (may be NULL: the batch is already ordered) maps a sorted position to its row of `chrom_key`, so the
// sorted chromosome column never has to be materialised for these n_chrom + 1 binary searches.
// RAW: chrom_key / raw_s / raw_e are the caller's columns as they came; the validity rules are applied on the way (an
// in-order batch has its invalid queries at the end: they map to chromosome n_chrom).
template <bool RAW>
__global__ void k_igd_chrom_segments(const u32 *__restrict__ chrom_key, const u32 *__restrict__ raw_s, const u32 *__restrict__ raw_e,
                                     const u32 *__restrict__ perm, u32 nq, u32 n_chrom, u32 *__restrict__ cq_off,
                                     const u32 *__restrict__ skip_if) {
    if (skip_if && *skip_if) return;  // the batch was partitioned instead (decided on the device)
    const u32 c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c > n_chrom) return;
    u32 lo = 0, hi = nq;
    while (lo < hi) {
        const u32 mid = lo + ((hi - lo) >> 1);
        const u32 row = perm ? perm[mid] : mid;
        u32 cc = chrom_key[row];
        if (RAW) {
            i32 s1, e1;
            igd_prep_one(cc, raw_s[row], raw_e[row], n_chrom, cc, s1, e1);
        }
        if (cc < c)
            lo = mid + 1;
        else
            hi = mid;
    }
    cq_off[c] = lo;
}

// dst_a[i] = a[idx[i]], dst_b[i] = b[idx[i]]: the two columns the sweep reads, one pass over the permutation
__global__ void k_gather2_u32(const u32 *__restrict__ a, const u32 *__restrict__ b, const u32 *__restrict__ idx, u32 n,
                              u32 *__restrict__ dst_a, u32 *__restrict__ dst_b) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const u32 k = idx[i];
        dst_a[i] = a[k];
        dst_b[i] = b[k];
    }
}

template <bool CLAMP>
__global__ void k_igd_tile_ranges(IgdView v, const u32 *__restrict__ tile_first, const u32 *__restrict__ tile_cnt,
                                  const u32 *__restrict__ tile_chrom, u32 n_tiles, const u32 *__restrict__ sorted_qs,
                                  const u32 *__restrict__ cq_off, u32 *__restrict__ ql, u32 *__restrict__ qh,
                                  const u32 *__restrict__ skip_if, HeavyBins heavy) {
    if (skip_if && *skip_if) return;
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_tiles) return;
    igd_tile_range_one<CLAMP>(v, tile_first, tile_cnt, tile_chrom, t, sorted_qs, cq_off, ql, qh, heavy);
}

// ---- the sweep ---------------------------------------------------------------------------------
// MODE 0: pairwise counts (count_set_overlaps); 1: binary counts (count_region_hits) with a per-query list of credited
// files; 2: binary counts for min_overlap == 1 through pme_file -- a record is the FIRST hit of its file for a query iff
// no earlier record of that file (and chromosome) ends after the query's start: earlier records start no later, so
// "ends after q.start" is all that is left of the overlap test, and the largest such end is a per-record constant of
// the database (IgdTiles::pme_file).  Binary counting then costs what pairwise counting costs (2.3 -> 0.5 ms for
// config 3) instead of a 32-entry membership test per hit.
// ---- static per-tile tables (built once per database, launch_igd_tile_tables) -------------------------------------------
// Everything a tile's queries need that depends on the DATABASE only is computed at index build and streamed with the records:
//   pm[r]      prefix maximum of the ends over the chromosome's records up to r: ascending, so "first record that can overlap
//              a query" is the first one with pm > q_start;
//   files16[r] the file ids as u16 (the LDS histogram limits a swept database to 16384 files);
//   tile_tab   per tile TAB_WORDS words: a descriptor and two direct-mapped search tables over the staged range (the tile + its
//              halo): lut[b] = first staged record whose key falls in bucket >= b, bucket(x) = (x - first key) >> shift, for the
//              starts and for pm.  A query's two binary searches then start from a handful of records instead of 2304
//              (2 + ~3 dependent LDS round trips instead of 24).
// In round 2 the sweep scanned the prefix maximum in LDS for every tile of every call (2.4k cycles per tile by the in-kernel
// stamps) and searched without tables (the searches were 0.19 of the sweep's 0.41 ms).
constexpr int TAB_DESC = 12;  // p0, cnt, chrom, n_lds, S0, S1, shift_s, P0, P1, shift_p, n_seg, -
constexpr int TAB_LUT_S_WORDS = (LUT_NB_S + 2) / 2, TAB_LUT_P_WORDS = (LUT_NB_P + 2) / 2;
constexpr int TAB_LUT_WORDS = TAB_LUT_S_WORDS + TAB_LUT_P_WORDS;
constexpr int TAB_WORDS = (TAB_DESC + TAB_LUT_WORDS + 3) / 4 * 4;
static_assert(TAB_WORDS == (int)IGD_TILE_TAB_WORDS, "common.h states the table size");

__global__ void __launch_bounds__(SW_TPB)
k_igd_tile_tables(IgdView v, const u32 *__restrict__ tile_first, const u32 *__restrict__ tile_cnt, const u32 *__restrict__ tile_chrom,
                  const i32 *__restrict__ tile_carry, u32 n_tiles, i32 *__restrict__ pm_out, unsigned short *__restrict__ files16,
                  u32 *__restrict__ tile_tab) {
    constexpr int CAP = IGD_TILE + IGD_HALO;
    constexpr int RPT = (CAP + SW_TPB - 1) / SW_TPB;
    __shared__ i32 t_s[CAP], t_e[CAP], t_pm[CAP];
    __shared__ unsigned short lut_s[LUT_NB_S + 2], lut_p[LUT_NB_P + 2];
    __shared__ i32 s_wmax[SW_TPB / 64];
    const u32 t = blockIdx.x;
    if (t >= n_tiles) return;
    const u32 p0 = tile_first[t], cnt = tile_cnt[t], c = tile_chrom[t];
    const u32 seg_hi = v.chrom_off[c + 1];
    const u32 n = min((u32)CAP, seg_hi - p0);  // tile + halo, never past the chromosome
    const i32 carry = tile_carry[t];
    for (u32 i = threadIdx.x; i < n; i += SW_TPB) {
        t_s[i] = v.starts[p0 + i];
        t_e[i] = v.ends[p0 + i];
        if (i < cnt && files16) {
            const u32 f = v.files[p0 + i];  // (pieces view: the continuation flag moves from bit 31 to bit 15; ids are < 16384 there)
            files16[p0 + i] = (unsigned short)((f & 0x7FFFu) | ((f >> 31) << 15));
        }
    }
    __syncthreads();
    // t_pm[i] = max(carry, ends[0..i]); blocked layout: thread k owns records [k * RPT, (k + 1) * RPT)
    const u32 base = threadIdx.x * RPT;
    i32 loc[RPT], run = 0;  // ends are > 0 (Igd::add drop rule)
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const u32 i = base + k;
        run = max(run, i < n ? t_e[i] : 0);
        loc[k] = run;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const i32 inc = wave_inclusive_max_nonneg(run);
    if (lane == 63) s_wmax[wave] = inc;
    const i32 excl = __builtin_amdgcn_update_dpp(0, inc, 0x138, 0xf, 0xf, false);  // wave_shr:1; lane 0 gets 0
    __syncthreads();
    i32 before = carry, all_max = carry;
    for (int w = 0; w < SW_TPB / 64; ++w) {
        const i32 x = s_wmax[w];
        before = max(before, w < wave ? x : 0);
        all_max = max(all_max, x);
    }
    before = max(before, excl);
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const u32 i = base + k;
        if (i < n) {
            const i32 pmv = max(before, loc[k]);
            t_pm[i] = pmv;
            if (i < cnt) pm_out[p0 + i] = pmv;  // own records: the chromosome-wide prefix maximum, the same for every tile
        }
    }
    const i32 S0 = t_s[0], S1 = t_s[n - 1], P0 = max(carry, t_e[0]), P1 = all_max;
    const u32 sh_s = lut_shift<LUT_NB_S>((u32)(S1 - S0)), sh_p = lut_shift<LUT_NB_P>((u32)(P1 - P0));
    {
        // the buckets behind the last key ("no such record"), disjoint from the buckets the loop below writes
        const u32 last_s = (u32)(S1 - S0) >> sh_s, last_p = (u32)(P1 - P0) >> sh_p;
        for (u32 b = threadIdx.x; b < (u32)LUT_NB_S + 2; b += SW_TPB)
            if (b > last_s) lut_s[b] = (unsigned short)n;
        for (u32 b = threadIdx.x; b < (u32)LUT_NB_P + 2; b += SW_TPB)
            if (b > last_p) lut_p[b] = (unsigned short)n;
    }
    __syncthreads();
    // buckets (bucket of the previous key, bucket of this key] start at record i; record 0 opens bucket 0
    for (u32 i = threadIdx.x; i < n; i += SW_TPB) {
        const i32 st = t_s[i], pmv = t_pm[i];
        const u32 bs1 = (u32)(st - S0) >> sh_s, bp1 = (u32)(pmv - P0) >> sh_p;
        u32 bs0 = 0, bp0 = 0;
        if (i) {
            bs0 = ((u32)(t_s[i - 1] - S0) >> sh_s) + 1u;
            bp0 = ((u32)(t_pm[i - 1] - P0) >> sh_p) + 1u;
        }
        for (; bs0 <= bs1; ++bs0) lut_s[bs0] = (unsigned short)i;
        for (; bp0 <= bp1; ++bp0) lut_p[bp0] = (unsigned short)i;
    }
    __syncthreads();
    u32 *tab = tile_tab + (size_t)t * TAB_WORDS;
    if (threadIdx.x == 0) {
        tab[0] = p0;
        tab[1] = cnt;
        tab[2] = c;
        tab[3] = n;
        tab[4] = (u32)S0;
        tab[5] = (u32)S1;
        tab[6] = sh_s;
        tab[7] = (u32)P0;
        tab[8] = (u32)P1;
        tab[9] = sh_p;
        tab[10] = seg_hi - p0;
        tab[11] = 0;
    }
    for (u32 w = threadIdx.x; w < (u32)TAB_LUT_WORDS; w += SW_TPB) {
        const unsigned short *src = w < (u32)TAB_LUT_S_WORDS ? lut_s : lut_p;
        const u32 k = w < (u32)TAB_LUT_S_WORDS ? w : w - TAB_LUT_S_WORDS;
        tab[TAB_DESC + w] = (u32)src[2 * k] | ((u32)src[2 * k + 1] << 16);
    }
}

gtars_status launch_igd_tile_tables(const IgdView &v, const u32 *tile_first, const u32 *tile_cnt, const u32 *tile_chrom,
                                    const i32 *tile_carry, u32 n_tiles, i32 *pm, unsigned short *files16, u32 *tile_tab, hipStream_t st) {
    if (!n_tiles) return GTARS_OK;
    hipLaunchKernelGGL(k_igd_tile_tables, dim3(n_tiles), dim3(SW_TPB), 0, st, v, tile_first, tile_cnt, tile_chrom, tile_carry, n_tiles, pm,
                       files16, tile_tab);
    GT_HIP(hipGetLastError());
    return GTARS_OK;
}

// ---- the sweep -------------------------------------------------------------------------------------------------------
// Query k of a block of SW_TPB goes to wave k % NWV, group (k / NWV) % NG, slot k / (NWV * NG): the first 64 queries of a tile
// land on 64 different (wave, group) pairs, so a sparse batch (a LOLA universe: ~40 queries per tile) still spreads over every
// wave, and a dense one gives every group the same number of queries.
__device__ __forceinline__ u32 sweep_query_slot(int lane, int wave) {
    constexpr int GL = IGD_GROUP_LANES, NG = 64 / GL, NWV = SW_TPB / 64;
    return ((u32)(lane % GL) * NG + (u32)(lane / GL)) * NWV + (u32)wave;
}

// MODE 0: pairwise counts (count_set_overlaps); 1: binary counts (count_region_hits) with a per-query list of credited
// files; 2: binary counts for min_overlap == 1 through pme_file -- a record is the FIRST hit of its file for a query iff
// no earlier record of that file (and chromosome) ends after the query's start: earlier records start no later, so
// "ends after q.start" is all that is left of the overlap test, and the largest such end is a per-record constant of
// the database (IgdTiles::pme_file).  Binary counting then costs what pairwise counting costs.
// MO1: min_overlap == 1 (a candidate record is a hit iff its end is > q_start; its start is not read).
// PIECES: the database is a pieces view (IgdView::pieces): bit 15 of a staged file id marks a continuation piece, which is a
// hit only if it also starts at or before the query's start (then it is the piece that holds the query's start; first pieces
// count whenever they overlap) -- min_overlap == 1 forms only.
//
// What a tile streams (round 4): starts, ends, u16 file ids (+ pme_file for MODE 2) and the two search tables -- NOT the
// prefix maxima.  A query's first candidate used to be found exactly (binary search of the staged prefix-max column, 4 bytes per
// record from HBM and in LDS, three dependent LDS round trips per query); now it is the LOWER BRACKET of the static table over
// the prefix maxima (IgdTiles::tab): every record in front of it has a prefix-max end <= q_start and so cannot overlap, and the
// records between the bracket and the exact position (about one on average with 1024 buckets) end at or before q_start too --
// they fail the hit test like any other candidate that does not overlap.  A bracket wider than LO_REFINE records (prefix maxima
// that stay flat behind a long record) is narrowed by a search of the prefix maxima in global memory (IgdView::pm; rare).
// LDS per workgroup: starts (4 B) + {end, file} or {end, pme_file} pairs (8 B, ONE ds_read_b64 per candidate) [+ u16 file ids for
// MODE 2]: 28 / 32 KB instead of 32 / 42, so the pme_file form keeps 4 workgroups per CU.
constexpr u32 LO_REFINE = 24;
// SIO: the instantiation that serves several query sets in order (n_sets_io rows of tile ranges); the others compile the one-set
// loop as it was (the set bookkeeping cost the one-set binary sweep 1.6 % when it shared the instantiation)
template <int MODE, bool MO1, bool B16 = false, bool PIECES = false, bool SIO = false>
__global__ void __launch_bounds__(SW_TPB, MODE == 1 ? 4 : 8)
k_igd_sweep(IgdView v, const i32 *__restrict__ pme_file, const i32 *__restrict__ pm, const unsigned short *__restrict__ files16,
            const u32 *__restrict__ tile_tab, u32 n_tiles, const u32 *__restrict__ sqs, const u32 *__restrict__ sqe, int interleaved,
            const u32 *__restrict__ ql, const u32 *__restrict__ qh, i32 min_overlap, unsigned long long *__restrict__ hits,
            const u32 *__restrict__ part_flag, const u32 *__restrict__ part_ab, const u32 *__restrict__ part_ql, u32 n_bins,
            const uint2 *__restrict__ heavy_list, const u32 *__restrict__ heavy_count, u32 heavy_part, u32 heavy_cap, u32 n_sets_io) {
    extern __shared__ __attribute__((aligned(16))) u32 sm[];
    // whether the batch had to be partitioned was decided on the device (the routing kernel): take the partition's
    // interleaved (start, end) pairs and bin offsets, or the batch as it arrived with the tile ranges
    // n_sets_io > 1: a batch of several query sets that are each in order (k_igd_begin) -- the batch as it arrived, ONE row of tile
    // ranges per set (stride n_tiles); a tile serves its queries set after set, counter row = set.  The partitioned continuation
    // carries the sets as tags of the pairs instead.
    if (part_flag && *part_flag) {
        sqs = part_ab;
        sqe = nullptr;
        interleaved = 1;
        ql = part_ql;
        qh = part_ql + 1;
        n_sets_io = 1;
    }
    if (!SIO) n_sets_io = 1;
    if (SIO && n_sets_io > 1) heavy_part = 0;  // (no parts of heavy tiles in that form)
    constexpr int CAP = IGD_TILE + IGD_HALO;
    // The columns are staged as whole 16-byte vectors from the 16-byte boundary at or below the tile's first record (a dword-per-
    // lane copy is 5 x 5 loads and as many LDS stores per thread and tile: a fifth of the kernel's instructions): LDS slot j of a
    // column holds the record at (p0 & ~3) + j, so record i of the tile sits at slot i + (p0 & 3) -- the per-tile pointers t_*
    // below point there.
    constexpr int CAPV = CAP + 4;
    static_assert(CAPV % 4 == 0, "every column starts on a 16-byte boundary");
    i32 *b_s = reinterpret_cast<i32 *>(sm);
    uint2 *b_x = reinterpret_cast<uint2 *>(b_s + CAPV);  // {end, file id} -- MODE 2: {end, pme_file}
    unsigned short *b_f = reinterpret_cast<unsigned short *>(b_x + CAPV);  // MODE 2: the file ids, u16
    // [n_bins]: n_files counters per query SET.  A partitioned batch may hold up to 4 sets (gtars_igd_count_sets_device): the set
    // of a query travels in bit 31 of its (start, end) pair (SetTags, sort.hip) and selects the row of counters its hits go to
    u32 *bins = reinterpret_cast<u32 *>(b_f + (MODE == 2 ? CAPV : 0));
    auto untag = [&](i32 &s, i32 &e) -> u32 {  // -> the query's first counter
        const u32 set = (((u32)s >> 31) << 1) | ((u32)e >> 31);
        s &= 0x7FFFFFFF;
        e &= 0x7FFFFFFF;
        return __umul24(set, v.n_files);  // < 16384 (the launcher's bound on sets x files)
    };
    static_assert(IGD_TILE + IGD_HALO < 4096, "a candidate count and a counter offset share one word: 12 + 14 bits");
    __shared__ u32 s_lutw[TAB_LUT_WORDS];
    const unsigned short *lut_s = reinterpret_cast<const unsigned short *>(s_lutw);
    const unsigned short *lut_p = reinterpret_cast<const unsigned short *>(s_lutw + TAB_LUT_S_WORDS);
    constexpr bool BINARY = MODE == 1;
    // B16 (MODE 2 only): 16-bit counters, two per LDS word.  A query credits a (set, file) at most once, so a counter is bounded
    // by the number of queries the workgroup has served since its counters were last flushed -- flushed every <= 65535 queries.
    // Half the LDS for the counters; the launcher takes this form only when that admits one more workgroup per CU -- the packed
    // increment costs three more instructions per hit.
    static_assert(!B16 || MODE == 2, "16-bit counters need the one-credit-per-query bound of the pme_file form");
    static_assert(!PIECES || (MO1 && MODE != 1), "a pieces view serves the min_overlap == 1 forms only");
    const u32 n_words = B16 ? (n_bins + 1) / 2 : n_bins;
    auto bump = [&](u32 b) {
        if (B16)
            atomicAdd(&bins[b >> 1], 1u << ((b & 1u) * 16u));
        else
            atomicAdd(&bins[b], 1u);
    };
    auto flush = [&]() {  // (workgroup-uniform; the caller has synchronised)
        for (u32 i = threadIdx.x; i < n_words; i += SW_TPB) {
            const u32 b = bins[i];
            if (!b) continue;
            if (B16) {
                if (b & 0xFFFFu) atomicAdd(&hits[2 * i], (unsigned long long)(b & 0xFFFFu));
                if (b >> 16) atomicAdd(&hits[2 * i + 1], (unsigned long long)(b >> 16));
                bins[i] = 0;
            } else {
                atomicAdd(&hits[i], (unsigned long long)b);
            }
        }
    };
    u32 since = 0;  // queries served since the last flush (B16)
    for (u32 i = threadIdx.x; i < n_words; i += SW_TPB) bins[i] = 0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#if IGD_STAMPS
    u64 st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_last = __builtin_amdgcn_s_memtime();
#endif
    // Work items: every tile's first `heavy_part` queries (item = tile), then the listed further parts of the tiles that own far
    // more queries than the average (HeavyBins: filled by the bin scan or by k_igd_tile_ranges) -- a batch with half of its
    // queries inside one 50-kb window kept ONE workgroup busy for 32 ms (10M queries) before; the parts of such a tile now spread
    // over the grid, each staging the tile again.
    const u32 n_heavy = heavy_part ? min(*heavy_count, heavy_cap) : 0u;
    const u32 n_items = n_tiles + n_heavy;
    // An item's descriptor (12 words of its tile's table + its query range) is REQUESTED one tile ahead: the chain descriptor ->
    // record addresses -> records was two dependent trips to HBM at the head of every tile (the stamps' "stage" phase: 9k of a
    // tile's 25k cycles).  Every wave requests it for itself with ONE vector load -- lane l < 12 word l of the table, lane 12 / 13
    // the tile's query range -- and picks the words out of the register with v_readlane when the tile's turn comes.  (Scalar
    // loads would be the natural form, but the compiler parks the loop-carried values in other SGPRs right behind the loads,
    // i.e. it waits for them where they are issued.)
    auto request = [&](u32 item, u32 &tile, u32 &part) -> u32 {
        tile = item;
        part = 0;
        if (item >= n_tiles) {  // a further part of a heavy tile (rare): which tile, by scalar loads
            const uint2 hp = heavy_list[item - n_tiles];
            tile = hp.x;
            part = hp.y;
        }
        // (lanes 12 + 2k / 13 + 2k: the tile's range of set k)
        const u32 rel = (u32)lane - TAB_DESC;
        const u32 *__restrict__ src = lane < TAB_DESC ? tile_tab + (size_t)tile * TAB_WORDS + lane
                                                      : ((rel & 1u) ? qh : ql) + (size_t)(rel >> 1) * n_tiles + tile;
        return lane < TAB_DESC + 2 * (SIO ? (int)max(n_sets_io, 1u) : 1) ? *src : 0u;
    };
    u32 n_tile = 0, n_part = 0, n_desc = 0;
    if (blockIdx.x < n_items) n_desc = request(blockIdx.x, n_tile, n_part);
    for (u32 item = blockIdx.x; item < n_items; item += gridDim.x) {
        const u32 tile = n_tile, desc = n_desc;
        auto word = [&](int k) -> u32 { return (u32)__builtin_amdgcn_readlane((int)desc, k); };
        const u32 p0 = word(0), n_lds = word(3), n_seg = word(10), sh_s = word(6), sh_p = word(9);
        const i32 S0 = (i32)word(4), S1 = (i32)word(5), P0 = (i32)word(7), P1 = (i32)word(8);
        u32 q_lo = word(TAB_DESC), q_hi = (GTARS_IGD_ABLATE & 8) ? q_lo : word(TAB_DESC + 1);
        if (heavy_part) {  // this item's share of the tile's queries
            q_lo = min(q_hi, q_lo + n_part * heavy_part);
            q_hi = min(q_hi, q_lo + heavy_part);
        }
        const u32 *__restrict__ tab = tile_tab + (size_t)tile * TAB_WORDS;
        if (item + gridDim.x < n_items) n_desc = request(item + gridDim.x, n_tile, n_part);
        // The tile's queries as ONE index space j = 0 .. vt - 1: set 0's range first, then (several sets served in order) the further
        // sets' ranges -- one loop over all of them, so that a small set does not cost a loop turn of its own per tile (a LOLA
        // user set leaves ~4 queries per tile next to the universe's ~41).  c1 <= c2 <= c3: where sets 1, 2, 3 begin in it.
        const u32 n_set0 = q_hi - q_lo;
        u32 lo1 = 0, lo2 = 0, lo3 = 0, c1 = n_set0, c2 = n_set0, c3 = n_set0, vt = n_set0;
        if (SIO && n_sets_io > 1) {
            lo1 = word(TAB_DESC + 2);
            c2 = c3 = vt = c1 + (word(TAB_DESC + 3) - lo1);
        }
        if (SIO && n_sets_io > 2) {
            lo2 = word(TAB_DESC + 4);
            c3 = vt = c2 + (word(TAB_DESC + 5) - lo2);
        }
        if (SIO && n_sets_io > 3) {
            lo3 = word(TAB_DESC + 6);
            vt = c3 + (word(TAB_DESC + 7) - lo3);
        }
        auto query_at = [&](u32 j, u32 &set) -> u32 {  // (j < vt)
            set = 0;
            if (!SIO || n_sets_io <= 1) return q_lo + j;  // (uniform)
            set = (j >= c1 ? 1u : 0u) + (j >= c2 ? 1u : 0u) + (j >= c3 ? 1u : 0u);
            return set == 0 ? q_lo + j : set == 1 ? lo1 + (j - c1) : set == 2 ? lo2 + (j - c2) : lo3 + (j - c3);
        };
        // the tile's first SW_TPB queries: loaded now, used once the tile is staged (MODE 0 / 2)
        i32 pf_s = 0, pf_e = 0;
        if constexpr (MODE != 1) {
            const u32 j0 = sweep_query_slot(lane, wave);
            u32 set0_ = 0;
            const u32 qi = j0 < vt ? query_at(j0, set0_) : 0u;
            if (j0 < vt) {
                if (interleaved) {
                    const uint2 se2 = reinterpret_cast<const uint2 *>(sqs)[qi];
                    pf_s = (i32)se2.x;
                    pf_e = (i32)se2.y;
                } else {
                    pf_s = max((i32)sqs[qi], 0);  // raw column of an in-order batch: the start clamp (igd.rs:517) here
                    pf_e = (i32)sqe[qi];
                }
            }
        }
        // stage the tile: records + the search tables.  No register prefetch across the query phase: with 4 workgroups per
        // CU the other workgroups' query phases cover this one's loads (measured: prefetching one tile ahead bought nothing and
        // cost 15 registers).
        const u32 d4 = p0 & 3u;
        i32 *t_s = b_s + d4;
        uint2 *t_x = b_x + d4;
        unsigned short *t_f = b_f + d4;
        {
            constexpr int RV = (CAPV / 4 + SW_TPB - 1) / SW_TPB;  // 16-byte vectors per thread and 4-byte column
            const u32 nv4 = (n_lds + d4 + 3u) >> 2;
            typedef u32 v4u __attribute__((ext_vector_type(4)));  // (a native vector: arrays of HIP's uint4 struct went to scratch)
            typedef u32 v2u __attribute__((ext_vector_type(2)));
            const v4u *g_s = reinterpret_cast<const v4u *>(v.starts + (p0 - d4)), *g_e = reinterpret_cast<const v4u *>(v.ends + (p0 - d4)),
                      *g_q = MODE == 2 ? reinterpret_cast<const v4u *>(pme_file + (p0 - d4)) : nullptr;
            const v2u *g_f = reinterpret_cast<const v2u *>(files16 + (p0 - d4));  // four u16 ids per vector of four records
            v4u rs[RV], re[RV], rq[MODE == 2 ? RV : 1];
            v2u rf[RV];
#pragma unroll
            for (int k = 0; k < RV; ++k) {
                const u32 q = threadIdx.x + (u32)k * SW_TPB;
                rs[k] = re[k] = v4u{0, 0, 0, 0};
                rf[k] = v2u{0, 0};
                if (MODE == 2) rq[MODE == 2 ? k : 0] = v4u{0, 0, 0, 0};
                if (q < nv4) {
                    rs[k] = g_s[q];
                    re[k] = g_e[q];
                    rf[k] = g_f[q];
                    if (MODE == 2) rq[MODE == 2 ? k : 0] = g_q[q];
                }
            }
            constexpr int LWN = (TAB_LUT_WORDS + SW_TPB - 1) / SW_TPB;  // table words per thread
            u32 lw[LWN];
#pragma unroll
            for (int k = 0; k < LWN; ++k) {
                const u32 w = threadIdx.x + (u32)k * SW_TPB;
                lw[k] = w < (u32)TAB_LUT_WORDS ? tab[TAB_DESC + w] : 0u;
            }
#pragma unroll
            for (int k = 0; k < RV; ++k) {
                const u32 q = threadIdx.x + (u32)k * SW_TPB;
                if (q < nv4) {
                    reinterpret_cast<v4u *>(b_s)[q] = rs[k];
                    v4u x0, x1;
                    if (MODE == 2) {
                        const v4u pq = rq[MODE == 2 ? k : 0];
                        x0 = v4u{re[k].x, pq.x, re[k].y, pq.y};
                        x1 = v4u{re[k].z, pq.z, re[k].w, pq.w};
                        reinterpret_cast<v2u *>(b_f)[q] = rf[k];
                    } else {
                        x0 = v4u{re[k].x, rf[k].x & 0xFFFFu, re[k].y, rf[k].x >> 16};
                        x1 = v4u{re[k].z, rf[k].y & 0xFFFFu, re[k].w, rf[k].y >> 16};
                    }
                    reinterpret_cast<v4u *>(b_x)[2 * q] = x0;
                    reinterpret_cast<v4u *>(b_x)[2 * q + 1] = x1;
                }
            }
#pragma unroll
            for (int k = 0; k < LWN; ++k) {
                const u32 w = threadIdx.x + (u32)k * SW_TPB;
                if (w < (u32)TAB_LUT_WORDS) s_lutw[w] = lw[k];
            }
        }
        __syncthreads();  // the tile is in LDS (and, first tile, the bins are zeroed)
        STAMP(0);
        // first candidate of a query that starts at s: the lower bracket of the prefix-max table (see the kernel's header)
        auto first_candidate = [&](i32 s) -> u32 {
            u32 lo, hi;
            lut_range(lut_p, s + 1, P0, P1, sh_p, n_lds, lo, hi);
            if (hi - lo > LO_REFINE && pm) {  // flat prefix maxima (a long record in front): the exact position, from global memory
                while (lo < hi) {
                    const u32 mid = lo + ((hi - lo) >> 1);
                    if (pm[p0 + mid] <= s)
                        lo = mid + 1;
                    else
                        hi = mid;
                }
            }
            return lo;
        };
        // record i (relative to p0): LDS if staged, global otherwise (rare: scans longer than the halo)
        auto r_start = [&](u32 i) -> i32 { return i < n_lds ? t_s[i] : v.starts[p0 + i]; };
        auto r_end = [&](u32 i) -> i32 { return i < n_lds ? (i32)t_x[i].x : v.ends[p0 + i]; };
        auto r_file = [&](u32 i) -> u32 { return i < n_lds ? (MODE == 2 ? (u32)t_f[i] : t_x[i].y) : v.files[p0 + i]; };
        if constexpr (MODE != 1) {
            // A hit is decided by the (query, record) PAIR alone.  A query's candidates are the records [lo, hi): lo = the first
            // record that can overlap it (bracket of the prefix-max table), hi = first record that starts at or after q_end (the
            // reference's scan stops there, igd.rs:772-846).  Waves work on their own, no workgroup barrier inside a tile:
            // (1) one lane per query finds (lo, hi): one table lookup, and the tile's start table + a short binary search;
            // (2) the pairs are walked by QUADS -- quad g takes the queries of its own 4 lanes one after the other (their state
            // arrives by DPP quad_perm moves), its lanes read CONSECUTIVE records, one 8-byte LDS read each.  Records past the
            // staged range (rare) are scanned by the query's own lane from global memory.
            constexpr int GL = IGD_GROUP_LANES;
            const int sub = lane % GL;
            const u32 kq = sweep_query_slot(lane, wave);
            for (u32 qb = 0; qb < vt; qb += SW_TPB) {
                if (B16) {
                    if (since + SW_TPB > 65535u) {  // (uniform) the 16-bit counters could wrap: hand them over first
                        __syncthreads();
                        flush();
                        __syncthreads();
                        since = 0;
                    }
                    since += SW_TPB;
                }
                const u32 j = qb + kq;
                i32 s = 0, e = 0;
                u32 lo = 0, len = 0, boff = 0;
                if (j < vt) {
                    u32 set;
                    const u32 qi = query_at(j, set);
                    if (qb == 0) {  // loaded at the top of the tile
                        s = pf_s;
                        e = pf_e;
                    } else if (interleaved) {
                        const uint2 se2 = reinterpret_cast<const uint2 *>(sqs)[qi];
                        s = (i32)se2.x;
                        e = (i32)se2.y;
                    } else {
                        s = max((i32)sqs[qi], 0);
                        e = (i32)sqe[qi];
                    }
                    boff = __umul24(set, v.n_files);
                    if (interleaved) boff = untag(s, e);
                    u32 a, b;
                    if (GTARS_IGD_ABLATE & 16) {
                        lo = ((u32)s * 2654435761u) % (n_lds > 40 ? n_lds - 40 : 1u);
                        a = lo + 20;
                    } else {
                        lo = first_candidate(s);
                        // a = first record with start >= e
                        lut_range(lut_s, e, S0, S1, sh_s, n_lds, a, b);
                        while (b - a > 4u) {  // (rare with 1024 buckets over ~2300 staged records)
                            const u32 m2 = a + ((b - a) >> 1);
                            if (t_s[m2] < e)
                                a = m2 + 1;
                            else
                                b = m2;
                        }
                        if (a < b) {
                            // the bracket's <= 4 starts by ONE 16-byte LDS read (4-byte aligned); they ascend, so the answer is a +
                            // the number of them below e.  (Slots past the bracket may lie past the staged records: ignored.)
                            typedef i32 i4 __attribute__((ext_vector_type(4)));
                            typedef i4 i4_a4 __attribute__((aligned(4)));
                            typedef const __attribute__((address_space(3))) i4_a4 *lds_i4;
                            const i4 x = *(lds_i4)(uintptr_t)(t_s + a);
                            const u32 nb = b - a;
                            a += (x.x < e ? 1u : 0u) + (nb > 1 && x.y < e ? 1u : 0u) + (nb > 2 && x.z < e ? 1u : 0u) + (nb > 3 && x.w < e ? 1u : 0u);
                        }
                    }
                    len = a > lo ? a - lo : 0u;  // (a >= lo: every record before lo ends at or before s, hence starts before e)
                    if (!(GTARS_IGD_ABLATE & 16) && a == n_lds && n_seg > n_lds) {
                        // the scan runs past the staged records: the rest from global memory, by this lane
                        for (u32 r = max(lo, n_lds); r < n_seg; ++r) {
                            const i32 rs = v.starts[p0 + r], re = v.ends[p0 + r];
                            if (rs >= e) break;
                            const i32 ov = (re < e ? re : e) - (rs > s ? rs : s);
                            if (ov < min_overlap) continue;
                            if (MODE == 2 && pme_file[p0 + r] > s) continue;
                            const u32 fr = v.files[p0 + r];
                            if (PIECES && (fr >> 31) && rs > s) continue;  // a continuation piece that does not hold the query's start
                            bump(boff + (fr & IGD_FILE_MASK));
                        }
                    }
                }
                if (GTARS_IGD_ABLATE & 1) len = min(len, 1u);
                STAMP(3);
                const u32 len_bo = len | (boff << 12);
                // the quad's four queries, one after the other
                // WALK_U candidates of a lane per step, their LDS reads issued together (inline asm: left to itself the compiler
                // reads a pair's end, branches on the overlap test and only then reads the file id -- two dependent LDS round
                // trips for every candidate, 490 cycles per step by the in-kernel stamps with 8 waves per SIMD).  Lanes past the
                // query's last candidate read whatever follows in LDS (in range of the allocation or zero) and ignore it.
                constexpr int WALK_U = 3;
                auto walk = [&](auto K) {
                    constexpr int k = decltype(K)::value;
                    const i32 qs_ = quad_bcast<k>(s);
                    const u32 lo_ = (u32)quad_bcast<k>((int)lo), lb = (u32)quad_bcast<k>((int)len_bo);
                    const i32 qe_ = MO1 ? 0 : quad_bcast<k>(e);
                    const u32 len_ = lb & 0xFFFu, bo_ = lb >> 12;
                    constexpr bool NEED_S = !(MO1 && !PIECES);
                    for (u32 j = (u32)sub; j < len_; j += GL * WALK_U) {
                        const u32 r = lo_ + j;
                        const u32 ax = (u32)(uintptr_t)(t_x + r), as = (u32)(uintptr_t)(t_s + r), af = (u32)(uintptr_t)(t_f + r);
                        u64 x[WALK_U];
                        u32 fs[WALK_U] = {0, 0, 0};
                        i32 ss[WALK_U] = {0, 0, 0};
                        static_assert(WALK_U == 3 && GL == 4, "the offsets below");
                        if (MODE == 2 && NEED_S)
                            asm volatile("ds_read_b64 %0, %9\n\tds_read_b64 %1, %9 offset:32\n\tds_read_b64 %2, %9 offset:64\n\t"
                                         "ds_read_u16 %3, %10\n\tds_read_u16 %4, %10 offset:8\n\tds_read_u16 %5, %10 offset:16\n\t"
                                         "ds_read_b32 %6, %11\n\tds_read_b32 %7, %11 offset:16\n\tds_read_b32 %8, %11 offset:32\n\t"
                                         "s_waitcnt lgkmcnt(0)"
                                         : "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(fs[0]), "=&v"(fs[1]), "=&v"(fs[2]), "=&v"(ss[0]), "=&v"(ss[1]), "=&v"(ss[2])
                                         : "v"(ax), "v"(af), "v"(as)
                                         : "memory");
                        else if (MODE == 2)
                            asm volatile("ds_read_b64 %0, %6\n\tds_read_b64 %1, %6 offset:32\n\tds_read_b64 %2, %6 offset:64\n\t"
                                         "ds_read_u16 %3, %7\n\tds_read_u16 %4, %7 offset:8\n\tds_read_u16 %5, %7 offset:16\n\t"
                                         "s_waitcnt lgkmcnt(0)"
                                         : "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(fs[0]), "=&v"(fs[1]), "=&v"(fs[2])
                                         : "v"(ax), "v"(af)
                                         : "memory");
                        else if (NEED_S)
                            asm volatile("ds_read_b64 %0, %6\n\tds_read_b64 %1, %6 offset:32\n\tds_read_b64 %2, %6 offset:64\n\t"
                                         "ds_read_b32 %3, %7\n\tds_read_b32 %4, %7 offset:16\n\tds_read_b32 %5, %7 offset:32\n\t"
                                         "s_waitcnt lgkmcnt(0)"
                                         : "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(ss[0]), "=&v"(ss[1]), "=&v"(ss[2])
                                         : "v"(ax), "v"(as)
                                         : "memory");
                        else
                            asm volatile("ds_read_b64 %0, %3\n\tds_read_b64 %1, %3 offset:32\n\tds_read_b64 %2, %3 offset:64\n\t"
                                         "s_waitcnt lgkmcnt(0)"
                                         : "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2])
                                         : "v"(ax)
                                         : "memory");
#pragma unroll
                        for (int u = 0; u < WALK_U; ++u) {
                            const i32 re = (i32)(u32)x[u];
                            const u32 hi32 = (u32)(x[u] >> 32);
                            u32 f = MODE == 2 ? fs[u] : hi32;
                            const i32 rs = ss[u];
                            bool hit;
                            if (MO1)
                                hit = re > qs_;
                            else
                                hit = (re < qe_ ? re : qe_) - (rs > qs_ ? rs : qs_) >= min_overlap;
                            if (MODE == 2) hit = hit && (i32)hi32 <= qs_;  // no earlier record of this file reaches the query
                            if (PIECES) {
                                hit = hit && (!(f >> 15) || rs <= qs_);
                                f &= 0x7FFFu;
                            }
                            if (hit && j + (u32)(u * GL) < len_) {
                                if (GTARS_IGD_ABLATE & 2) {
                                    if (re == 0x7FFFFFF0) bins[0] = 1;
                                } else {
                                    bump(bo_ + f);
                                }
                            }
                        }
                    }
                };
                walk(std::integral_constant<int, 0>{});
                walk(std::integral_constant<int, 1>{});
                walk(std::integral_constant<int, 2>{});
                walk(std::integral_constant<int, 3>{});
                STAMP(4);
            }
        } else
        for (u32 qi = q_lo + threadIdx.x; qi < q_hi; qi += SW_TPB) {
            // (start, end) pairs as the partition leaves them, or two sorted columns
            i32 s, e;
            u32 boff = 0;
            if (interleaved) {
                const uint2 se2 = reinterpret_cast<const uint2 *>(sqs)[qi];
                s = (i32)se2.x;
                e = (i32)se2.y;
                boff = untag(s, e);
            } else {
                s = max((i32)sqs[qi], 0);
                e = (i32)sqe[qi];
            }
            // an overlap of >= 1 bp needs end > q_start: start at the first candidate (every staged record in front of it ends at
            // or before q_start).  It is never before lower_bound(q_start - max_len), so ownership by this tile still holds; if no
            // staged record qualifies the scan goes on in global memory from the end of the staged range.
            const u32 lo = first_candidate(s);
            // binary counting: the files already credited to this query, as packed u16 pairs in
            // registers (no memory latency in the membership test); 0xFFFF = empty (n_files <= 16384)
            u32 n_seen = 0;
            u32 sl[IGD_SEEN / 2];
#pragma unroll
            for (int k = 0; k < IGD_SEEN / 2; ++k) sl[k] = 0xFFFFFFFFu;
            for (u32 r = lo; r < ((GTARS_IGD_ABLATE & 1) ? min(n_seg, lo + 1u) : n_seg); ++r) {
                i32 rs, re;
                u32 f;
                if (r < n_lds) {
                    const uint2 x = t_x[r];
                    rs = t_s[r];
                    re = (i32)x.x;
                    f = x.y;
                } else {
                    rs = v.starts[p0 + r];
                    re = v.ends[p0 + r];
                    f = v.files[p0 + r];
                }
                if (rs >= e) break;
                const i32 ov = (re < e ? re : e) - (rs > s ? rs : s);
                if (ov < min_overlap) continue;
                if (BINARY) {
                    // credit (query, file) only at its first hit in database order (igd.rs:563-590):
                    // compare with the files already credited to this query
                    bool first = true;
#pragma unroll
                    for (int k = 0; k < IGD_SEEN / 2; ++k)
                        first = first && ((sl[k] & 0xFFFFu) != f) && ((sl[k] >> 16) != f);
                    if (first && n_seen >= (u32)IGD_SEEN) {
                        // list overflow (a query with very many distinct files): exact look-back over the
                        // earlier records
                        for (u32 k = lo; k < r; ++k) {
                            if (r_file(k) == f) {
                                const i32 ks = r_start(k), ke = r_end(k);
                                if ((ke < e ? ke : e) - (ks > s ? ks : s) >= min_overlap) {
                                    first = false;
                                    break;
                                }
                            }
                        }
                    }
                    if (!first) continue;
#pragma unroll
                    for (int k = 0; k < IGD_SEEN / 2; ++k) {
                        if ((u32)k == (n_seen >> 1)) sl[k] = (n_seen & 1u) ? ((sl[k] & 0xFFFFu) | (f << 16)) : ((sl[k] & 0xFFFF0000u) | f);
                    }
                    ++n_seen;
                }
                if (GTARS_IGD_ABLATE & 2) {
                    if (f == 0xFFFFFFF0u) bins[0] = 1;
                } else {
                    atomicAdd(&bins[boff + f], 1u);
                }
            }
        }
        STAMP(3);
        __syncthreads();  // every query of the current tile served: LDS may be overwritten
        STAMP(5);
    }
#if IGD_STAMPS
    if (threadIdx.x == 0) {
        for (int k = 0; k < 7; ++k) atomicAdd(&g_sweep_stamps[k], st_acc[k]);
        atomicAdd(&g_sweep_stamps[7], 1ull);
    }
#endif
    __syncthreads();
    flush();
}


// ---- the rank-histogram sweep (round 5): pairwise counts for min_overlap == 1 without a candidate walk ------------------------
// k_igd_sweep above walks every query's candidate records in LDS and adds every hit to its file's counter: ~21 LDS reads and ~15
// LDS atomics on random banks per config-3 query, and the SQ counters say the LDS pipe is what bounds it (71 % busy, 40 % of that
// bank conflicts).  For min_overlap == 1 a (query, record) pair is a hit iff  r.start < q.end  and  r.end > q.start, and since a
// valid query has q.start < q.end and a stored record r.start < r.end (igd.rs:114-116, 514-517),  r.end <= q.start  implies
// r.start < q.end.  So over the tile's staged records R and the queries Q it owns
//     hits(r) = #{q : r.start < q.end} - #{q : r.end <= q.start}
//             = #{q : a_q > r} - #{q : e_q > erank(r)},        a_q = #{r in R : r.start < q.end}   (a RANK among the starts),
//                                                              e_q = #{r in R : r.end <= q.start}  (a RANK among the SORTED ends),
// erank(r) = position of r's end in the tile's ends sorted ascending.  A query therefore costs two table-assisted searches and
// TWO LDS atomics (histograms of a_q and e_q) whatever its number of hits; per tile two prefix sums turn the histograms into
// PA[r] = #{q : a_q <= r} and PE[k] = #{q : e_q <= k}, hits(r) = PE[erank(r)] - PA[r], and one add per record with hits goes to
// its file's counter.  What it needs from the database, built once per tile with the index (k_igd_tile_tables_rank): the staged
// records' ends in ascending order, erank as u16, and search tables over the starts and the sorted ends -- 12 bytes per staged
// record streamed (starts, sorted ends, u16 file ids, u16 eranks) instead of 10; the raw ends are not read at all.
// Records past the staged range (scans longer than the halo) are counted by the query's own lane from global memory, as above.
constexpr int TABR_DESC = 12;  // p0, cnt, chrom, n_lds, S0, S1, shift_s, E0, E1, shift_e, n_seg, -
constexpr int LUT_NB_R = (int)IGD_LUT_R_NB;
constexpr int TABR_LUT1 = (LUT_NB_R + 2) / 2;  // words of one table
constexpr int TABR_WORDS = (int)IGD_TILE_TABR_WORDS;
static_assert(TABR_WORDS >= TABR_DESC + 2 * TABR_LUT1, "common.h states the table size");

__global__ void __launch_bounds__(SW_TPB)
k_igd_tile_tables_rank(IgdView v, const u32 *__restrict__ tile_first, const u32 *__restrict__ tile_cnt, const u32 *__restrict__ tile_chrom,
                       u32 n_tiles, i32 *__restrict__ ends_sorted, unsigned short *__restrict__ erank, u32 *__restrict__ tab_r) {
    constexpr int CAP = IGD_TILE + IGD_HALO;
    constexpr int NP = 4096;  // size of the sorting network (>= CAP, a power of two)
    static_assert(CAP <= NP, "the network holds a tile and its halo");
    __shared__ u64 key[NP];  // (end << 32) | staged index; padding sorts last
    __shared__ i32 t_s[CAP];
    __shared__ unsigned short lut_s[LUT_NB_R + 2], lut_e[LUT_NB_R + 2];
    const u32 t = blockIdx.x;
    if (t >= n_tiles) return;
    const u32 p0 = tile_first[t], cnt = tile_cnt[t], c = tile_chrom[t];
    const u32 seg_hi = v.chrom_off[c + 1];
    const u32 n = min((u32)CAP, seg_hi - p0);  // tile + halo, never past the chromosome
    for (u32 i = threadIdx.x; i < (u32)NP; i += SW_TPB) key[i] = i < n ? ((u64)(u32)v.ends[p0 + i] << 32) | i : ~0ull;  // (ends > 0)
    for (u32 i = threadIdx.x; i < n; i += SW_TPB) t_s[i] = v.starts[p0 + i];
    __syncthreads();
    for (u32 k = 2; k <= (u32)NP; k <<= 1)
        for (u32 j = k >> 1; j > 0; j >>= 1) {
            for (u32 i = threadIdx.x; i < (u32)NP; i += SW_TPB) {
                const u32 x = i ^ j;
                if (x > i) {
                    const u64 a = key[i], b = key[x];
                    if ((a > b) == ((i & k) == 0)) {
                        key[i] = b;
                        key[x] = a;
                    }
                }
            }
            __syncthreads();
        }
    const size_t blk = (size_t)t * IGD_TILE_BLOCK;
    const u32 d4 = p0 & 3u;
    for (u32 k = threadIdx.x; k < IGD_TILE_BLOCK; k += SW_TPB) {
        ends_sorted[blk + k] = k < n ? (i32)(u32)(key[k] >> 32) : 0x7FFFFFFF;
        if (k < d4 || k >= d4 + n) erank[blk + k] = 0;
    }
    for (u32 k = threadIdx.x; k < n; k += SW_TPB) erank[blk + d4 + (u32)(key[k] & 0xFFFFFFFFull)] = (unsigned short)k;
    const i32 S0 = t_s[0], S1 = t_s[n - 1], E0 = (i32)(u32)(key[0] >> 32), E1 = (i32)(u32)(key[n - 1] >> 32);
    const u32 sh_s = lut_shift<LUT_NB_R>((u32)(S1 - S0)), sh_e = lut_shift<LUT_NB_R>((u32)(E1 - E0));
    {
        const u32 last_s = (u32)(S1 - S0) >> sh_s, last_e = (u32)(E1 - E0) >> sh_e;
        for (u32 b = threadIdx.x; b < (u32)LUT_NB_R + 2; b += SW_TPB) {
            if (b > last_s) lut_s[b] = (unsigned short)n;
            if (b > last_e) lut_e[b] = (unsigned short)n;
        }
    }
    __syncthreads();
    for (u32 i = threadIdx.x; i < n; i += SW_TPB) {
        const i32 st = t_s[i], en = (i32)(u32)(key[i] >> 32);
        const u32 bs1 = (u32)(st - S0) >> sh_s, be1 = (u32)(en - E0) >> sh_e;
        u32 bs0 = 0, be0 = 0;
        if (i) {
            bs0 = ((u32)(t_s[i - 1] - S0) >> sh_s) + 1u;
            be0 = ((u32)((i32)(u32)(key[i - 1] >> 32) - E0) >> sh_e) + 1u;
        }
        for (; bs0 <= bs1; ++bs0) lut_s[bs0] = (unsigned short)i;
        for (; be0 <= be1; ++be0) lut_e[be0] = (unsigned short)i;
    }
    __syncthreads();
    u32 *tab = tab_r + (size_t)t * TABR_WORDS;
    if (threadIdx.x == 0) {
        tab[0] = p0;
        tab[1] = cnt;
        tab[2] = c;
        tab[3] = n;
        tab[4] = (u32)S0;
        tab[5] = (u32)S1;
        tab[6] = sh_s;
        tab[7] = (u32)E0;
        tab[8] = (u32)E1;
        tab[9] = sh_e;
        tab[10] = seg_hi - p0;
        tab[11] = 0;
    }
    for (u32 w = threadIdx.x; w < (u32)(2 * TABR_LUT1); w += SW_TPB) {
        const unsigned short *src = w < (u32)TABR_LUT1 ? lut_s : lut_e;
        const u32 k = w < (u32)TABR_LUT1 ? w : w - TABR_LUT1;
        tab[TABR_DESC + w] = (u32)src[2 * k] | ((u32)src[2 * k + 1] << 16);
    }
}

gtars_status launch_igd_tile_tables_rank(const IgdView &v, const u32 *tile_first, const u32 *tile_cnt, const u32 *tile_chrom, u32 n_tiles,
                                         i32 *ends_sorted, unsigned short *erank, u32 *tab_r, hipStream_t st) {
    if (!n_tiles) return GTARS_OK;
    hipLaunchKernelGGL(k_igd_tile_tables_rank, dim3(n_tiles), dim3(SW_TPB), 0, st, v, tile_first, tile_cnt, tile_chrom, n_tiles, ends_sorted,
                       erank, tab_r);
    GT_HIP(hipGetLastError());
    return GTARS_OK;
}

// One global_load_lds_dwordx4: 16 bytes per lane from the lane's own global address straight into LDS at (wave-uniform) `lds` +
// lane x 16.  In inline assembly ON PURPOSE: the compiler knows nothing of these transfers, so it neither drains the vector-memory
// queue in front of every later LDS read (it cannot tell which LDS bytes a transfer writes) nor turns its counted waits for
// ordinary loads into vmcnt(0) while one is in flight (both seen in the listing of the builtin form).  The caller counts them.
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"  // (m0 IS written by the statement: say so, reserved or not)
__device__ __forceinline__ void glds16(const void *g, const void *lds) {
    const u32 base = (u32)__builtin_amdgcn_readfirstlane((int)(u32)(uintptr_t)lds);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(base) : "memory", "m0");
}
#pragma clang diagnostic pop
constexpr int RK_CAPV = IGD_TILE + IGD_HALO + 4;
constexpr int RK_HW = (RK_CAPV / 2 + 3) / 4 * 4;  // words of one histogram: 16-bit counters two per word, whole 16-byte vectors
constexpr u32 RK_SUB = 65024;                     // queries per histogram fill: what 16-bit counters hold (a multiple of SW_TPB)
static_assert(RK_SUB % SW_TPB == 0 && RK_SUB <= 65535, "16-bit counters");
size_t igd_sweep_rank_lds_bytes(u32 n_bins) { return ((size_t)RK_CAPV * 2 + (size_t)RK_HW * 2 + n_bins) * 4; }

__global__ void __launch_bounds__(SW_TPB, 8)
k_igd_sweep_rank(IgdView v, const unsigned short *__restrict__ files16, const i32 *__restrict__ ends_sorted,
                 const unsigned short *__restrict__ erank, const u32 *__restrict__ tab_r, u32 n_tiles, const u32 *__restrict__ sqs,
                 const u32 *__restrict__ sqe, int interleaved, const u32 *__restrict__ ql, const u32 *__restrict__ qh,
                 unsigned long long *__restrict__ hits, const u32 *__restrict__ part_flag, const u32 *__restrict__ part_ab,
                 const u32 *__restrict__ part_ql, u32 n_bins, const uint2 *__restrict__ heavy_list, const u32 *__restrict__ heavy_count,
                 u32 heavy_part, u32 heavy_cap) {
    extern __shared__ __attribute__((aligned(16))) u32 sm[];
    if (part_flag && *part_flag) {  // (the batch was partitioned: see k_igd_sweep)
        sqs = part_ab;
        sqe = nullptr;
        interleaved = 1;
        ql = part_ql;
        qh = part_ql + 1;
    }
    constexpr int CAPV = RK_CAPV;
    constexpr u32 NW = SW_TPB / 64;
    i32 *b_s = reinterpret_cast<i32 *>(sm);         // starts: slot j holds the record at (p0 & ~3) + j
    i32 *b_e = b_s + CAPV;                          // the staged records' ends, ascending
    u32 *ha = reinterpret_cast<u32 *>(b_e + CAPV);  // histogram of a_q over the starts' SLOTS, then its prefix sums inside a chunk
    u32 *ge = ha + RK_HW;                           // histogram of e_q over the sorted ends, then its prefix sums inside a chunk
    u32 *bins = ge + RK_HW;
    constexpr u32 LUTV = (2 * TABR_LUT1 + 3) / 4;   // 16-byte vectors of the two search tables
    static_assert(TABR_DESC % 4 == 0 && TABR_DESC + 4 * (int)LUTV <= TABR_WORDS, "the tables are copied as whole vectors of their row");
    __shared__ __attribute__((aligned(16))) u32 s_lutw[4 * LUTV];
    // chunks of 256 slots (a wave's four per lane): round 0 = waves 0 .. NW - 1, round 1 = the vectors beyond SW_TPB
    constexpr u32 R1_WAVES = (RK_CAPV / 4 - SW_TPB + 63) / 64, NCH = NW + R1_WAVES;
    static_assert(NCH <= 16, "the chunks' bases are scanned inside one DPP row");
    __shared__ u32 s_tot[2][16];  // [array][chunk]: the chunks' totals of the two prefix sums
    const unsigned short *lut_s = reinterpret_cast<const unsigned short *>(s_lutw);
    const unsigned short *lut_e = reinterpret_cast<const unsigned short *>(s_lutw + TABR_LUT1);
    for (u32 i = threadIdx.x; i < n_bins; i += SW_TPB) bins[i] = 0;
    if (threadIdx.x < 32u) s_tot[threadIdx.x >> 4][threadIdx.x & 15u] = 0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#if IGD_STAMPS
    u64 st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_last = __builtin_amdgcn_s_memtime();
#endif
    const u32 n_heavy = heavy_part ? min(*heavy_count, heavy_cap) : 0u;
    const u32 n_items = n_tiles + n_heavy;
    auto request = [&](u32 item, u32 &tile, u32 &part) -> u32 {  // (as in k_igd_sweep: one vector load per wave)
        tile = item;
        part = 0;
        if (item >= n_tiles) {
            const uint2 hp = heavy_list[item - n_tiles];
            tile = hp.x;
            part = hp.y;
        }
        const u32 *__restrict__ src = lane < TABR_DESC ? tab_r + (size_t)tile * TABR_WORDS + lane : (lane == TABR_DESC ? ql : qh) + tile;
        return lane < TABR_DESC + 2 ? *src : 0u;
    };
    typedef u32 v4u __attribute__((ext_vector_type(4)));
    typedef u32 v2u __attribute__((ext_vector_type(2)));
    // A tile's starts, sorted ends and search tables go from global memory straight into LDS (global_load_lds_dwordx4: the
    // destination is a wave-uniform base + lane x 16, so wave w always fills vectors [64 w, 64 w + 64) of an array -- its own
    // region, tile after tile), REQUESTED while the previous tile's prefix sums and adds run: those three arrays are dead once
    // the previous tile's queries are ranked.  The file ids and eranks of a thread's own slots (and the tile's first queries)
    // follow into registers when the previous tile's are dead.  Nothing in the loop waits for a load it has just issued, and no
    // barrier drains the vector-memory queue (lds_barrier; the one explicit vmcnt(0) sits at the top of a tile).
    i32 pf_s = 0, pf_e = 0;
    auto issue_dma = [&](u32 tile, u32 p0, u32 n_lds, u32 q_lo, u32 q_hi) {
        const u32 d4 = p0 & 3u, nv4 = (n_lds + d4 + 3u) >> 2, nv4e = (n_lds + 3u) >> 2;
        const u32 *g_s = reinterpret_cast<const u32 *>(v.starts + (p0 - d4));
        const u32 *g_e = reinterpret_cast<const u32 *>(ends_sorted + (size_t)tile * IGD_TILE_BLOCK);
        const u32 *g_t = tab_r + (size_t)tile * TABR_WORDS + TABR_DESC;
#pragma unroll
        for (u32 k = 0; k < 2; ++k) {
            const u32 v0 = k * SW_TPB + (u32)wave * 64u, q = v0 + (u32)lane;  // (v0: wave-uniform)
            if (k == 0 || (u32)wave < R1_WAVES) {
                if (q < nv4) glds16(g_s + 4u * q, b_s + 4 * v0);
                if (q < nv4e) glds16(g_e + 4u * q, b_e + 4 * v0);
            }
        }
        {
            const u32 v0 = (u32)wave * 64u, q = v0 + (u32)lane;
            if (v0 < LUTV && q < LUTV) glds16(g_t + 4u * q, s_lutw + 4 * v0);
        }
        // the tile's first SW_TPB queries (the previous tile's are ranked): ONE code path for pairs and for columns -- two loads
        // from selected addresses; a branch per form made the compiler wait for everything in flight between its arms.  (Raw
        // values: the start clamp of a column batch is applied where they are used -- here it would be a wait for the load.)
        const u32 qi = q_lo + threadIdx.x;
        pf_s = pf_e = 0;
        if (qi < q_hi) {
            const u32 *ps = interleaved ? sqs + 2u * (size_t)qi : sqs + qi;
            const u32 *pe = interleaved ? sqs + 2u * (size_t)qi + 1 : sqe + qi;
            pf_s = (i32)*ps;
            pf_e = (i32)*pe;
        }
    };
    // File ids and eranks of this thread's slots: ALWAYS two loads (four in the waves that own a second vector), from a clamped
    // address when the thread's vector lies behind the tile -- the wait at the top of a tile counts them (they are the youngest
    // operations in flight there and are only needed when the tile's sums are done)
    v2u rf[2], rr[2];
    auto issue_regs = [&](u32 tile, u32 p0, u32 n_lds) {
        const u32 d4 = p0 & 3u, nv4 = (n_lds + d4 + 3u) >> 2;
        const v2u *g_f = reinterpret_cast<const v2u *>(files16 + (p0 - d4));
        const v2u *g_r = reinterpret_cast<const v2u *>(erank + (size_t)tile * IGD_TILE_BLOCK);
#pragma unroll
        for (u32 k = 0; k < 2; ++k) {
            if (k == 0 || (u32)wave < R1_WAVES) {
                const u32 q = min(threadIdx.x + k * SW_TPB, nv4 - 1u);
                rf[k] = g_f[q];
                rr[k] = g_r[q];
            }
        }
    };
    const bool two_vectors = (u32)wave < R1_WAVES;  // (uniform)
    auto item_queries = [&](u32 desc, u32 part, u32 &q_lo, u32 &q_hi) {
        q_lo = (u32)__builtin_amdgcn_readlane((int)desc, TABR_DESC);
        q_hi = (GTARS_IGD_ABLATE & 8) ? q_lo : (u32)__builtin_amdgcn_readlane((int)desc, TABR_DESC + 1);
        if (heavy_part) {
            q_lo = min(q_hi, q_lo + part * heavy_part);
            q_hi = min(q_hi, q_lo + heavy_part);
        }
    };
    // first index in [0, n) of an ascending LDS array with arr[i] >= x: bracket from the table, then four elements at once
    auto bracket = [&](const i32 *arr, const unsigned short *lut, i32 x, i32 k0, i32 k1, u32 sh, u32 n, u32 &a, u32 &b) {
        lut_range(lut, x, k0, k1, sh, n, a, b);
        while (b - a > 4u) {  // (2.25 records per bucket on average)
            const u32 m2 = a + ((b - a) >> 1);
            if (arr[m2] < x)
                a = m2 + 1;
            else
                b = m2;
        }
    };
    typedef i32 i4 __attribute__((ext_vector_type(4)));
    typedef i4 i4_a4 __attribute__((aligned(4)));
    typedef const __attribute__((address_space(3))) i4_a4 *lds_i4;
    auto below = [&](const i4 &y0, i32 x, u32 nb) -> u32 {  // how many of the bracket's nb <= 4 ascending elements are < x
        return (nb > 0 && y0.x < x ? 1u : 0u) + (nb > 1 && y0.y < x ? 1u : 0u) + (nb > 2 && y0.z < x ? 1u : 0u) + (nb > 3 && y0.w < x ? 1u : 0u);
    };
    u32 c_tile = 0, c_part = 0, c_desc = 0;
    if (blockIdx.x < n_items) {
        c_desc = request(blockIdx.x, c_tile, c_part);
        u32 q_lo, q_hi;
        item_queries(c_desc, c_part, q_lo, q_hi);
        const u32 p0 = (u32)__builtin_amdgcn_readlane((int)c_desc, 0), n_lds = (u32)__builtin_amdgcn_readlane((int)c_desc, 3);
        issue_dma(c_tile, p0, n_lds, q_lo, q_hi);
        issue_regs(c_tile, p0, n_lds);
    } else {
        rf[0] = rf[1] = rr[0] = rr[1] = v2u{0, 0};
    }
    __syncthreads();  // the bins are zero
    for (u32 item = blockIdx.x; item < n_items; item += gridDim.x) {
        const u32 desc = c_desc;
        auto word = [&](int k) -> u32 { return (u32)__builtin_amdgcn_readlane((int)desc, k); };
        const u32 p0 = word(0), n_lds = word(3), n_seg = word(10), sh_s = word(6), sh_e = word(9);
        const i32 S0 = (i32)word(4), S1 = (i32)word(5), E0 = (i32)word(7), E1 = (i32)word(8);
        u32 q_lo, q_hi;
        item_queries(desc, c_part, q_lo, q_hi);
        const bool has_next = item + gridDim.x < n_items;
        // this tile's arrays are in LDS (this wave's share) and its first queries in registers: everything but the youngest two /
        // four loads -- the file ids and eranks, which the end of the tile needs
        if (two_vectors)
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else
            asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        u32 n_tile = 0, n_part = 0, n_desc = 0;
        if (has_next) n_desc = request(item + gridDim.x, n_tile, n_part);
        u32 nq_lo = 0, nq_hi = 0, np0 = 0, nn_lds = 0;
        auto decode_next = [&]() {
            item_queries(n_desc, n_part, nq_lo, nq_hi);
            np0 = (u32)__builtin_amdgcn_readlane((int)n_desc, 0);
            nn_lds = (u32)__builtin_amdgcn_readlane((int)n_desc, 3);
        };
        const u32 d4 = p0 & 3u;
        const i32 *t_s = b_s + d4;
        const u32 nv4 = (n_lds + d4 + 3u) >> 2;
        STAMP(0);
        const bool asked = q_lo < q_hi;  // (uniform) does anybody ask about this tile?  if not, only the next one's loads are issued
        u32 sub = q_lo;
        do {  // rounds of RK_SUB queries: one for all but monstrous tiles
            const u32 sub_hi = min(q_hi, sub + RK_SUB);
            const bool last_sub = sub + RK_SUB >= q_hi;
            if (asked) {
            for (u32 i = threadIdx.x; i < (u32)(2 * RK_HW / 4); i += SW_TPB) reinterpret_cast<v4u *>(ha)[i] = v4u{0, 0, 0, 0};  // ha | ge
            lds_barrier();  // every wave's share of the tile is in LDS, the histograms are zero
            STAMP(1);
            for (u32 qb = sub; qb < sub_hi; qb += SW_TPB) {
                const u32 qi = qb + threadIdx.x;
                if (qi < sub_hi) {
                    i32 s, e;
                    if (qb == q_lo) {
                        s = interleaved ? pf_s : max(pf_s, 0);
                        e = pf_e;
                    } else if (interleaved) {
                        const uint2 se2 = reinterpret_cast<const uint2 *>(sqs)[qi];
                        s = (i32)se2.x;
                        e = (i32)se2.y;
                    } else {
                        s = max((i32)sqs[qi], 0);
                        e = (i32)sqe[qi];
                    }
                    // a = staged records that start before the query's end, k = ... that end at or before its start; the two
                    // searches side by side (their LDS round trips overlap)
                    u32 a, ab, k, kb;
                    bracket(t_s, lut_s, e, S0, S1, sh_s, n_lds, a, ab);
                    bracket(b_e, lut_e, s + 1, E0, E1, sh_e, n_lds, k, kb);
                    const i4 ya0 = *(lds_i4)(uintptr_t)(t_s + a);
                    const i4 yk0 = *(lds_i4)(uintptr_t)(b_e + k);
                    a += below(ya0, e, ab - a);
                    k += below(yk0, s + 1, kb - k);
                    const u32 as = a + d4;
                    atomicAdd(&ha[as >> 1], 1u << ((as & 1u) * 16u));
                    atomicAdd(&ge[k >> 1], 1u << ((k & 1u) * 16u));
                    if (a == n_lds && n_seg > n_lds) {
                        // the scan runs past the staged records: the rest from global memory, by this lane
                        for (u32 r = n_lds; r < n_seg; ++r) {
                            const i32 rs = v.starts[p0 + r];
                            if (rs >= e) break;
                            if (v.ends[p0 + r] > s) atomicAdd(&bins[v.files[p0 + r] & IGD_FILE_MASK], 1u);
                        }
                    }
                }
            }
            STAMP(2);
            lds_barrier();  // the histograms are complete; the starts, sorted ends and tables are dead
            STAMP(3);
            }
            // (the compiler's own wait for the file ids and eranks belongs HERE, in front of the transfers it does not count -- on
            // every path: placed at their use, behind the transfers, it would wait for those as well)
            asm volatile("" : "+v"(rf[0]), "+v"(rf[1]), "+v"(rr[0]), "+v"(rr[1]));
            if (last_sub && has_next) {
                decode_next();
                issue_dma(n_tile, np0, nn_lds, nq_lo, nq_hi);
            }
            if (asked) {
            // Inclusive prefix sums over the slots, in place: thread q owns the four slots of vector q (and of vector q + SW_TPB);
            // a wave's 256 slots are a CHUNK: sums inside the chunk now, the chunks' bases after the barrier.  (Everything goes
            // back to LDS in between: keeping both vectors' eight sums of both arrays in registers across the barrier spilled.)
#pragma unroll
            for (u32 k = 0; k < ((GTARS_IGD_ABLATE & 32) ? 0u : 2u); ++k) {
                const u32 q = threadIdx.x + k * SW_TPB;
                if (k == 0 || (u32)wave < R1_WAVES) {  // (uniform per wave: the second round is one or two waves)
                    v2u wa = v2u{0, 0}, we = v2u{0, 0};
                    if (q < nv4) {
                        wa = reinterpret_cast<const v2u *>(ha)[q];
                        we = reinterpret_cast<const v2u *>(ge)[q];
                    }
                    const u32 a0 = wa.x & 0xFFFFu, a1 = a0 + (wa.x >> 16), a2 = a1 + (wa.y & 0xFFFFu), a3 = a2 + (wa.y >> 16);
                    const u32 e0 = we.x & 0xFFFFu, e1 = e0 + (we.x >> 16), e2 = e1 + (we.y & 0xFFFFu), e3 = e2 + (we.y >> 16);
                    const u32 ta = wave_inclusive_scan_u32(a3, lane), te = wave_inclusive_scan_u32(e3, lane);
                    const u32 xa = ta - a3, xe = te - e3;  // what the lanes in front of this one hold
                    if (q < nv4) {
                        reinterpret_cast<v2u *>(ha)[q] = v2u{(a0 + xa) | ((a1 + xa) << 16), (a2 + xa) | ((a3 + xa) << 16)};
                        reinterpret_cast<v2u *>(ge)[q] = v2u{(e0 + xe) | ((e1 + xe) << 16), (e2 + xe) | ((e3 + xe) << 16)};
                    }
                    if (lane == 63) {
                        s_tot[0][k * NW + (u32)wave] = ta;
                        s_tot[1][k * NW + (u32)wave] = te;
                    }
                }
            }
            lds_barrier();
            STAMP(4);
            // the chunks' exclusive bases, every wave for itself: lanes 0..15 hold array 0's, lanes 16..31 array 1's (four DPP steps
            // inside a row of 16 lanes)
            u32 base;
            {
                const u32 x = lane < 32 ? s_tot[lane >> 4][lane & 15] : 0u;  // (totals of chunks that do not exist are zero)
                u32 inc = x;
                inc += (u32)__builtin_amdgcn_update_dpp(0, (int)inc, 0x111, 0xf, 0xf, false);  // row_shr:1
                inc += (u32)__builtin_amdgcn_update_dpp(0, (int)inc, 0x112, 0xf, 0xf, false);  // row_shr:2
                inc += (u32)__builtin_amdgcn_update_dpp(0, (int)inc, 0x114, 0xf, 0xf, false);  // row_shr:4
                inc += (u32)__builtin_amdgcn_update_dpp(0, (int)inc, 0x118, 0xf, 0xf, false);  // row_shr:8
                base = inc - x;
            }
            STAMP(5);
            const unsigned short *pe16 = reinterpret_cast<const unsigned short *>(ge);
#pragma unroll
            for (u32 k = 0; k < ((GTARS_IGD_ABLATE & 32) ? 0u : 2u); ++k) {
                const u32 q = threadIdx.x + k * SW_TPB;
                if (k == 0 || (u32)wave < R1_WAVES) {
                    const u32 base_a = (u32)__builtin_amdgcn_readlane((int)base, (int)(k * NW) + wave);
                    v2u wa = v2u{0, 0};
                    if (q < nv4) wa = reinterpret_cast<const v2u *>(ha)[q];
                    const u32 pa[4] = {wa.x & 0xFFFFu, wa.x >> 16, wa.y & 0xFFFFu, wa.y >> 16};
                    const u32 er[4] = {rr[k].x & 0xFFFFu, rr[k].x >> 16, rr[k].y & 0xFFFFu, rr[k].y >> 16};
                    const u32 fi[4] = {rf[k].x & 0xFFFFu, rf[k].x >> 16, rf[k].y & 0xFFFFu, rf[k].y >> 16};
                    u32 pe[4], be[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        pe[j] = pe16[er[j]];  // (er = 0 for slots without a record: a valid address)
                        be[j] = (u32)__builtin_amdgcn_ds_bpermute((int)(4u * (16u + (er[j] >> 8))), (int)base);
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const u32 i = 4u * q + (u32)j - d4;  // staged record of this slot (wraps for the slots in front of the tile)
                        // hits(r) = #{q : e_q <= erank(r)} - #{q : a_q <= r}
                        const u32 c = (pe[j] + be[j]) - (pa[j] + base_a);
                        if (q < nv4 && i < n_lds && c) atomicAdd(&bins[fi[j]], c);
                    }
                }
            }
            }
            if (last_sub && has_next) issue_regs(n_tile, np0, nn_lds);
            if (asked) lds_barrier();  // the histograms may be overwritten
            STAMP(6);
            sub += RK_SUB;
        } while (sub < q_hi);
        c_desc = n_desc;
        c_tile = n_tile;
        c_part = n_part;
    }
#if IGD_STAMPS
    if (threadIdx.x == 0) {
        for (int k = 0; k < 7; ++k) atomicAdd(&g_sweep_stamps[k], st_acc[k]);
        atomicAdd(&g_sweep_stamps[7], 1ull);
    }
#endif
    __syncthreads();
    for (u32 i = threadIdx.x; i < n_bins; i += SW_TPB) {
        const u32 b = bins[i];
        if (b) atomicAdd(&hits[i], (unsigned long long)b);
    }
}

// ---- pme_file: per record, the largest end among the EARLIER records of the same file on the same chromosome ----
// Built once per database on the device (first binary count): a stable radix sort of the record positions by file id
// groups every (file, chromosome) run in stored order, a segmented exclusive prefix maximum runs over the ends in that
// order (three phases: wave aggregates, their scan, apply), and the result is scattered back to stored order.
constexpr int SG_ITEMS = 8;             // consecutive elements per lane
constexpr int SG_WAVE = 64 * SG_ITEMS;  // elements per wave
constexpr int SG_TPB = 256;

struct SegAgg {
    u32 f;  // a segment head lies in the range
    i32 v;  // maximum since the last head (of the whole range if there is none)
};
__device__ __forceinline__ SegAgg seg_combine(SegAgg a, SegAgg b) {  // a then b
    SegAgg r;
    r.f = a.f | b.f;
    r.v = b.f ? b.v : max(a.v, b.v);
    return r;
}
__device__ __forceinline__ SegAgg seg_wave_inclusive(SegAgg x, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        SegAgg y;
        y.f = __shfl_up(x.f, d, 64);
        y.v = __shfl_up(x.v, d, 64);
        if (lane >= d) x = seg_combine(y, x);
    }
    return x;
}

// sorted (file, position) pairs -> per element: segment-head flag and value (the record's end)
__device__ __forceinline__ u32 chrom_of(const u32 *__restrict__ chrom_off, u32 n_chrom, u32 r) {
    u32 lo = 0, hi = n_chrom;  // last c with chrom_off[c] <= r
    while (lo + 1 < hi) {
        const u32 mid = (lo + hi) >> 1;
        if (chrom_off[mid] <= r)
            lo = mid;
        else
            hi = mid;
    }
    return lo;
}

// PHASE 0: wave aggregates only; PHASE 1: exclusive values written through `pos` to stored order
template <int PHASE>
__global__ void __launch_bounds__(SG_TPB)
k_pme_scan(const u32 *__restrict__ file_sorted, const u32 *__restrict__ pos, const i32 *__restrict__ ends,
           const u32 *__restrict__ chrom_off, u32 n_chrom, u32 n, SegAgg *__restrict__ wave_agg, i32 *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const u32 w = (blockIdx.x * SG_TPB + threadIdx.x) >> 6;
    const u32 base = w * SG_WAVE + (u32)lane * SG_ITEMS;
    if (w * SG_WAVE >= n) return;
    u32 head[SG_ITEMS], p[SG_ITEMS];
    i32 val[SG_ITEMS];
    u32 prev_f = 0xFFFFFFFFu, prev_c = 0xFFFFFFFFu;
    if (base > 0 && base < n) {
        prev_f = file_sorted[base - 1];
        prev_c = chrom_of(chrom_off, n_chrom, pos[base - 1]);
    }
#pragma unroll
    for (int k = 0; k < SG_ITEMS; ++k) {
        const u32 i = base + k;
        if (i < n) {
            const u32 f = file_sorted[i];
            p[k] = pos[i];
            const u32 c = chrom_of(chrom_off, n_chrom, p[k]);
            head[k] = (i == 0 || f != prev_f || c != prev_c) ? 1u : 0u;
            val[k] = ends[p[k]];
            prev_f = f;
            prev_c = c;
        } else {
            head[k] = 1u;
            val[k] = 0;
            p[k] = 0xFFFFFFFFu;
        }
    }
    SegAgg mine{0u, 0};
    i32 ex[SG_ITEMS];
    u32 headed[SG_ITEMS];  // a head at or before item k inside this lane
#pragma unroll
    for (int k = 0; k < SG_ITEMS; ++k) {
        if (head[k]) {
            mine.f = 1u;
            mine.v = 0;
        }
        headed[k] = mine.f;
        ex[k] = mine.v;
        mine.v = max(mine.v, val[k]);
    }
    const SegAgg inc = seg_wave_inclusive(mine, lane);
    if (PHASE == 0) {
        if (lane == 63) wave_agg[w] = inc;
        return;
    }
    SegAgg carry;  // everything before this lane: the waves before (wave_agg holds their EXCLUSIVE scan) and the lanes before
    carry.f = __shfl_up(inc.f, 1, 64);
    carry.v = __shfl_up(inc.v, 1, 64);
    if (lane == 0) carry = SegAgg{0u, 0};
    carry = seg_combine(wave_agg[w], carry);
#pragma unroll
    for (int k = 0; k < SG_ITEMS; ++k)
        if (p[k] != 0xFFFFFFFFu) out[p[k]] = headed[k] ? ex[k] : max(carry.v, ex[k]);
}

// exclusive scan of the wave aggregates, in place (one workgroup; a few thousand entries per round)
__global__ void __launch_bounds__(1024)
k_pme_scan_aggs(SegAgg *__restrict__ agg, u32 n_waves) {
    __shared__ SegAgg s_w[16];
    __shared__ SegAgg s_carry;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_carry = SegAgg{0u, 0};
    __syncthreads();
    for (u32 base = 0; base < n_waves; base += 1024) {
        const u32 i = base + threadIdx.x;
        const SegAgg x = i < n_waves ? agg[i] : SegAgg{0u, 0};
        const SegAgg inc = seg_wave_inclusive(x, lane);
        if (lane == 63) s_w[wave] = inc;
        __syncthreads();
        SegAgg before = s_carry;
        for (int k = 0; k < wave; ++k) before = seg_combine(before, s_w[k]);
        SegAgg ex;
        ex.f = __shfl_up(inc.f, 1, 64);
        ex.v = __shfl_up(inc.v, 1, 64);
        if (lane == 0) ex = SegAgg{0u, 0};
        ex = seg_combine(before, ex);
        if (i < n_waves) agg[i] = ex;
        __syncthreads();
        if (threadIdx.x == 1023) s_carry = seg_combine(ex, x);
        __syncthreads();
    }
}

__global__ void k_file_keys(const u32 *__restrict__ files, u32 *__restrict__ key, u32 n) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) key[i] = files[i] & IGD_FILE_MASK;
}

__global__ void k_iota_u32(u32 *__restrict__ p, u32 n) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = i;
}

size_t igd_pme_ws_bytes(u32 n) {
    const u32 n_waves = (n + SG_WAVE - 1) / SG_WAVE;
    return (size_t)n * 4 * 4 + (size_t)n_waves * sizeof(SegAgg) + radix_sort_ws_bytes(n) + 256;
}

gtars_status igd_build_pme_file(const IgdView &v, i32 *pme, void *ws, size_t ws_bytes, hipStream_t st) {
    const u32 n = v.n;
    if (!n) return GTARS_OK;
    if (ws_bytes < igd_pme_ws_bytes(n)) return fail(GTARS_ERR_INTERNAL, "pme_file workspace too small");
    u32 *k0 = (u32 *)ws, *v0 = k0 + n, *k1 = v0 + n, *v1 = k1 + n;
    const u32 n_waves = (n + SG_WAVE - 1) / SG_WAVE;
    SegAgg *agg = (SegAgg *)(v1 + n);
    void *sort_ws = (void *)(((uintptr_t)(agg + n_waves) + 63) & ~(uintptr_t)63);
    const size_t sort_bytes = ws_bytes - (size_t)((char *)sort_ws - (char *)ws);
    hipLaunchKernelGGL(k_file_keys, dim3((n + 255) / 256), dim3(256), 0, st, v.files, k0, n);  // (without a pieces view's flag bit)
    hipLaunchKernelGGL(k_iota_u32, dim3((n + 255) / 256), dim3(256), 0, st, v0, n);
    int bits = 1;
    while (bits < 32 && (1ull << bits) < (u64)v.n_files) ++bits;
    bits = (bits + 7) & ~7;
    int res = 0;
    gtars_status s = radix_sort_pairs(k0, v0, k1, v1, n, 0, bits, sort_ws, sort_bytes, &res, st);
    if (s) return s;
    const u32 *fs = res ? k1 : k0, *ps = res ? v1 : v0;
    const unsigned grid = (unsigned)(((u64)n_waves * 64 + SG_TPB - 1) / SG_TPB);
    hipLaunchKernelGGL(k_pme_scan<0>, dim3(grid), dim3(SG_TPB), 0, st, fs, ps, v.ends, v.chrom_off, v.n_chrom, n, agg, pme);
    hipLaunchKernelGGL(k_pme_scan_aggs, dim3(1), dim3(1024), 0, st, agg, n_waves);
    hipLaunchKernelGGL(k_pme_scan<1>, dim3(grid), dim3(SG_TPB), 0, st, fs, ps, v.ends, v.chrom_off, v.n_chrom, n, agg, pme);
    GT_HIP(hipGetLastError());
    return GTARS_OK;
}

// ---- launcher ------------------------------------------------------------------------------------

bool igd_sweep_supported(const IgdView &v, u64 nq) {
    if (cfg_get("GTARS_NO_IGD_SWEEP")) return false;
    // The sweep reads the whole database once per batch (0.37 ms per 5e7 records) whatever the batch size; the per-query
    // kernel costs ~1.2 ns per query at config-3/4 densities.  Crossover measured on config 4 (5e7 records): ~300k
    // queries -- the 1e5-region user set goes per query (0.17 instead of 0.37 ms), the 1e6-region universe sweeps.
    const u64 min_q = cfg_get("GTARS_IGD_SWEEP_MIN") ? (u64)atoll(cfg_get("GTARS_IGD_SWEEP_MIN"))
                                                    : std::max<u64>(1u << 16, (u64)v.n / 256);
    // u32 LDS bins: a workgroup adds at most (its queries x hits) -- keep the batch below 2^31 queries
    return v.n > 0 && v.n_files > 0 && v.n_files <= 16384 && nq >= min_q && nq < (1ull << 31);
}

constexpr u32 HEAVY_PART_MIN = 4096;  // queries per part of a heavy tile, at least (a multiple of SW_TPB)
size_t igd_sweep_ws_bytes(u64 nq, u32 n_tiles, u32 n_chrom) {
    // kc ks ke | sorted qs qe chrom | perm (or owner tiles) | ql qh | cq_off | bin offsets | slack | heavy-tile parts |
    // partition / sort scratch
    return 64 + (size_t)nq * 4 * 7 + (size_t)n_tiles * (8 * IGD_MAX_SETS + 4) + ((size_t)n_chrom + 2) * 4 * IGD_MAX_SETS + 512 +
           ((size_t)nq / HEAVY_PART_MIN + 16) * 8 +
           std::max(device_sort_perm_ws_bytes((u32)nq), multisplit_ws_bytes(n_tiles + 1, (u32)nq));
}

// per-tile maximum end (index build: the carry-in of the sweep's prefix maximum is its running maximum)
__global__ void k_igd_tile_max_end(const i32 *__restrict__ ends, const u32 *__restrict__ tile_first,
                                   const u32 *__restrict__ tile_cnt, u32 n_tiles, i32 *__restrict__ tile_max) {
    __shared__ i32 s_m[4];
    const u32 t = blockIdx.x;
    if (t >= n_tiles) return;
    const u32 p0 = tile_first[t], cnt = tile_cnt[t];
    i32 m = 0;
    for (u32 i = threadIdx.x; i < cnt; i += blockDim.x) m = max(m, ends[p0 + i]);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = max(m, __shfl_xor(m, d, 64));
    if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) tile_max[t] = max(max(s_m[0], s_m[1]), max(s_m[2], s_m[3]));
}

gtars_status launch_igd_tile_max_end(const i32 *ends, const u32 *tile_first, const u32 *tile_cnt, u32 n_tiles, i32 *tile_max,
                                     hipStream_t st) {
    if (!n_tiles) return GTARS_OK;
    hipLaunchKernelGGL(k_igd_tile_max_end, dim3(n_tiles), dim3(256), 0, st, ends, tile_first, tile_cnt, n_tiles, tile_max);
    GT_HIP(hipGetLastError());
    return GTARS_OK;
}

// The routing kernel can serve this database -- with the fine tables in LDS or, for larger databases, the tile bounds in global
// memory -- up to 65534 tiles = 134M records (16-bit owner keys): the batch is then PARTITIONED by owner tile (or taken as it
// lies when it is in order).  Beyond that, and for GTARS_IGD_FULL_SORT=1 (tests / A-B runs), the batch is fully sorted by
// (chromosome, start).  (Round 2's two-kernel preparation in front of the partition is gone: no database size selected it.)
static bool igd_route_fits(const IgdView &v, const IgdTiles &tl, bool *fine) {
    if (!tl.route_lut || !tl.bnd || cfg_flag("GTARS_IGD_FULL_SORT") || tl.n_tiles + 1 > MS_MAX_BINS_2L) return false;
    const size_t limit = 160 * 1024 - 64;
    const bool with_fine = tl.route_flut && igd_route_fine_lds_bytes(tl.n_tiles, v.n_chrom, tl.route_fn) <= limit &&
                           !cfg_flag("GTARS_IGD_ROUTE_BND_GLOBAL");
    if (fine) *fine = with_fine;
    return with_fine || igd_route_lds_bytes(tl.n_tiles, v.n_chrom, tl.route_n) <= limit;
}

// several query sets in one sweep: only through the partition (the set of a query travels with its pair), at most 4 sets (two
// tag bits), one row of LDS counters per set
bool igd_sweep_sets_supported(const IgdView &v, const IgdTiles &tl, u64 nq, u32 n_sets) {
    return n_sets >= 1 && n_sets <= 4 && (u64)n_sets * v.n_files <= 16384 && igd_sweep_supported(v, nq) && igd_route_fits(v, tl, nullptr);
}

gtars_status launch_igd_sweep(const IgdView &v, const IgdTiles &tl, const u32 *qc, const u32 *qs, const u32 *qe, u64 nq64,
                              i32 min_overlap, int binary, u64 *hits, void *ws, size_t ws_bytes, hipStream_t st, u32 call_no, u32 n_sets,
                              const u32 *set_bounds) {
    const u32 nq = (u32)nq64;
    const u32 n_tiles = tl.n_tiles;
    if (n_sets > 1 && (!set_bounds || !igd_sweep_sets_supported(v, tl, nq64, n_sets)))
        return fail(GTARS_ERR_INTERNAL, "IGD sweep: this batch of query sets needs the per-set path");
    if (n_sets <= 1) set_bounds = nullptr;
    const u32 n_bins = v.n_files * std::max<u32>(n_sets, 1);
    if (ws_bytes < igd_sweep_ws_bytes(nq, n_tiles, v.n_chrom)) return fail(GTARS_ERR_INTERNAL, "IGD sweep workspace too small");
    // the first 16 words of the workspace outlive the call: the two order flags (see k_igd_call_init) and the heavy-tile count
    u32 *persist = (u32 *)ws;
    u32 *kc = persist + 16, *ks = kc + nq, *ke = ks + nq;
    u32 *sc = ke + nq, *ss = sc + nq, *se = ss + nq, *perm = se + nq;  // perm doubles as the owner-tile column
    u32 *ql = perm + nq, *qh = ql + (size_t)IGD_MAX_SETS * n_tiles, *cq_off = qh + (size_t)IGD_MAX_SETS * n_tiles;  // (a row per set)
    u32 *bin_off = cq_off + (size_t)IGD_MAX_SETS * (v.n_chrom + 2);  // [n_tiles + 2]
    u32 *d_unsorted = persist + (call_no & 1u), *d_next_flag = persist + ((call_no + 1u) & 1u);  // the "not in owner order" flag
    u32 *after_off = bin_off + n_tiles + 2;
    // tiles that own far more queries than the average are served in parts of heavy_part queries (k_igd_sweep's work items)
    HeavyBins heavy;
    heavy.list = reinterpret_cast<uint2 *>(((uintptr_t)(after_off + 16) + 7) & ~(uintptr_t)7);
    heavy.count = persist + 2;  // number of listed heavy-tile parts
    heavy.part = std::max<u32>(HEAVY_PART_MIN, (u32)std::min<u64>(8ull * (nq / std::max<u32>(n_tiles, 1)), 1u << 30) / SW_TPB * SW_TPB);
    heavy.n_real_bins = n_tiles;
    heavy.cap = nq / heavy.part + 1;
    if (cfg_flag("GTARS_IGD_NO_HEAVY_PARTS")) heavy.part = 0;  // tests / A-B runs
    void *scratch = (void *)(((uintptr_t)(heavy.list + (size_t)nq / HEAVY_PART_MIN + 8) + 63) & ~(uintptr_t)63);
    const size_t scratch_bytes = ws_bytes - (size_t)((char *)scratch - (char *)ws);
    (void)sc;
    // GTARS_IGD_ALWAYS_SORT (tests): start from "not in order" -- the order check only ever raises the flag.  Several sets: always
    // partitioned (a concatenation of sets is not in order, and the partition is what tags the pairs)
    const bool always_sort = cfg_flag("GTARS_IGD_ALWAYS_SORT");
    // Several sets that are each in (chromosome, start) order -- a LOLA call's universe and user sets -- are swept as they arrive, a
    // row of tile ranges per set (round 5; the walked min_overlap >= 1 forms without the credited-file list); otherwise, and always
    // for the remaining forms, a batch of several sets is partitioned (the partition is what tags the pairs).
    const int mode_ = !binary ? 0 : (min_overlap == 1 && tl.pme_file ? 2 : 1);
    const bool sets_in_order_ok = set_bounds && mode_ != 1 && !v.pieces && !cfg_flag("GTARS_IGD_SETS_ALWAYS_PARTITION");
    SetRows rows;
    rows.nq = nq;
    if (sets_in_order_ok) {
        rows.n = n_sets;
        rows.b1 = set_bounds[0], rows.b2 = set_bounds[1], rows.b3 = set_bounds[2];
    }
    const u32 n_sets_io = sets_in_order_ok ? n_sets : 1u;
    const u32 flag0 = always_sort || (set_bounds && !sets_in_order_ok) ? 1u : 0u;
    bool fine = false;
    const bool routed = igd_route_fits(v, tl, &fine);
    int dev = 0, cus = 256;
    GT_HIP(hipGetDevice(&dev));
    GT_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const u32 *t_ql = ql, *t_qh = qh;
    int interleaved = 0;
    const u32 *part_flag = nullptr, *part_ab = nullptr, *part_ql = nullptr;
    static std::mutex attr_mu;
    if (routed) {
        // ---- the partition path: begin (zeroing + order check) | route (or tile ranges) | split A | split B ----
        if (scratch_bytes < multisplit_ws_bytes(n_tiles + 1, nq)) return fail(GTARS_ERR_INTERNAL, "IGD sweep workspace too small");
        // The routing kernel's grid: a two-level split (large batches, many tiles) takes the workgroups' counter ROWS, so the grid
        // is this kernel's own to choose -- every CU gets a chunk once there are 4096 queries per workgroup (a 1.1M-query batch
        // ran on 67 CUs: 35 us), and never more than 65532 queries per workgroup (16-bit counters: batches beyond 16.7M queries
        // take more workgroups than CUs, which queue); a one-level split wants one table row per workgroup of ITS grid.
        u32 *d_tot = multisplit_totals(scratch, n_tiles + 1, nq);  // null: one-level split
        u32 rt_wg = multisplit_workgroups(nq), rt_chunk = multisplit_chunk(nq);  // (a multiple of the 4 queries a lane takes per step)
        if (d_tot) {
            rt_wg = std::max<u32>(1, std::min<u32>((u32)cus, (nq + 4095) / 4096));
            rt_wg = std::max<u32>(rt_wg, (nq + 65531u) / 65532u);
            // (test hook: a small cap sends a small batch through the many-workgroup form -- more rows than CUs)
            const u32 cap = (u32)std::min<long>(65532, std::max<long>(64, cfg_int("GTARS_IGD_ROUTE_CHUNK_MAX", 65532)));
            rt_wg = std::max<u32>(rt_wg, std::min<u32>(512u, (nq + cap - 1) / cap));  // (<= 512 rows: multisplit_ws_bytes)
            rt_chunk = ((nq + rt_wg - 1) / rt_wg + 3u) & ~3u;
            // whole steps of 4096 queries when the chunks are long: the partial last step of a chunk runs through the general loop
            // without a prefetch and cost 7.1k cycles against 3.9k for a full step (in-kernel stamps, 10M queries: nine full steps and
            // 54 % of a tenth on 256 workgroups -> ten full steps on 245)
            const u32 whole = (rt_chunk + 4095u) & ~4095u;
            if (rt_chunk >= 8u * 4096u && whole <= 61440u && !cfg_flag("GTARS_IGD_ROUTE_NO_WHOLE_STEPS")) {
                rt_chunk = whole;
                rt_wg = (nq + rt_chunk - 1) / rt_chunk;
            }
        }
        if (rt_chunk > 65535u) return fail(GTARS_ERR_INTERNAL, "IGD sweep: routing chunk exceeds the 16-bit counters");  // (cannot happen)
        {
            static bool done[64] = {};
            std::lock_guard<std::mutex> lock(attr_mu);
            if (dev >= 0 && dev < 64 && !done[dev]) {
                const void *fns[] = {reinterpret_cast<const void *>(k_igd_route<true, true>), reinterpret_cast<const void *>(k_igd_route<false, true>),
                                     reinterpret_cast<const void *>(k_igd_route<true, false>), reinterpret_cast<const void *>(k_igd_route<false, false>)};
                for (const void *fn : fns) GT_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                done[dev] = true;
            }
        }
        const size_t rt_lds = fine ? igd_route_fine_lds_bytes(n_tiles, v.n_chrom, tl.route_fn) : igd_route_lds_bytes(n_tiles, v.n_chrom, tl.route_n);
        const u32 n_tot0 = d_tot ? (u32)multisplit_zeroed_words(n_tiles + 1) : 0u, n_init = std::max<u32>(std::max<u32>(n_bins, n_tot0), 1u);
        const bool vec = ((((uintptr_t)qc) | ((uintptr_t)qs) | ((uintptr_t)qe)) & 15u) == 0 && (((uintptr_t)perm) & 7u) == 0;
        {
            // result vector, totals / cursors, flags -- and, unless the partition is certain, the order check -- in ONE launch
            ProfScope p("k_igd_begin", st);
            // (every workgroup reads at least one 12-KB step of the batch before it can see disorder: cus x 4 of them keep 12 MB in
            // flight for a batch that IS in order and cost a shuffled one 12 MB, not 25)
            const u64 og_check = std::min<u64>((u64)cus * 4, ((u64)nq + ORD_TPB * 4 - 1) / (ORD_TPB * 4));
            const unsigned og = (unsigned)std::max<u64>(1, flag0 ? (n_init + ORD_TPB - 1) / ORD_TPB : std::max<u64>(og_check, std::min<u64>(64, (n_init + ORD_TPB - 1) / ORD_TPB)));
            auto begin = n_sets_io > 1 ? (vec ? k_igd_begin<true, true> : k_igd_begin<false, true>) : (vec ? k_igd_begin<true, false> : k_igd_begin<false, false>);
            hipLaunchKernelGGL(begin, dim3(og), dim3(ORD_TPB), 0, st, (unsigned long long *)hits, n_bins,
                               d_tot, n_tot0, d_unsorted, d_next_flag, heavy.count, flag0, qc, qs, qe, nq, v.n_chrom, cq_off, rows);
        }
        {
            ProfScope p("k_igd_route", st);
            auto route = fine ? (vec ? k_igd_route<true, true> : k_igd_route<false, true>) : (vec ? k_igd_route<true, false> : k_igd_route<false, false>);
            hipLaunchKernelGGL(route, dim3(rt_wg), dim3(RT_TPB), rt_lds, st, qc, qs, qe, nq, v.n_chrom, tl.bnd, tl.chrom_tile_off,
                               fine ? tl.route_fbase : tl.route_base, tl.route_len, fine ? tl.route_flut : tl.route_lut,
                               fine ? tl.route_fn : tl.route_n, fine ? tl.route_fshift : tl.route_shift, n_tiles, rt_chunk,
                               reinterpret_cast<unsigned short *>(perm), multisplit_table(scratch), d_tot, d_unsorted,
                               d_tot ? multisplit_coarse_totals(scratch, n_tiles + 1, nq) : (u32 *)nullptr, multisplit_coarse_shift(n_tiles + 1),
                               tl.route_kq, v, tl.first, tl.cnt, tl.chrom, (const u32 *)cq_off, flag0 ? (u32 *)nullptr : ql, qh, heavy, n_sets_io);
        }
        // No host round trip: both continuations are enqueued and the flag the order check leaves on the device picks one -- the
        // partition kernels return at once for a batch that is already in owner order (the routing launch has computed its tile
        // ranges instead), and the sweep takes its inputs accordingly.
        // K1 (multisplit): (start, end) pairs grouped by owner tile; bin_off[t], bin_off[t + 1] bound tile t's queries
        gtars_status s1 = multisplit_pairs(reinterpret_cast<const unsigned short *>(perm), qs, qe, nq, n_tiles + 1, n_tiles,
                                           reinterpret_cast<uint2 *>(ss), bin_off, scratch, scratch_bytes, st, d_unsorted, set_bounds, &heavy,
                                           rt_wg);  // ss, se adjacent: 2 * nq words
        if (s1) return s1;
        part_flag = d_unsorted;
        part_ab = ss;
        part_ql = bin_off;
        // the in-order continuation reads the batch where it lies: the caller's raw columns (clamped on the way)
        ss = const_cast<u32 *>(qs);
        se = const_cast<u32 *>(qe);
    } else {
        // ---- the full-sort path (databases beyond 134M records; GTARS_IGD_FULL_SORT): prepared columns, radix sort ----
        hipLaunchKernelGGL(k_igd_call_init, dim3((std::max<u32>(n_bins, 1u) + 255) / 256), dim3(256), 0, st, (unsigned long long *)hits, n_bins,
                           d_unsorted, flag0, (u32 *)nullptr, 0u, qc, qs, qe, 0u, v.n_chrom, d_next_flag, heavy.count);
        {
            ProfScope p("k_igd_prep_queries", st);
            const u32 n_wg = std::max<u32>(1, std::min<u32>((u32)cus, (nq + 4095) / 4096));
            const u32 chunk = ((nq + n_wg - 1) / n_wg + PREP_TPB - 1) / PREP_TPB * PREP_TPB;
            hipLaunchKernelGGL(k_igd_prep_queries, dim3(n_wg), dim3(PREP_TPB), 0, st, qc, qs, qe, nq, v.n_chrom, chunk, kc, ks, ke, d_unsorted);
        }
        u32 h_unsorted = 1;
        if (!always_sort) {
            GT_HIP(hipMemcpyAsync(&h_unsorted, d_unsorted, sizeof(u32), hipMemcpyDeviceToHost, st));
            GT_HIP(hipStreamSynchronize(st));
        }
        if (h_unsorted) {
            // K1 (radix sort): order the queries by (chromosome, start)
            gtars_status s1 = device_sort_perm_ws(kc, ks, nullptr, nq, v.n_chrom + 1, perm, scratch, scratch_bytes, st);
            if (s1) return s1;
            ProfScope p("k_gather2_u32", st);
            hipLaunchKernelGGL(k_gather2_u32, dim3((nq + 255) / 256), dim3(256), 0, st, ks, ke, perm, nq, ss, se);
        } else {
            perm = nullptr;  // the batch is in (chromosome, start) order already
            ss = ks;
            se = ke;
        }
        ProfScope p("k_igd_tile_ranges", st);
        hipLaunchKernelGGL(k_igd_chrom_segments<false>, dim3((v.n_chrom + 1 + 63) / 64), dim3(64), 0, st, kc, (const u32 *)nullptr,
                           (const u32 *)nullptr, perm, nq, v.n_chrom, cq_off, (const u32 *)nullptr);
        hipLaunchKernelGGL(k_igd_tile_ranges<false>, dim3((n_tiles + 255) / 256), dim3(256), 0, st, v, tl.first, tl.cnt, tl.chrom, n_tiles, ss,
                           cq_off, ql, qh, (const u32 *)nullptr, heavy);
    }
    const int mode = !binary ? 0 : (min_overlap == 1 && tl.pme_file ? 2 : 1);
    if (!tl.pm || !tl.files16 || !tl.tab) return fail(GTARS_ERR_INTERNAL, "IGD sweep: the per-tile tables were not built");
    // starts | {end, file} pairs (mode 2: {end, pme_file} pairs + u16 file ids) | counters (u32, or 16-bit two per word: mode 2
    // when that admits another workgroup per CU)
    const size_t lds_rec = (size_t)(IGD_TILE + IGD_HALO + 4) * (mode == 2 ? 14 : 12);
    size_t lds = lds_rec + (((size_t)n_bins + 1) & ~(size_t)1) * 4;
    const size_t lds16 = lds_rec + ((size_t)n_bins + 3) / 4 * 8;
    const bool mo1 = min_overlap == 1;
    auto kern = mode == 2 ? k_igd_sweep<2, true> : mode == 1 ? k_igd_sweep<1, false> : mo1 ? k_igd_sweep<0, true> : k_igd_sweep<0, false>;
    const size_t lds_cu = 160 * 1024;  // what a CU has; the static part (search tables) is ~2 KB per workgroup
    const size_t lds_static = (size_t)TAB_LUT_WORDS * 4 + 256;
    const bool b16 = mode == 2 && !cfg_flag("GTARS_IGD_NO_B16") &&
                     (cfg_flag("GTARS_IGD_FORCE_B16") ||  // (test hook)
                      (lds_cu / (lds16 + lds_static) > lds_cu / (lds + lds_static) && lds_cu / (lds + lds_static) < 4));
    if (b16) {
        kern = k_igd_sweep<2, true, true>;
        lds = lds16;
    }
    if (n_sets_io > 1)  // (several sets served in order: the instantiations with the set bookkeeping; not for pieces views -- below)
        kern = mode == 2 ? (b16 ? k_igd_sweep<2, true, true, false, true> : k_igd_sweep<2, true, false, false, true>)
                         : mo1 ? k_igd_sweep<0, true, false, false, true> : k_igd_sweep<0, false, false, false, true>;
    if (v.pieces) {
        if (mode == 1 || !mo1) return fail(GTARS_ERR_INTERNAL, "IGD sweep: a pieces view serves min_overlap == 1 only");
        kern = mode == 2 ? (b16 ? k_igd_sweep<2, true, true, true> : k_igd_sweep<2, true, false, true>) : k_igd_sweep<0, true, false, true>;
    }
    const bool rank_form = mode == 0 && mo1 && !v.pieces && !set_bounds && tl.ends_sorted && tl.erank && tl.tab_r &&
                           !cfg_flag("GTARS_IGD_NO_RANK");
    if (rank_form) {
        // pairwise counts, min_overlap == 1: the rank-histogram form (no candidate walk)
        const size_t lds_r = igd_sweep_rank_lds_bytes(n_bins);
        {
            static std::mutex mu;
            static bool done[64] = {};
            std::lock_guard<std::mutex> lock(mu);
            if (dev >= 0 && dev < 64 && !done[dev]) {
                GT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_igd_sweep_rank), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)igd_sweep_rank_lds_bytes(16384)));
                done[dev] = true;
            }
        }
        int per_cu = 1;
        GT_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_igd_sweep_rank, SW_TPB, lds_r));
        if (per_cu < 1) return fail(GTARS_ERR_INTERNAL, "k_igd_sweep_rank does not fit on a CU");
        const unsigned grid = (unsigned)std::min<u64>((u64)cus * per_cu, n_tiles);
        {
            ProfScope p("k_igd_sweep<pairwise>", st);
            hipLaunchKernelGGL(k_igd_sweep_rank, dim3(grid), dim3(SW_TPB), lds_r, st, v, tl.files16, tl.ends_sorted, tl.erank, tl.tab_r, n_tiles,
                               ss, se, interleaved, t_ql, t_qh, (unsigned long long *)hits, part_flag, part_ab, part_ql, n_bins,
                               (const uint2 *)heavy.list, (const u32 *)heavy.count, heavy.part, heavy.cap);
        }
        GT_HIP(hipGetLastError());
        prof_note_fact("igd_sweep_rank_form");
        if (routed) prof_note_device_flag("igd_batch_partitioned", "igd_batch_in_owner_order", d_unsorted, st);
        return GTARS_OK;
    }
    {
        // the dynamic-LDS limit belongs to the function, not to the calling thread: raised once per device to the largest
        // size any launch can ask for (5 staged arrays + 16384 file bins) and never lowered
        static std::mutex mu;
        static bool done[20][64] = {};
        int dev = 0;
        GT_HIP(hipGetDevice(&dev));
        const int slot = (b16 ? 4 : mode == 0 && !mo1 ? 3 : mode) + (v.pieces ? 5 : 0) + (n_sets_io > 1 ? 10 : 0);
        std::lock_guard<std::mutex> lock(mu);
        if (dev >= 0 && dev < 64 && !done[slot][dev]) {
            GT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)(((size_t)(IGD_TILE + IGD_HALO + 4) * 4 + 16384 + 2048) * 4)));
            done[slot][dev] = true;
        }
    }
    int per_cu = 1;
    GT_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, SW_TPB, lds));
    if (per_cu < 1) return fail(GTARS_ERR_INTERNAL, "k_igd_sweep does not fit on a CU");
    const unsigned grid = (unsigned)std::min<u64>((u64)cus * per_cu, n_tiles);
    {
        ProfScope p(binary ? "k_igd_sweep<binary>" : "k_igd_sweep<pairwise>", st);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(SW_TPB), lds, st, v, tl.pme_file, tl.pm, tl.files16, tl.tab, n_tiles, ss, se,
                           interleaved, t_ql, t_qh, min_overlap, (unsigned long long *)hits, part_flag, part_ab, part_ql, n_bins,
                           (const uint2 *)heavy.list, (const u32 *)heavy.count, heavy.part, heavy.cap, n_sets_io);
    }
    GT_HIP(hipGetLastError());
    // which continuation the device took (profiling mode: a deterministic fact for the tests, not a timing)
    if (routed) prof_note_device_flag("igd_batch_partitioned", "igd_batch_in_owner_order", d_unsorted, st);
    return GTARS_OK;
}

}  // namespace gtars

#if IGD_STAMPS
extern "C" int gtars_debug_route_stamps(unsigned long long *out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(gtars::g_route_stamps), 64) != hipSuccess) return 1;
    if (reset) {
        unsigned long long z[8] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(gtars::g_route_stamps), z, 64) != hipSuccess) return 1;
    }
    return 0;
}
extern "C" int gtars_debug_sweep_stamps(unsigned long long *out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(gtars::g_sweep_stamps), 64) != hipSuccess) return 1;
    if (reset) {
        unsigned long long z[8] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(gtars::g_sweep_stamps), z, 64) != hipSuccess) return 1;
    }
    return 0;
}
#endif
