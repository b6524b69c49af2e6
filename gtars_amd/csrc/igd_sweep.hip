// igd_sweep.hip -- K5, batch form: Igd::count_set_overlaps / count_region_hits
// (gtars-igd/src/igd.rs:544-590) for a large query batch as ONE streaming pass over the database.
//
// The per-query kernel in kernels.hip binary-searches a 5e7-record array for every query (random HBM
// accesses).  Here the roles are swapped: the QUERIES are sorted once by (chromosome, start) with the
// device radix sort (K1), the database -- already sorted by (chromosome, start) -- is cut into tiles of
// IGD_TILE records, and each workgroup stages one tile in LDS (coalesced 16-byte loads: the database is
// read from HBM once, plus a 256-record halo) and serves the queries it OWNS: a query belongs to the
// tile that holds its lower_bound position (first record with start >= q.start - max_len), which is a
// contiguous range [ql, qh) of the sorted queries, found per tile by two binary searches in a tiny
// pre-pass (one thread per tile).  The owner does an LDS binary search and scans forward while
// start < q.end -- through the halo and, for the rare long scan, on into global memory -- applying
// exactly the reference's hit rule
//     min(r.end, qe) - max(r.start, qs) >= min_overlap                           (igd.rs:792-795)
// One thread sees all hits of a query, in database order, so pairwise counts see each (query, record)
// pair once and binary (LOLA support) counting credits a (query, file) pair at its first hit.
// Per-file counts go to u32 LDS bins, flushed once per workgroup with u64 atomics.
//
// Bound: HBM.  Algorithmic bytes 12*Nq + 16*Ndb + 8*F; no MFMA (integer compare/scan).
#include <algorithm>
#include <mutex>

#include "common.h"
#include "scan.h"

namespace gtars {

#ifndef GTARS_IGD_ABLATE
#define GTARS_IGD_ABLATE 0  // timing experiments (results wrong): 1 one record per query, 2 no histogram atomics, 4 no prefix-max scan
#endif
constexpr int IGD_TILE = (int)IGD_TILE_RECORDS;
constexpr int SW_TPB = 512;

constexpr int IGD_HALO = 256;
constexpr int IGD_SEEN = 32;   // per-thread list of credited files (binary counting)  // records after the tile kept in LDS too (a query's scan may run past its tile)

// ---- query preparation: validity rules of Igd::count_overlaps (igd.rs:514-517) ------------------
// Also finds every query's OWNER tile (IgdTiles::bnd: the first tile of its chromosome with bnd > start) by a binary
// search of the bounds, which the workgroup keeps in LDS (4 B per tile), and notes whether the batch is already in
// (chromosome, start) order.  tid = n_tiles: no owner (invalid query, unknown chromosome, or past every record's reach).
constexpr int PREP_TPB = 1024;
__device__ __forceinline__ void igd_prep_one(u32 c_in, u32 s_in, u32 e_in, u32 n_chrom, u32 &c, i32 &s, i32 &e) {
    s = (i32)s_in;
    e = (i32)e_in;  // `as i32` (igd.rs:549-550)
    c = c_in;
    if (s >= e || e <= 0 || c >= n_chrom) {
        c = n_chrom;  // sorts behind every real chromosome; never served
        s = 0;
        e = 0;
    } else if (s < 0) {
        s = 0;  // clamp (igd.rs:517)
    }
}

template <bool BUCKET>
__global__ void __launch_bounds__(PREP_TPB)
k_igd_prep_queries(const u32 *__restrict__ qc, const u32 *__restrict__ qs, const u32 *__restrict__ qe, u32 nq, u32 n_chrom,
                   const u32 *__restrict__ bnd, const u32 *__restrict__ chrom_tile_off, u32 n_tiles, u32 chunk,
                   u32 *__restrict__ kc, u32 *__restrict__ ks, u32 *__restrict__ ke, u32 *__restrict__ tid,
                   u32 *__restrict__ unsorted) {
    extern __shared__ u32 s_bnd[];  // [n_tiles] | chrom_tile_off [n_chrom + 1]
    u32 *s_cto = s_bnd + n_tiles;
    if (BUCKET) {
        for (u32 t = threadIdx.x; t < n_tiles; t += PREP_TPB) s_bnd[t] = bnd[t];
        for (u32 c = threadIdx.x; c <= n_chrom; c += PREP_TPB) s_cto[c] = chrom_tile_off[c];
        __syncthreads();
    }
    const u32 lo = blockIdx.x * chunk, hi = min(nq, lo + chunk);
    const int lane = threadIdx.x & 63;
    bool bad = false;
    for (u32 base = lo; base < hi; base += PREP_TPB) {
        const u32 i = base + threadIdx.x;
        const bool ok = i < hi;
        u32 c = n_chrom;
        i32 s = 0, e = 0;
        if (ok) igd_prep_one(qc[i], qs[i], qe[i], n_chrom, c, s, e);
        // already in (chromosome, start) order?  then the sweep can skip the partition (BED inputs usually are)
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
            if (BUCKET) {
                u32 t = n_tiles;
                if (c < n_chrom) {
                    u32 l = s_cto[c];
                    const u32 h0 = s_cto[c + 1];
                    u32 h = h0;
                    while (l < h) {
                        const u32 mid = l + ((h - l) >> 1);
                        if (s_bnd[mid] <= (u32)s)
                            l = mid + 1;
                        else
                            h = mid;
                    }
                    t = l < h0 ? l : n_tiles;
                }
                tid[i] = t;
            }
        }
    }
    if (__any(bad) && lane == 0) *unsorted = 1u;
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

// first index in [lo, hi) with a[i] >= key
__device__ __forceinline__ u32 lb_u32(const u32 *__restrict__ a, u32 lo, u32 hi, u64 key) {
    while (lo < hi) {
        const u32 mid = lo + ((hi - lo) >> 1);
        if ((u64)a[mid] < key)
            lo = mid + 1;
        else
            hi = mid;
    }
    return lo;
}

// per-chromosome segments of the sorted queries
// `perm` (may be NULL: the batch is already ordered) maps a sorted position to its row of `chrom_key`, so the
// sorted chromosome column never has to be materialised for these n_chrom + 1 binary searches
__global__ void k_igd_chrom_segments(const u32 *__restrict__ chrom_key, const u32 *__restrict__ perm, u32 nq, u32 n_chrom,
                                     u32 *__restrict__ cq_off, const u32 *__restrict__ skip_if) {
    if (skip_if && *skip_if) return;  // the batch was partitioned instead (decided on the device)
    const u32 c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c > n_chrom) return;
    u32 lo = 0, hi = nq;
    while (lo < hi) {
        const u32 mid = lo + ((hi - lo) >> 1);
        if (chrom_key[perm ? perm[mid] : mid] < c)
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

// Every query is OWNED by exactly one tile: the tile that holds its lower_bound position
// p = first record of the chromosome with start >= key, key = max(q.start - max_len, 0).
// key is monotone in q.start, so the owned queries of a tile are a contiguous range [ql, qh) of the
// sorted queries:   last_start(previous tile) < key <= last_start(this tile).
__global__ void k_igd_tile_ranges(IgdView v, const u32 *__restrict__ tile_first, const u32 *__restrict__ tile_cnt,
                                  const u32 *__restrict__ tile_chrom, u32 n_tiles, const u32 *__restrict__ sorted_qs,
                                  const u32 *__restrict__ cq_off, u32 *__restrict__ ql, u32 *__restrict__ qh,
                                  const u32 *__restrict__ skip_if) {
    if (skip_if && *skip_if) return;
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_tiles) return;
    const u32 c = tile_chrom[t], p0 = tile_first[t], cnt = tile_cnt[t];
    const u64 max_len = (u64)v.chrom_maxlen[c];
    const u32 lo = cq_off[c], hi = cq_off[c + 1];
    const bool first_of_chrom = p0 == v.chrom_off[c];
    // key > prev_last  <=>  q.start >= prev_last + max_len + 1   (keys clamped to 0 belong to the first tile)
    ql[t] = first_of_chrom ? lo : lb_u32(sorted_qs, lo, hi, (u64)v.starts[p0 - 1] + max_len + 1);
    qh[t] = lb_u32(sorted_qs, lo, hi, (u64)v.starts[p0 + cnt - 1] + max_len + 1);
}

// ---- the sweep ---------------------------------------------------------------------------------
// MODE 0: pairwise counts (count_set_overlaps); 1: binary counts (count_region_hits) with a per-query list of credited
// files; 2: binary counts for min_overlap == 1 through pme_file -- a record is the FIRST hit of its file for a query iff
// no earlier record of that file (and chromosome) ends after the query's start: earlier records start no later, so
// "ends after q.start" is all that is left of the overlap test, and the largest such end is a per-record constant of
// the database (IgdTiles::pme_file).  Binary counting then costs what pairwise counting costs (2.3 -> 0.5 ms for
// config 3) instead of a 32-entry membership test per hit.
template <int MODE>
__global__ void __launch_bounds__(SW_TPB)
k_igd_sweep(IgdView v, const i32 *__restrict__ pme_file, const u32 *__restrict__ tile_first, const u32 *__restrict__ tile_cnt,
            const u32 *__restrict__ tile_chrom, const i32 *__restrict__ tile_carry, u32 n_tiles,
            const u32 *__restrict__ sqs, const u32 *__restrict__ sqe, int interleaved,
            const u32 *__restrict__ ql, const u32 *__restrict__ qh, i32 min_overlap,
            unsigned long long *__restrict__ hits, const u32 *__restrict__ part_flag, const u32 *__restrict__ part_ab,
            const u32 *__restrict__ part_ql, u32 cq) {
    extern __shared__ __attribute__((aligned(16))) u32 sm[];
    // whether the batch had to be partitioned was decided on the device (k_igd_prep_queries): take the partition's
    // interleaved (start, end) pairs and bin offsets, or the batch as it arrived with the tile ranges
    if (part_flag && *part_flag) {
        sqs = part_ab;
        sqe = nullptr;
        interleaved = 1;
        ql = part_ql;
        qh = part_ql + 1;
    }
    constexpr int CAP = IGD_TILE + IGD_HALO;
    i32 *t_s = reinterpret_cast<i32 *>(sm);
    i32 *t_e = t_s + CAP;
    // file ids as u16 (the LDS histogram limits a database to 16384 files): 4.5 KB less per workgroup -- with the chunk arrays
    // below it decides between 2 and 3 workgroups per CU for the pme-binary form, and every workgroup less costs 20-75 %
    unsigned short *t_f = reinterpret_cast<unsigned short *>(t_e + CAP);
    constexpr bool BINARY = MODE == 1;
    i32 *t_pm = reinterpret_cast<i32 *>(t_f + CAP);  // prefix maximum of the ends (carry-in included); CAP is even
    i32 *t_pf = t_pm + CAP;                           // MODE 2: pme_file of the staged records
    u32 *bins = reinterpret_cast<u32 *>(t_pf + (MODE == 2 ? CAP : 0));  // [n_files]
    // MODE 0 / 2 (a hit is decided by the (query, record) pair alone): the queries of a chunk with their record
    // ranges, so that the PAIRS can be dealt evenly to the threads
    // chunk arrays (cq = 128 / 256 / 512 queries per chunk, chosen by the launcher from the batch's queries per tile: sparse
    // batches -- a LOLA universe against a large database -- need few, and the LDS saved is a workgroup more per CU)
    u32 *c_off = bins + ((v.n_files + 1u) & ~1u);                   // [cq + 1] (+1 pad)
    i32 *c_s = reinterpret_cast<i32 *>(c_off + cq + 2), *c_e = c_s + cq;
    unsigned short *c_lo = reinterpret_cast<unsigned short *>(c_e + cq);  // [cq], values < CAP
    __shared__ u32 s_part[SW_TPB / 64];
    __shared__ i32 s_wmax[SW_TPB / 64];
    for (u32 i = threadIdx.x; i < v.n_files; i += SW_TPB) bins[i] = 0;

    // Software pipeline over the workgroup's tiles: the NEXT tile's records are loaded into registers
    // before the current tile's queries are served and committed to LDS afterwards, so their HBM latency
    // hides behind the LDS-bound query loop.
    constexpr int RPT = (CAP + SW_TPB - 1) / SW_TPB;
    i32 rg_s[RPT], rg_e[RPT], rg_p[MODE == 2 ? RPT : 1];
    u32 rg_f[RPT];
    struct TileDesc {
        u32 p0, cnt, c, seg_hi, n_lds;
        i32 carry;
    };
    auto describe = [&](u32 t) {
        TileDesc d;
        d.p0 = tile_first[t];
        d.cnt = tile_cnt[t];
        d.c = tile_chrom[t];
        d.seg_hi = v.chrom_off[d.c + 1];
        d.n_lds = min((u32)CAP, d.seg_hi - d.p0);  // tile + halo, never past the chromosome
        d.carry = tile_carry[t];
        return d;
    };
    auto issue = [&](const TileDesc &d) {
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            const u32 i = threadIdx.x + (u32)k * SW_TPB;
            if (i < d.n_lds) {
                rg_s[k] = v.starts[d.p0 + i];
                rg_e[k] = v.ends[d.p0 + i];
                rg_f[k] = v.files[d.p0 + i];
                if (MODE == 2) rg_p[MODE == 2 ? k : 0] = pme_file[d.p0 + i];
            }
        }
    };
    auto commit = [&](const TileDesc &d) {
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            const u32 i = threadIdx.x + (u32)k * SW_TPB;
            if (i < d.n_lds) {
                t_s[i] = rg_s[k];
                t_e[i] = rg_e[k];
                t_f[i] = (unsigned short)rg_f[k];
                if (MODE == 2) t_pf[i] = rg_p[MODE == 2 ? k : 0];
            }
        }
    };
    TileDesc cur{}, nxt{};
    if (blockIdx.x < n_tiles) {
        cur = describe(blockIdx.x);
        issue(cur);
        __syncthreads();  // bins zeroed
        commit(cur);
    }
    for (u32 tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        __syncthreads();  // the current tile is in LDS
        {
            // t_pm[i] = max(carry, ends[0..i]): ascending, so "first record that can overlap a query" is a
            // binary search for t_pm > q_start.  Blocked layout: thread t owns records [t*RPT, (t+1)*RPT).
            const u32 base = threadIdx.x * RPT;
            i32 loc[RPT];
            i32 run = 0;  // ends are > 0 (Igd::add drop rule)
#pragma unroll
            for (int k = 0; k < RPT; ++k) {
                const u32 i = base + k;
                run = max(run, i < cur.n_lds ? t_e[i] : 0);
                loc[k] = run;
            }
            // exclusive maximum over the threads before this one: wave shuffles, then one LDS word per wave
            const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
            const i32 inc = wave_inclusive_max_nonneg(run);  // ends are > 0
            if (lane == 63) s_wmax[wave] = inc;
            const i32 excl = __builtin_amdgcn_update_dpp(0, inc, 0x138, 0xf, 0xf, false);  // wave_shr:1; lane 0 gets 0
            __syncthreads();
            i32 before = cur.carry;
            for (int w = 0; w < wave; ++w) before = max(before, s_wmax[w]);
            before = max(before, excl);
#pragma unroll
            for (int k = 0; k < RPT; ++k) {
                const u32 i = base + k;
                if (i < cur.n_lds) t_pm[i] = max(before, loc[k]);
            }
            __syncthreads();
        }
        const u32 next = tile + gridDim.x;
        if (next < n_tiles) {
            nxt = describe(next);
            issue(nxt);
        }
        const u32 p0 = cur.p0, cnt = cur.cnt, c = cur.c, seg_hi = cur.seg_hi, n_lds = cur.n_lds;
        const i32 max_len = v.chrom_maxlen[c];
        const u32 n_seg = seg_hi - p0;  // records from the tile start to the end of the chromosome
        // record i (relative to p0): LDS if staged, global otherwise (rare: scans longer than the halo)
        auto r_start = [&](u32 i) -> i32 { return i < n_lds ? t_s[i] : v.starts[p0 + i]; };
        auto r_end = [&](u32 i) -> i32 { return i < n_lds ? t_e[i] : v.ends[p0 + i]; };
        auto r_file = [&](u32 i) -> u32 { return i < n_lds ? t_f[i] : v.files[p0 + i]; };
        const u32 q_lo = ql[tile], q_hi = qh[tile];
        if constexpr (MODE != 1) {
            // Pair-balanced scan.  A query's candidates are the records [lo, hi): lo = first record that can overlap it
            // (prefix-max end > q_start), hi = first record that starts at or after q_end (the reference's scan
            // stops there, igd.rs:772-846).  Per chunk of cq queries: one thread per query finds (lo, hi), the
            // lengths are scanned, and every thread then takes the same number of consecutive (query, record) pairs
            // -- no lane waits for the longest scan of its wave (a thread-per-query loop runs 20 records on average
            // and 45 for the slowest lane).  Records past the staged range (rare) are scanned by the query's own thread.
            const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
            for (u32 cb = q_lo; cb < q_hi; cb += cq) {
                const u32 qi = cb + threadIdx.x;
                i32 s = 0, e = 0;
                u32 lo = 0, len = 0;
                if (threadIdx.x < cq && qi < q_hi) {
                    if (interleaved) {
                        const uint2 se2 = reinterpret_cast<const uint2 *>(sqs)[qi];
                        s = (i32)se2.x;
                        e = (i32)se2.y;
                    } else {
                        s = (i32)sqs[qi];
                        e = (i32)sqe[qi];
                    }
                    u32 hi;
                    if (min_overlap >= 1) {
                        hi = n_lds;
                        while (lo < hi) {
                            const u32 mid = lo + ((hi - lo) >> 1);
                            if (t_pm[mid] <= s)
                                lo = mid + 1;
                            else
                                hi = mid;
                        }
                    } else {
                        const i32 key = s > max_len ? s - max_len : 0;
                        hi = cnt;
                        while (lo < hi) {
                            const u32 mid = lo + ((hi - lo) >> 1);
                            if (t_s[mid] < key)
                                lo = mid + 1;
                            else
                                hi = mid;
                        }
                    }
                    // first staged record at or after lo whose start is >= q_end
                    u32 a = lo, b = n_lds;
                    while (a < b) {
                        const u32 mid = a + ((b - a) >> 1);
                        if (t_s[mid] < e)
                            a = mid + 1;
                        else
                            b = mid;
                    }
                    len = a - lo;
                    if (a == n_lds && n_seg > n_lds) {
                        // the scan runs past the staged records: the rest from global memory, by this thread
                        for (u32 r = max(lo, n_lds); r < n_seg; ++r) {
                            const i32 rs = v.starts[p0 + r], re = v.ends[p0 + r];
                            if (rs >= e) break;
                            const i32 ov = (re < e ? re : e) - (rs > s ? rs : s);
                            if (ov < min_overlap) continue;
                            if (MODE == 2 && pme_file[p0 + r] > s) continue;
                            atomicAdd(&bins[v.files[p0 + r]], 1u);
                        }
                    }
                }
                if (threadIdx.x < cq) {
                    c_s[threadIdx.x] = s;
                    c_e[threadIdx.x] = e;
                    c_lo[threadIdx.x] = (unsigned short)lo;
                }
                // exclusive scan of the lengths over the chunk
                const u32 inc = wave_inclusive_scan_u32(len, lane);
                if (lane == 63) s_part[wave] = inc;
                __syncthreads();
                u32 wbase = 0, total = 0;
#pragma unroll
                for (int w = 0; w < SW_TPB / 64; ++w) {
                    const u32 x = s_part[w];
                    wbase += w < wave ? x : 0u;
                    total += x;
                }
                if (threadIdx.x < cq) c_off[threadIdx.x] = wbase + inc - len;
                if (threadIdx.x == 0) c_off[cq] = total;
                __syncthreads();
                if (total) {
                    const u32 ppt = (total + SW_TPB - 1) / SW_TPB;
                    u32 p = threadIdx.x * ppt;
                    const u32 p_end = min(p + ppt, total);
                    if (p < p_end) {
                        // the query that holds pair p: last j with c_off[j] <= p
                        u32 j = 0, jh = cq;
                        while (jh - j > 1) {
                            const u32 mid = (j + jh) >> 1;
                            if (c_off[mid] <= p)
                                j = mid;
                            else
                                jh = mid;
                        }
                        u32 j_end = c_off[j + 1];
                        u32 r = c_lo[j] + (p - c_off[j]);
                        i32 qs_ = c_s[j], qe_ = c_e[j];
                        for (; p < p_end; ++p, ++r) {
                            if (p == j_end) {  // next query with a non-empty range
                                do {
                                    ++j;
                                    j_end = c_off[j + 1];
                                } while (j_end == p);
                                r = c_lo[j];
                                qs_ = c_s[j];
                                qe_ = c_e[j];
                            }
                            const i32 rs = t_s[r], re = t_e[r];
                            const i32 ov = (re < qe_ ? re : qe_) - (rs > qs_ ? rs : qs_);
                            bool hit = ov >= min_overlap;
                            if (MODE == 2) hit = hit && t_pf[r] <= qs_;  // no earlier record of this file reaches the query
                            if (hit) {
                                if (GTARS_IGD_ABLATE & 2) {
                                    if (rs == 0x7FFFFFF0) bins[0] = 1;
                                } else {
                                    atomicAdd(&bins[t_f[r]], 1u);
                                }
                            }
                        }
                    }
                }
                __syncthreads();  // c_* reused by the next chunk
            }
        } else
        for (u32 qi = q_lo + threadIdx.x; qi < q_hi; qi += SW_TPB) {
            // (start, end) pairs as the partition leaves them, or two sorted columns
            i32 s, e;
            if (interleaved) {
                const uint2 se2 = reinterpret_cast<const uint2 *>(sqs)[qi];
                s = (i32)se2.x;
                e = (i32)se2.y;
            } else {
                s = (i32)sqs[qi];
                e = (i32)sqe[qi];
            }
            u32 lo = 0, hi;
            if (min_overlap >= 1) {
                // an overlap of >= 1 bp needs end > q_start: start at the first staged record whose prefix-max
                // end is > q_start.  It is never before lower_bound(q_start - max_len) (everything in between
                // ends at or before q_start), so ownership by this tile still holds; if no staged record
                // qualifies the scan goes on in global memory from the end of the staged range.
                hi = n_lds;
                while (lo < hi) {
                    const u32 mid = lo + ((hi - lo) >> 1);
                    if (t_pm[mid] <= s)
                        lo = mid + 1;
                    else
                        hi = mid;
                }
            } else {
                const i32 key = s > max_len ? s - max_len : 0;
                // lower_bound of key: inside the tile by ownership
                hi = cnt;
                while (lo < hi) {
                    const u32 mid = lo + ((hi - lo) >> 1);
                    if (t_s[mid] < key)
                        lo = mid + 1;
                    else
                        hi = mid;
                }
            }
            // binary counting: the files already credited to this query, as packed u16 pairs in
            // registers (no memory latency in the membership test); 0xFFFF = empty (n_files <= 16384)
            u32 n_seen = 0;
            u32 sl[IGD_SEEN / 2];
#pragma unroll
            for (int k = 0; k < IGD_SEEN / 2; ++k) sl[k] = 0xFFFFFFFFu;
            for (u32 r = lo; r < ((GTARS_IGD_ABLATE & 1) ? min(n_seg, lo + 1u) : n_seg); ++r) {
                // the three fields of a record in one LDS round trip
                i32 rs, re, pf = 0;
                u32 f;
                if (r < n_lds) {
                    rs = t_s[r];
                    re = t_e[r];
                    f = t_f[r];
                    if (MODE == 2) pf = t_pf[r];
                } else {
                    rs = v.starts[p0 + r];
                    re = v.ends[p0 + r];
                    f = v.files[p0 + r];
                    if (MODE == 2) pf = pme_file[p0 + r];
                }
                if (rs >= e) break;
                const i32 ov = (re < e ? re : e) - (rs > s ? rs : s);
                if (ov < min_overlap) continue;
                if (MODE == 2 && pf > s) continue;  // an earlier record of this file already hit the query
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
                    atomicAdd(&bins[f], 1u);
                }
            }
        }
        __syncthreads();  // every query of the current tile served: LDS may be overwritten
        if (next < n_tiles) {
            commit(nxt);
            cur = nxt;
        }
    }
    __syncthreads();
    for (u32 i = threadIdx.x; i < v.n_files; i += SW_TPB) {
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
    GT_HIP(hipMemcpyAsync(k0, v.files, (size_t)n * 4, hipMemcpyDeviceToDevice, st));
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
    if (getenv("GTARS_NO_IGD_SWEEP")) return false;
    // The sweep reads the whole database once per batch (0.37 ms per 5e7 records) whatever the batch size; the per-query
    // kernel costs ~1.2 ns per query at config-3/4 densities.  Crossover measured on config 4 (5e7 records): ~300k
    // queries -- the 1e5-region user set goes per query (0.17 instead of 0.37 ms), the 1e6-region universe sweeps.
    const u64 min_q = getenv("GTARS_IGD_SWEEP_MIN") ? (u64)atoll(getenv("GTARS_IGD_SWEEP_MIN"))
                                                    : std::max<u64>(1u << 16, (u64)v.n / 256);
    // u32 LDS bins: a workgroup adds at most (its queries x hits) -- keep the batch below 2^31 queries
    return v.n > 0 && v.n_files > 0 && v.n_files <= 16384 && nq >= min_q && nq < (1ull << 31);
}

size_t igd_sweep_ws_bytes(u64 nq, u32 n_tiles, u32 n_chrom) {
    // kc ks ke | sorted qs qe chrom | perm (or owner tiles) | ql qh | cq_off | bin offsets | slack | partition / sort scratch
    return (size_t)nq * 4 * 7 + (size_t)n_tiles * 12 + ((size_t)n_chrom + 2) * 4 + 512 +
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

gtars_status launch_igd_sweep(const IgdView &v, const IgdTiles &tl, const u32 *qc, const u32 *qs, const u32 *qe, u64 nq64,
                              i32 min_overlap, int binary, u64 *hits, void *ws, size_t ws_bytes, hipStream_t st) {
    const u32 nq = (u32)nq64;
    const u32 n_tiles = tl.n_tiles;
    if (ws_bytes < igd_sweep_ws_bytes(nq, n_tiles, v.n_chrom)) return fail(GTARS_ERR_INTERNAL, "IGD sweep workspace too small");
    GT_HIP(hipMemsetAsync(hits, 0, sizeof(u64) * v.n_files, st));
    u32 *kc = (u32 *)ws, *ks = kc + nq, *ke = ks + nq;
    u32 *sc = ke + nq, *ss = sc + nq, *se = ss + nq, *perm = se + nq;  // perm doubles as the owner-tile column
    u32 *ql = perm + nq, *qh = ql + n_tiles, *cq_off = qh + n_tiles;
    u32 *bin_off = cq_off + v.n_chrom + 2;  // [n_tiles + 2]
    u32 *d_unsorted = bin_off + n_tiles + 2;
    void *scratch = (void *)(((uintptr_t)(d_unsorted + 16) + 63) & ~(uintptr_t)63);
    const size_t scratch_bytes = ws_bytes - (size_t)((char *)scratch - (char *)ws);
    (void)sc;
    // GTARS_IGD_ALWAYS_SORT (tests): start from "not in order" -- the preparation kernel only ever raises the flag
    GT_HIP(hipMemsetAsync(d_unsorted, getenv("GTARS_IGD_ALWAYS_SORT") ? 1 : 0, 1, st));
    GT_HIP(hipMemsetAsync((char *)d_unsorted + 1, 0, 3, st));
    // queries grouped by owner tile in one partition pass, when the tile bounds fit in LDS (76M records); otherwise
    // (and for GTARS_IGD_FULL_SORT=1) the batch is fully sorted by (chromosome, start) with the radix sort
    const bool full_sort = getenv("GTARS_IGD_FULL_SORT") != nullptr;  // tests / A-B runs
    const bool bucket = !full_sort && tl.bnd && n_tiles + 1 <= MS_MAX_BINS;
    int dev = 0, cus = 256;
    GT_HIP(hipGetDevice(&dev));
    GT_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    {
        ProfScope p("k_igd_prep_queries", st);
        const u32 n_wg = std::max<u32>(1, std::min<u32>((u32)cus, (nq + 4095) / 4096));
        const u32 chunk = ((nq + n_wg - 1) / n_wg + PREP_TPB - 1) / PREP_TPB * PREP_TPB;
        const size_t lds = bucket ? ((size_t)n_tiles + v.n_chrom + 1) * 4 : 0;
        auto kern = bucket ? k_igd_prep_queries<true> : k_igd_prep_queries<false>;
        if (lds > 48 * 1024)
            GT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)((MS_MAX_BINS + 4096) * 4)));
        hipLaunchKernelGGL(kern, dim3(n_wg), dim3(PREP_TPB), lds, st, qc, qs, qe, nq, v.n_chrom, tl.bnd, tl.chrom_tile_off, n_tiles,
                           chunk, kc, ks, ke, perm, d_unsorted);
    }
    const u32 *t_ql = ql, *t_qh = qh;
    int interleaved = 0;
    const u32 *part_flag = nullptr, *part_ab = nullptr, *part_ql = nullptr;
    u32 h_unsorted = 1;
    if (bucket) {
        // No host round trip: both continuations are enqueued and the flag the preparation kernel leaves on the device
        // picks one -- the partition kernels return at once for a batch that is already in owner order, the range
        // kernels for one that is not, and the sweep takes its inputs accordingly.
        // K1 (multisplit): (start, end) pairs grouped by owner tile; bin_off[t], bin_off[t + 1] bound tile t's queries
        gtars_status s1 = multisplit_pairs(perm, ks, ke, nq, n_tiles + 1, n_tiles, reinterpret_cast<uint2 *>(ss), bin_off, scratch,
                                           scratch_bytes, st, d_unsorted);  // ss and se are adjacent: 2 * nq words
        if (s1) return s1;
        part_flag = d_unsorted;
        part_ab = ss;
        part_ql = bin_off;
        ss = ks;
        se = ke;
        ProfScope p("k_igd_tile_ranges", st);
        hipLaunchKernelGGL(k_igd_chrom_segments, dim3((v.n_chrom + 1 + 63) / 64), dim3(64), 0, st, kc, (const u32 *)nullptr, nq,
                           v.n_chrom, cq_off, d_unsorted);
        hipLaunchKernelGGL(k_igd_tile_ranges, dim3((n_tiles + 255) / 256), dim3(256), 0, st, v, tl.first, tl.cnt, tl.chrom, n_tiles, ss,
                           cq_off, ql, qh, d_unsorted);
    } else {
        if (!getenv("GTARS_IGD_ALWAYS_SORT")) {
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
        hipLaunchKernelGGL(k_igd_chrom_segments, dim3((v.n_chrom + 1 + 63) / 64), dim3(64), 0, st, kc, perm, nq, v.n_chrom, cq_off,
                           (const u32 *)nullptr);
        hipLaunchKernelGGL(k_igd_tile_ranges, dim3((n_tiles + 255) / 256), dim3(256), 0, st, v, tl.first, tl.cnt, tl.chrom, n_tiles, ss,
                           cq_off, ql, qh, (const u32 *)nullptr);
    }
    const int mode = !binary ? 0 : (min_overlap == 1 && tl.pme_file ? 2 : 1);
    // starts | ends | files (u16) | prefix-max ends | [pme_file] | bins
    // + chunk arrays: offsets u32 [cq + 2] | starts, ends i32 [cq] | first records u16 [cq]
    u32 cq = 512;
    if (mode != 1) {
        const u64 per_tile = n_tiles ? (u64)nq / n_tiles : nq;  // queries per tile on average
        cq = per_tile <= 48 ? 128u : per_tile <= 110 ? 256u : 512u;
    }
    const size_t lds = (size_t)(IGD_TILE + IGD_HALO) * (mode == 2 ? 18 : 14) + (((size_t)v.n_files + 1) & ~(size_t)1) * 4 +
                       (mode != 1 ? ((size_t)cq + 2) * 4 + (size_t)cq * 10 : 0);
    auto kern = mode == 2 ? k_igd_sweep<2> : mode == 1 ? k_igd_sweep<1> : k_igd_sweep<0>;
    {
        // the dynamic-LDS limit belongs to the function, not to the calling thread: raised once per device to the largest
        // size any launch can ask for (5 staged arrays + 16384 file bins) and never lowered
        static std::mutex mu;
        static bool done[3][64] = {};
        int dev = 0;
        GT_HIP(hipGetDevice(&dev));
        std::lock_guard<std::mutex> lock(mu);
        if (dev >= 0 && dev < 64 && !done[mode][dev]) {
            GT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)(((size_t)(IGD_TILE + IGD_HALO) * 5 + 16384 + 2048) * 4)));
            done[mode][dev] = true;
        }
    }
    int per_cu = 1;
    GT_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, SW_TPB, lds));
    if (per_cu < 1) return fail(GTARS_ERR_INTERNAL, "k_igd_sweep does not fit on a CU");
    const unsigned grid = (unsigned)std::min<u64>((u64)cus * per_cu, n_tiles);
    {
        ProfScope p(binary ? "k_igd_sweep<binary>" : "k_igd_sweep<pairwise>", st);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(SW_TPB), lds, st, v, tl.pme_file, tl.first, tl.cnt, tl.chrom, tl.carry, n_tiles, ss, se,
                           interleaved, t_ql, t_qh, min_overlap, (unsigned long long *)hits, part_flag, part_ab, part_ql, cq);
    }
    GT_HIP(hipGetLastError());
    // which continuation the device took (profiling mode: a deterministic fact for the tests, not a timing)
    if (bucket) prof_note_device_flag("igd_batch_partitioned", "igd_batch_in_owner_order", d_unsorted, st);
    return GTARS_OK;
}

}  // namespace gtars
