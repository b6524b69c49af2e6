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

#include "common.h"
#include "scan.h"

namespace gtars {

constexpr int IGD_TILE = (int)IGD_TILE_RECORDS;
constexpr int SW_TPB = 512;

constexpr int IGD_HALO = 256;
constexpr int IGD_SEEN = 32;   // per-thread list of credited files (binary counting)  // records after the tile kept in LDS too (a query's scan may run past its tile)

// ---- query preparation: validity rules of Igd::count_overlaps (igd.rs:514-517) ------------------
__global__ void k_igd_prep_queries(const u32 *__restrict__ qc, const u32 *__restrict__ qs, const u32 *__restrict__ qe,
                                   u32 nq, u32 n_chrom, u32 *__restrict__ kc, u32 *__restrict__ ks,
                                   u32 *__restrict__ ke, u32 *__restrict__ unsorted) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq) return;
    auto prep = [&](u32 k, u32 &c, i32 &s, i32 &e) {
        s = (i32)qs[k];
        e = (i32)qe[k];  // `as i32` (igd.rs:549-550)
        c = qc[k];
        if (s >= e || e <= 0 || c >= n_chrom) {
            c = n_chrom;  // sorts behind every real chromosome; never served
            s = 0;
            e = 0;
        } else if (s < 0) {
            s = 0;  // clamp (igd.rs:517)
        }
    };
    u32 c;
    i32 s, e;
    prep(i, c, s, e);
    kc[i] = c;
    ks[i] = (u32)s;
    ke[i] = (u32)e;
    if (i > 0) {
        // already in (chromosome, start) order?  then the sweep can skip its sort (BED inputs usually are)
        u32 pc;
        i32 ps, pe;
        prep(i - 1, pc, ps, pe);
        if (pc > c || (pc == c && (u32)ps > (u32)s)) *unsorted = 1u;
    }
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
                                     u32 *__restrict__ cq_off) {
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
                                  const u32 *__restrict__ cq_off, u32 *__restrict__ ql, u32 *__restrict__ qh) {
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
template <bool BINARY>
__global__ void __launch_bounds__(SW_TPB)
k_igd_sweep(IgdView v, const u32 *__restrict__ tile_first, const u32 *__restrict__ tile_cnt,
            const u32 *__restrict__ tile_chrom, const i32 *__restrict__ tile_carry, u32 n_tiles,
            const u32 *__restrict__ sqs, const u32 *__restrict__ sqe,
            const u32 *__restrict__ ql, const u32 *__restrict__ qh, i32 min_overlap,
            unsigned long long *__restrict__ hits) {
    extern __shared__ __attribute__((aligned(16))) u32 sm[];
    constexpr int CAP = IGD_TILE + IGD_HALO;
    i32 *t_s = reinterpret_cast<i32 *>(sm);
    i32 *t_e = t_s + CAP;
    u32 *t_f = reinterpret_cast<u32 *>(t_e + CAP);
    i32 *t_pm = reinterpret_cast<i32 *>(t_f + CAP);  // prefix maximum of the ends (carry-in included)
    u32 *bins = reinterpret_cast<u32 *>(t_pm + CAP);  // [n_files]
    __shared__ i32 s_wmax[SW_TPB / 64];
    for (u32 i = threadIdx.x; i < v.n_files; i += SW_TPB) bins[i] = 0;

    // Software pipeline over the workgroup's tiles: the NEXT tile's records are loaded into registers
    // before the current tile's queries are served and committed to LDS afterwards, so their HBM latency
    // hides behind the LDS-bound query loop.
    constexpr int RPT = (CAP + SW_TPB - 1) / SW_TPB;
    i32 rg_s[RPT], rg_e[RPT];
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
                t_f[i] = rg_f[k];
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
            i32 inc = run;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const i32 y = __shfl_up(inc, d, 64);
                if (lane >= d) inc = max(inc, y);
            }
            if (lane == 63) s_wmax[wave] = inc;
            i32 excl = __shfl_up(inc, 1, 64);
            if (lane == 0) excl = 0;
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
        for (u32 qi = q_lo + threadIdx.x; qi < q_hi; qi += SW_TPB) {
            const i32 s = (i32)sqs[qi], e = (i32)sqe[qi];
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
            for (u32 r = lo; r < n_seg; ++r) {
                // the three fields of a record in one LDS round trip
                i32 rs, re;
                u32 f;
                if (r < n_lds) {
                    rs = t_s[r];
                    re = t_e[r];
                    f = t_f[r];
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
                atomicAdd(&bins[f], 1u);
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

// ---- launcher ------------------------------------------------------------------------------------

bool igd_sweep_supported(const IgdView &v, u64 nq) {
    if (getenv("GTARS_NO_IGD_SWEEP")) return false;
    const u64 min_q = getenv("GTARS_IGD_SWEEP_MIN") ? (u64)atoll(getenv("GTARS_IGD_SWEEP_MIN")) : (1u << 16);
    // u32 LDS bins: a workgroup adds at most (its queries x hits) -- keep the batch below 2^31 queries
    return v.n > 0 && v.n_files > 0 && v.n_files <= 16384 && nq >= min_q && nq < (1ull << 31);
}

size_t igd_sweep_ws_bytes(u64 nq, u32 n_tiles, u32 n_chrom) {
    // kc ks ke | sorted qs qe chrom | perm | ql qh | cq_off | max_qlen | sort scratch
    return (size_t)nq * 4 * 7 + (size_t)n_tiles * 8 + ((size_t)n_chrom + 2) * 4 + 256 + device_sort_perm_ws_bytes((u32)nq);
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

gtars_status launch_igd_sweep(const IgdView &v, const u32 *tile_first, const u32 *tile_cnt, const u32 *tile_chrom,
                              const i32 *tile_carry, u32 n_tiles, const u32 *qc, const u32 *qs, const u32 *qe, u64 nq64, i32 min_overlap,
                              int binary, u64 *hits, void *ws, size_t ws_bytes, hipStream_t st) {
    const u32 nq = (u32)nq64;
    if (ws_bytes < igd_sweep_ws_bytes(nq, n_tiles, v.n_chrom)) return fail(GTARS_ERR_INTERNAL, "IGD sweep workspace too small");
    GT_HIP(hipMemsetAsync(hits, 0, sizeof(u64) * v.n_files, st));
    u32 *kc = (u32 *)ws, *ks = kc + nq, *ke = ks + nq;
    u32 *sc = ke + nq, *ss = sc + nq, *se = ss + nq, *perm = se + nq;
    u32 *ql = perm + nq, *qh = ql + n_tiles, *cq_off = qh + n_tiles;
    void *sort_ws = (void *)(((uintptr_t)(cq_off + v.n_chrom + 2) + 63) & ~(uintptr_t)63);
    const size_t sort_ws_bytes = device_sort_perm_ws_bytes(nq);
    const unsigned g = (nq + 255) / 256;
    u32 *d_unsorted = (u32 *)((char *)ws + igd_sweep_ws_bytes(nq, n_tiles, v.n_chrom) - 64);  // inside the slack
    GT_HIP(hipMemsetAsync(d_unsorted, 0, sizeof(u32), st));
    {
        ProfScope p("k_igd_prep_queries", st);
        hipLaunchKernelGGL(k_igd_prep_queries, dim3(g), dim3(256), 0, st, qc, qs, qe, nq, v.n_chrom, kc, ks, ke, d_unsorted);
    }
    u32 h_unsorted = 1;
    if (!getenv("GTARS_IGD_ALWAYS_SORT")) {
        GT_HIP(hipMemcpyAsync(&h_unsorted, d_unsorted, sizeof(u32), hipMemcpyDeviceToHost, st));
        GT_HIP(hipStreamSynchronize(st));
    }
    if (h_unsorted) {
        // K1: order the queries by (chromosome, start)
        gtars_status s1 = device_sort_perm_ws(kc, ks, nullptr, nq, v.n_chrom + 1, perm, sort_ws, sort_ws_bytes, st);
        if (s1) return s1;
        {
            ProfScope p("k_gather2_u32", st);
            hipLaunchKernelGGL(k_gather2_u32, dim3(g), dim3(256), 0, st, ks, ke, perm, nq, ss, se);
        }
    } else {
        perm = nullptr;  // the batch is in (chromosome, start) order already
        ss = ks;
        se = ke;
    }
    {
        ProfScope p("k_igd_tile_ranges", st);
        hipLaunchKernelGGL(k_igd_chrom_segments, dim3((v.n_chrom + 1 + 63) / 64), dim3(64), 0, st, kc, perm, nq, v.n_chrom, cq_off);
        hipLaunchKernelGGL(k_igd_tile_ranges, dim3((n_tiles + 255) / 256), dim3(256), 0, st, v, tile_first, tile_cnt,
                           tile_chrom, n_tiles, ss, cq_off, ql, qh);
    }
    const size_t lds = ((size_t)(IGD_TILE + IGD_HALO) * 4 + v.n_files) * 4;
    int dev = 0, cus = 256;
    GT_HIP(hipGetDevice(&dev));
    GT_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    auto kern = binary ? k_igd_sweep<true> : k_igd_sweep<false>;
    if (lds > 48 * 1024)
        GT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int per_cu = 1;
    GT_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, SW_TPB, lds));
    if (per_cu < 1) return fail(GTARS_ERR_INTERNAL, "k_igd_sweep does not fit on a CU");
    const unsigned grid = (unsigned)std::min<u64>((u64)cus * per_cu, n_tiles);
    {
        ProfScope p(binary ? "k_igd_sweep<binary>" : "k_igd_sweep<pairwise>", st);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(SW_TPB), lds, st, v, tile_first, tile_cnt, tile_chrom, tile_carry, n_tiles, ss, se,
                           ql, qh, min_overlap, (unsigned long long *)hits);
    }
    GT_HIP(hipGetLastError());
    return GTARS_OK;
}

}  // namespace gtars
