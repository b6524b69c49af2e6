// kernels.hip -- hand-written gfx950 kernels for the gtars overlap/tokenize path.
//
// All work here is integer search / scan / compaction and is bound by the
// memory system (HBM streaming of queries and results, L2/LDS-resident index):
// there is no contraction, hence no MFMA.
//
//   k_count            K2: per-query overlap count (+any), optional min-overlap
//   k_enum_fused       K3/K4: per-query enumerate in Bits / AIList order with a
//                      single-pass chained ("decoupled look-back") prefix sum
//                      across workgroups, wave-level scan inside a workgroup,
//                      CSR offsets (u64) + hit payloads written once
//   k_fill             second pass for two-pass callers (offsets given)
//   k_scan_*           u32 counts -> u64 exclusive offsets (three-phase)
//   k_igd_*            K5: IGD per-file hit histograms (pairwise / binary)
//
// Reference semantics (file:line relative to the reference checkout):
//   overlap test        gtars-core/src/models/interval.rs:47-50
//   Bits find order     gtars-overlaprs/src/bits.rs:141-156, 433-446
//   AIList find order   gtars-overlaprs/src/ailist.rs:153-178, 238-263
//   min-overlap filter  gtars-overlaprs/src/multi_chrom_overlapper.rs:483-563
//   IGD hit rule        gtars-igd/src/igd.rs:504-540, 753-847
#include "common.h"
#include "scan.h"

namespace gtars {

// ------------------------------------------------------------------ helpers

// first i in [lo, hi) with a[i] >= key (a ascending)
__device__ __forceinline__ u32 lower_bound_u32(const u32 *__restrict__ a, u32 lo, u32 hi, u32 key) {
    while (lo < hi) {
        u32 mid = lo + ((hi - lo) >> 1);
        u32 v = a[mid];
        if (v < key)
            lo = mid + 1;
        else
            hi = mid;
    }
    return lo;
}

__device__ __forceinline__ u32 lower_bound_i32(const i32 *__restrict__ a, u32 lo, u32 hi, i32 key) {
    while (lo < hi) {
        u32 mid = lo + ((hi - lo) >> 1);
        i32 v = a[mid];
        if (v < key)
            lo = mid + 1;
        else
            hi = mid;
    }
    return lo;
}

// overlap_bp (multi_chrom_overlapper.rs:561-563)
__device__ __forceinline__ i64 overlap_bp(u32 as, u32 ae, u32 bs, u32 be) {
    u32 mn = ae < be ? ae : be;
    u32 mx = as > bs ? as : bs;
    return (i64)mn - (i64)mx;
}

// Walk the hits of one query in reference result order and call f(pos) for each.
// KIND 0 = Bits order, 1 = AIList order.  FILTER applies overlap_bp >= min_bp.
template <int KIND, bool FILTER, class F>
__device__ __forceinline__ void walk_hits(const IndexView &v, u32 c, u32 qs, u32 qe, i32 min_bp, F &&f) {
    if (c >= v.n_chrom) return;
    const u32 seg_lo = v.chrom_off[c], seg_hi = v.chrom_off[c + 1];
    if (seg_lo == seg_hi) return;
    if (KIND == 0) {
        // Bits::find: off = lower_bound(start - max_len); scan while iv.start < stop
        const u32 max_len = v.chrom_aux[c];
        const u32 key = qs >= max_len ? qs - max_len : 0u;
        u32 i = lower_bound_u32(v.starts, seg_lo, seg_hi, key);
        for (; i < seg_hi; ++i) {
            const u32 s = v.starts[i];
            if (s >= qe) break;
            const u32 e = v.ends[i];
            if (e > qs) {
                if (!FILTER || overlap_bp(qs, qe, s, e) >= (i64)min_bp) f(i);
            }
        }
    } else {
        // AIList::find: sub-list by sub-list, from the last start < end downwards
        const u32 sb = v.chrom_sub[c], se = v.chrom_sub[c + 1];
        for (u32 h = sb; h + 1 < se; ++h) {
            const u32 lo = v.sub_off[h], hi = v.sub_off[h + 1];
            u32 i = lower_bound_u32(v.starts, lo, hi, qe);  // partition_point(x < end)
            while (i > lo) {
                --i;
                const u32 e = v.ends[i];
                if (qs >= e) {
                    if (qs > v.max_ends[i]) break;
                } else {
                    if (!FILTER || overlap_bp(qs, qe, v.starts[i], e) >= (i64)min_bp) f(i);
                }
            }
        }
    }
}

template <int KIND, bool FILTER>
__device__ __forceinline__ u32 count_hits(const IndexView &v, u32 c, u32 qs, u32 qe, i32 min_bp) {
    u32 n = 0;
    walk_hits<KIND, FILTER>(v, c, qs, qe, min_bp, [&](u32) { ++n; });
    return n;
}

// ------------------------------------------------------------------ K2 count

template <int KIND, bool FILTER>
__global__ void __launch_bounds__(256)
k_count(IndexView v, const u32 *__restrict__ qc, const u32 *__restrict__ qs, const u32 *__restrict__ qe,
        u64 nq, i32 min_bp, u32 *__restrict__ counts, u8 *__restrict__ any) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 q = (u64)blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += stride) {
        const u32 n = count_hits<KIND, FILTER>(v, qc[q], qs[q], qe[q], min_bp);
        if (counts) counts[q] = n;
        if (any) any[q] = n ? 1 : 0;
    }
}

// ------------------------------------------- three-phase scan (two-pass path)

constexpr int SCAN_TPB = 256;
constexpr int SCAN_IPT = 8;
constexpr int SCAN_TILE = SCAN_TPB * SCAN_IPT;

__global__ void __launch_bounds__(SCAN_TPB)
k_scan_reduce(const u32 *__restrict__ counts, u64 n, u64 *__restrict__ partials) {
    __shared__ u64 lds[4];
    const u64 base = (u64)blockIdx.x * SCAN_TILE;
    u64 s = 0;
#pragma unroll
    for (int j = 0; j < SCAN_IPT; ++j) {
        const u64 i = base + (u64)j * SCAN_TPB + threadIdx.x;
        if (i < n) s += counts[i];
    }
    s = wave_reduce_sum_u64(s);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = lds[0] + lds[1] + lds[2] + lds[3];
}

// single block: exclusive scan of the per-tile partials in place, total -> partials[np]
__global__ void __launch_bounds__(1024) k_scan_partials(u64 *__restrict__ partials, u64 np) {
    __shared__ u64 lds[16];
    __shared__ u64 carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (u64 base = 0; base < np; base += 1024) {
        const u64 i = base + threadIdx.x;
        u64 x = i < np ? partials[i] : 0, inc = x;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            u64 y = __shfl_up(inc, d, 64);
            if (lane >= d) inc += y;
        }
        if (lane == 63) lds[wave] = inc;
        __syncthreads();
        u64 wbase = 0, tot = 0;
        for (int w = 0; w < 16; ++w) {
            if (w < wave) wbase += lds[w];
            tot += lds[w];
        }
        const u64 carry = carry_s;
        if (i < np) partials[i] = carry + wbase + inc - x;
        __syncthreads();
        if (threadIdx.x == 0) carry_s = carry + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) partials[np] = carry_s;
}

__global__ void __launch_bounds__(SCAN_TPB)
k_scan_apply(const u32 *__restrict__ counts, u64 n, const u64 *__restrict__ partials, u64 np,
             u64 *__restrict__ offsets) {
    __shared__ u32 lds[4];
    const u64 base = (u64)blockIdx.x * SCAN_TILE + (u64)threadIdx.x * SCAN_IPT;
    u32 c[SCAN_IPT];
    u32 s = 0;
#pragma unroll
    for (int j = 0; j < SCAN_IPT; ++j) {
        c[j] = (base + j < n) ? counts[base + j] : 0;
        s += c[j];
    }
    u32 total;
    u32 ex = block_exclusive_scan<SCAN_TPB>(s, lds, total);
    u64 run = partials[blockIdx.x] + ex;
#pragma unroll
    for (int j = 0; j < SCAN_IPT; ++j) {
        if (base + j < n) offsets[base + j] = run;
        run += c[j];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) offsets[n] = partials[np];
}

size_t scan_ws_bytes(u64 n) {
    const u64 np = (n + SCAN_TILE - 1) / SCAN_TILE;
    return (size_t)(np + 1) * sizeof(u64);
}

gtars_status launch_scan_u32_to_u64(const u32 *counts, u64 n, u64 *offsets, void *ws, size_t ws_bytes,
                                    hipStream_t st) {
    const u64 np = (n + SCAN_TILE - 1) / SCAN_TILE;
    if (ws_bytes < (np + 1) * sizeof(u64)) return fail(GTARS_ERR_INTERNAL, "scan workspace too small");
    u64 *partials = (u64 *)ws;
    if (n == 0) {
        GT_HIP(hipMemsetAsync(offsets, 0, sizeof(u64), st));
        return GTARS_OK;
    }
    {
        ProfScope p("k_scan_reduce", st);
        hipLaunchKernelGGL(k_scan_reduce, dim3((unsigned)np), dim3(SCAN_TPB), 0, st, counts, n, partials);
    }
    {
        ProfScope p("k_scan_partials", st);
        hipLaunchKernelGGL(k_scan_partials, dim3(1), dim3(1024), 0, st, partials, np);
    }
    {
        ProfScope p("k_scan_apply", st);
        hipLaunchKernelGGL(k_scan_apply, dim3((unsigned)np), dim3(SCAN_TPB), 0, st, counts, n, partials, np,
                           offsets);
    }
    GT_HIP(hipGetLastError());
    return GTARS_OK;
}

// ------------------------------------------------------------------ fill pass

template <int KIND, bool FILTER>
__global__ void __launch_bounds__(256)
k_fill(IndexView v, const u32 *__restrict__ qc, const u32 *__restrict__ qs, const u32 *__restrict__ qe,
       u64 nq, i32 min_bp, const u64 *__restrict__ offsets, u32 *__restrict__ ovals,
       u32 *__restrict__ ostarts, u32 *__restrict__ oends) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 q = (u64)blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += stride) {
        u64 o = offsets[q];
        walk_hits<KIND, FILTER>(v, qc[q], qs[q], qe[q], min_bp, [&](u32 i) {
            if (ovals) ovals[o] = v.vals[i];
            if (ostarts) ostarts[o] = v.starts[i];
            if (oends) oends[o] = v.ends[i];
            ++o;
        });
    }
}

// ------------------------------------------------- K3/K4 fused enumerate+scan
//
// Persistent workgroups take tiles of ENUM_TILE consecutive queries through a
// ticket counter (so a tile only ever waits on tiles that have already
// started -> forward progress without any dispatch-order assumption).  Per
// tile: every thread counts the hits of ENUM_QPT consecutive queries, the
// workgroup scans the counts (wave shuffles + 4 LDS words), wave 0 resolves the
// tile's global base by a decoupled look-back over 8-byte {status,value}
// granules (one relaxed agent-scope 64-bit atomic each: flag and payload
// travel in the same granule, so no fence is needed), then offsets and hit
// payloads are written straight to their final place.

constexpr int ENUM_TPB = 256;
constexpr int ENUM_QPT = 4;
constexpr int ENUM_TILE = ENUM_TPB * ENUM_QPT;

__global__ void k_gather_hits(const u32 *__restrict__ sv, const u32 *__restrict__ ss, const u32 *__restrict__ se,
                              const u32 *__restrict__ pos, u64 n, u32 *__restrict__ vals, u32 *__restrict__ starts,
                              u32 *__restrict__ ends) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u32 p = pos[i];
    if (vals) vals[i] = sv[p];
    if (starts) starts[i] = ss[p];
    if (ends) ends[i] = se[p];
}

gtars_status launch_gather_hits(const IndexView &v, const u32 *pos, u64 n, u32 *vals, u32 *starts, u32 *ends, hipStream_t st) {
    if (n == 0) return GTARS_OK;
    ProfScope p("k_gather_hits", st);
    hipLaunchKernelGGL(k_gather_hits, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, v.vals, v.starts, v.ends, pos, n, vals,
                       starts, ends);
    GT_HIP(hipGetLastError());
    return GTARS_OK;
}

u32 enumerate_fused_tile_queries() { return ENUM_TILE; }
size_t enumerate_fused_ws_bytes(u64 nq) {
    return scan_ws_bytes_for_tiles((nq + ENUM_TILE - 1) / ENUM_TILE);
}

template <int KIND, bool FILTER>
__global__ void __launch_bounds__(ENUM_TPB)
k_enum_fused(IndexView v, const u32 *__restrict__ qc, const u32 *__restrict__ qs,
             const u32 *__restrict__ qe, u64 nq, i32 min_bp, u64 *__restrict__ offsets,
             u32 *__restrict__ ovals, u32 *__restrict__ ostarts, u32 *__restrict__ oends, u64 cap,
             ScanWs *ws) {
    __shared__ u32 s_tile;
    __shared__ u64 s_prefix;
    __shared__ u32 s_scan[4];
    const u32 num_tiles = (u32)((nq + ENUM_TILE - 1) / ENUM_TILE);
    const int lane = threadIdx.x & 63;
    const bool vec_ok = ((((uintptr_t)qc) | ((uintptr_t)qs) | ((uintptr_t)qe)) & 15u) == 0;

    for (;;) {
        if (threadIdx.x == 0) s_tile = atomicAdd(&ws->ticket, 1u);
        __syncthreads();
        const u32 tile = s_tile;
        if (tile >= num_tiles) break;

        const u64 q0 = (u64)tile * ENUM_TILE + (u64)threadIdx.x * ENUM_QPT;
        u32 c[ENUM_QPT], s[ENUM_QPT], e[ENUM_QPT], cnt[ENUM_QPT];
        if (vec_ok && q0 + ENUM_QPT <= nq) {
            const uint4 c4 = *reinterpret_cast<const uint4 *>(qc + q0);
            const uint4 s4 = *reinterpret_cast<const uint4 *>(qs + q0);
            const uint4 e4 = *reinterpret_cast<const uint4 *>(qe + q0);
            c[0] = c4.x; c[1] = c4.y; c[2] = c4.z; c[3] = c4.w;
            s[0] = s4.x; s[1] = s4.y; s[2] = s4.z; s[3] = s4.w;
            e[0] = e4.x; e[1] = e4.y; e[2] = e4.z; e[3] = e4.w;
        } else {
#pragma unroll
            for (int j = 0; j < ENUM_QPT; ++j) {
                const bool ok = q0 + j < nq;
                c[j] = ok ? qc[q0 + j] : GTARS_UNKNOWN_CHROM;
                s[j] = ok ? qs[q0 + j] : 0;
                e[j] = ok ? qe[q0 + j] : 0;
            }
        }
        u32 tsum = 0;
#pragma unroll
        for (int j = 0; j < ENUM_QPT; ++j) {
            cnt[j] = count_hits<KIND, FILTER>(v, c[j], s[j], e[j], min_bp);
            tsum += cnt[j];
        }
        u32 block_total;
        const u32 excl = block_exclusive_scan<ENUM_TPB>(tsum, s_scan, block_total);

        if (threadIdx.x < 64) {
            const u64 p = lookback(ws->state, tile, (u64)block_total, lane, &ws->err);
            if (lane == 0) s_prefix = p;
        }
        __syncthreads();
        u64 run = s_prefix + excl;
        if (tile == num_tiles - 1 && threadIdx.x == ENUM_TPB - 1) {
            // last thread of the last tile holds the grand total
            const u64 tot = s_prefix + (u64)block_total;
            offsets[nq] = tot;
            ws->total = tot;
        }
#pragma unroll
        for (int j = 0; j < ENUM_QPT; ++j) {
            if (q0 + j < nq) offsets[q0 + j] = run;
            if (cnt[j]) {
                u64 o = run;
                walk_hits<KIND, FILTER>(v, c[j], s[j], e[j], min_bp, [&](u32 i) {
                    if (o < cap) {
                        if (ovals) ovals[o] = v.vals[i];
                        if (ostarts) ostarts[o] = v.starts[i];
                        if (oends) oends[o] = v.ends[i];
                    }
                    ++o;
                });
            }
            run += cnt[j];
        }
        __syncthreads();  // s_tile / s_prefix reuse
    }
}

template <int KIND, bool FILTER>
static void launch_enum_t(const IndexView &v, const u32 *qc, const u32 *qs, const u32 *qe, u64 nq,
                          i32 min_bp, const EnumOut &out, ScanWs *ws, unsigned grid, hipStream_t st) {
    hipLaunchKernelGGL((k_enum_fused<KIND, FILTER>), dim3(grid), dim3(ENUM_TPB), 0, st, v, qc, qs, qe, nq,
                       min_bp, out.offsets, out.vals, out.starts, out.ends, out.capacity, ws);
}

static unsigned persistent_grid(u64 tiles, int blocks_per_cu) {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, dev) == hipSuccess) cus = p.multiProcessorCount;
    }
    u64 g = (u64)cus * blocks_per_cu;
    if (g > tiles) g = tiles;
    if (g == 0) g = 1;
    return (unsigned)g;
}

gtars_status launch_enumerate_fused(const IndexView &v, int kind, const u32 *qc, const u32 *qs,
                                    const u32 *qe, u64 nq, int has_min, i32 min_overlap,
                                    const EnumOut &out, void *scan_ws, size_t scan_ws_bytes_,
                                    hipStream_t st) {
    if (nq == 0) {
        GT_HIP(hipMemsetAsync(out.offsets, 0, sizeof(u64), st));
        GT_HIP(hipMemsetAsync(scan_ws, 0, sizeof(ScanWs), st));
        return GTARS_OK;
    }
    const u64 tiles = (nq + ENUM_TILE - 1) / ENUM_TILE;
    if (tiles > 0xFFFFFFF0ull) return fail(GTARS_ERR_INVALID_ARG, "query batch too large for one launch");
    const size_t need = enumerate_fused_ws_bytes(nq);
    if (scan_ws_bytes_ < need) return fail(GTARS_ERR_INTERNAL, "fused scan workspace too small");
    GT_HIP(hipMemsetAsync(scan_ws, 0, need, st));
    const bool filter = has_min && min_overlap > 1;
    const i32 min_bp = has_min ? min_overlap : 0;
    const unsigned grid = persistent_grid(tiles, 8);
    ScanWs *ws = (ScanWs *)scan_ws;
    {
        ProfScope p(kind == GTARS_KIND_BITS ? "k_enum_fused<bits>" : "k_enum_fused<ailist>", st);
        if (kind == GTARS_KIND_BITS) {
            if (filter)
                launch_enum_t<0, true>(v, qc, qs, qe, nq, min_bp, out, ws, grid, st);
            else
                launch_enum_t<0, false>(v, qc, qs, qe, nq, min_bp, out, ws, grid, st);
        } else {
            if (filter)
                launch_enum_t<1, true>(v, qc, qs, qe, nq, min_bp, out, ws, grid, st);
            else
                launch_enum_t<1, false>(v, qc, qs, qe, nq, min_bp, out, ws, grid, st);
        }
    }
    GT_HIP(hipGetLastError());
    return GTARS_OK;
}

static unsigned stream_grid(u64 n, int tpb) {
    u64 g = (n + tpb - 1) / tpb;
    const u64 cap = 256ull * 16;
    if (g > cap) g = cap;
    if (g == 0) g = 1;
    return (unsigned)g;
}

// Bits::count (bits.rs:337-344): len - #{ends < start + 1} - #{starts >= stop} by two runs of the reference's
// bsearch_seq (bits.rs:304-322) over the separately sorted starts / ends.  `start + 1` wraps and the final
// subtraction wraps, as release-mode Rust does (the result differs from find().len() only for zero-length
// or inverted queries).
__device__ __forceinline__ u64 bsearch_seq_dev(u32 key, const u32 *__restrict__ elems, u64 n) {
    if (n == 0 || elems[0] >= key) return 0;
    if (elems[n - 1] < key) return n;
    u64 cursor = 0, length = n;
    while (length > 1) {
        const u64 half = length >> 1;
        length -= half;
        cursor += (elems[cursor + half - 1] < key) ? half : 0;
    }
    return cursor;
}

__global__ void k_bits_count(IndexView v, const u32 *__restrict__ ends_sorted, const u32 *__restrict__ qc,
                             const u32 *__restrict__ qs, const u32 *__restrict__ qe, u64 nq, u64 *__restrict__ out) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < nq; i += (u64)gridDim.x * blockDim.x) {
        const u32 c = qc[i];
        u64 r = 0;
        if (c < v.n_chrom) {
            const u32 lo = v.chrom_off[c], hi = v.chrom_off[c + 1];
            const u64 len = hi - lo;
            if (len) {  // a chromosome without intervals has no index (MultiChromOverlapper: count 0)
                const u64 first = bsearch_seq_dev(qs[i] + 1u, ends_sorted + lo, len);
                const u64 last = bsearch_seq_dev(qe[i], v.starts + lo, len);
                r = len - first - (len - last);
            }
        }
        out[i] = r;
    }
}

gtars_status launch_bits_count(const IndexView &v, const u32 *ends_sorted, const u32 *qc, const u32 *qs, const u32 *qe,
                               u64 nq, u64 *out, hipStream_t st) {
    if (nq == 0) return GTARS_OK;
    ProfScope p("k_bits_count", st);
    hipLaunchKernelGGL(k_bits_count, dim3(stream_grid(nq, 256)), dim3(256), 0, st, v, ends_sorted, qc, qs, qe, nq, out);
    GT_HIP(hipGetLastError());
    return GTARS_OK;
}

gtars_status launch_count(const IndexView &v, int kind, const u32 *qc, const u32 *qs, const u32 *qe,
                          u64 nq, int has_min, i32 min_overlap, u32 *counts, u8 *any, hipStream_t st) {
    if (nq == 0) return GTARS_OK;
    const bool filter = has_min && min_overlap > 1;
    const i32 min_bp = has_min ? min_overlap : 0;
    const unsigned grid = stream_grid(nq, 256);
    ProfScope p("k_count", st);
    if (kind == GTARS_KIND_BITS) {
        if (filter)
            hipLaunchKernelGGL((k_count<0, true>), dim3(grid), dim3(256), 0, st, v, qc, qs, qe, nq, min_bp, counts, any);
        else
            hipLaunchKernelGGL((k_count<0, false>), dim3(grid), dim3(256), 0, st, v, qc, qs, qe, nq, min_bp, counts, any);
    } else {
        if (filter)
            hipLaunchKernelGGL((k_count<1, true>), dim3(grid), dim3(256), 0, st, v, qc, qs, qe, nq, min_bp, counts, any);
        else
            hipLaunchKernelGGL((k_count<1, false>), dim3(grid), dim3(256), 0, st, v, qc, qs, qe, nq, min_bp, counts, any);
    }
    GT_HIP(hipGetLastError());
    return GTARS_OK;
}

gtars_status launch_fill(const IndexView &v, int kind, const u32 *qc, const u32 *qs, const u32 *qe, u64 nq,
                         int has_min, i32 min_overlap, const u64 *offsets, u32 *vals, u32 *starts,
                         u32 *ends, hipStream_t st) {
    if (nq == 0) return GTARS_OK;
    const bool filter = has_min && min_overlap > 1;
    const i32 min_bp = has_min ? min_overlap : 0;
    const unsigned grid = stream_grid(nq, 256);
    ProfScope p("k_fill", st);
    if (kind == GTARS_KIND_BITS) {
        if (filter)
            hipLaunchKernelGGL((k_fill<0, true>), dim3(grid), dim3(256), 0, st, v, qc, qs, qe, nq, min_bp, offsets, vals, starts, ends);
        else
            hipLaunchKernelGGL((k_fill<0, false>), dim3(grid), dim3(256), 0, st, v, qc, qs, qe, nq, min_bp, offsets, vals, starts, ends);
    } else {
        if (filter)
            hipLaunchKernelGGL((k_fill<1, true>), dim3(grid), dim3(256), 0, st, v, qc, qs, qe, nq, min_bp, offsets, vals, starts, ends);
        else
            hipLaunchKernelGGL((k_fill<1, false>), dim3(grid), dim3(256), 0, st, v, qc, qs, qe, nq, min_bp, offsets, vals, starts, ends);
    }
    GT_HIP(hipGetLastError());
    return GTARS_OK;
}

// ------------------------------------------------ per-segment sort + unique
// IndexedRegionSet::find_overlaps sorts and de-duplicates each query's source
// indices (indexed_region_set.rs:258-260).  One thread sorts its own segment in
// place: insertion sort for the usual handful of hits, heap sort (n log n, no
// scratch memory) for the long segments of dense databases.
__global__ void __launch_bounds__(256)
k_sort_unique_segments(u32 *__restrict__ vals, const u64 *__restrict__ offsets, u64 nq,
                       u32 *__restrict__ new_counts) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 q = (u64)blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += stride) {
        const u64 lo = offsets[q], hi = offsets[q + 1];
        u32 *a = vals + lo;
        const u64 n = hi - lo;
        if (n <= 24) {
            for (u64 i = 1; i < n; ++i) {
                const u32 x = a[i];
                u64 j = i;
                while (j > 0 && a[j - 1] > x) {
                    a[j] = a[j - 1];
                    --j;
                }
                a[j] = x;
            }
        } else {
            auto sift = [&](u64 root, u64 end) {  // max-heap on a[0, end)
                const u32 x = a[root];
                for (;;) {
                    u64 ch = 2 * root + 1;
                    if (ch >= end) break;
                    if (ch + 1 < end && a[ch + 1] > a[ch]) ++ch;
                    if (a[ch] <= x) break;
                    a[root] = a[ch];
                    root = ch;
                }
                a[root] = x;
            };
            for (u64 i = n / 2; i > 0; --i) sift(i - 1, n);
            for (u64 end = n - 1; end > 0; --end) {
                const u32 t = a[0];
                a[0] = a[end];
                a[end] = t;
                sift(0, end);
            }
        }
        u64 k = 0;
        for (u64 i = 0; i < n; ++i)
            if (i == 0 || a[i] != a[i - 1]) a[k++] = a[i];
        new_counts[q] = (u32)k;
    }
}

gtars_status launch_sort_unique_segments(u32 *vals, const u64 *offsets, u64 nq, u32 *new_counts,
                                         hipStream_t st) {
    if (nq == 0) return GTARS_OK;
    ProfScope p("k_sort_unique_segments", st);
    hipLaunchKernelGGL(k_sort_unique_segments, dim3(stream_grid(nq, 256)), dim3(256), 0, st, vals, offsets,
                       nq, new_counts);
    GT_HIP(hipGetLastError());
    return GTARS_OK;
}

// AIList::find order for a nested AIList index from its flat companion's hits (round 5).  The companion's LDS tokenizer has
// written every query's hits as STORED POSITIONS of the companion, in Bits order; AIList::find emits sub-list after sub-list,
// each walked from its last candidate down (ailist.rs:153-178, 238-263).  key_by_pos[p] = the hit's MIRRORED position in the
// AIList's stored order (sub-list start + sub-list length - 1 - index in the sub-list): ascending keys = sub-lists ascending,
// positions descending -- exactly that order -- so a query's hits are sorted by key in place (insertion sort: most queries have
// one or two hits; heap sort beyond 32) and then replaced by their values (val_by_key).  One thread per query.  Queries whose ids
// lie (partly) beyond `capacity` are left alone: the launch ends in GTARS_ERR_CAPACITY anyway.
__global__ void __launch_bounds__(256)
k_ailist_reorder(u32 *__restrict__ ids, const u64 *__restrict__ offsets, u64 nq, u64 capacity, const u32 *__restrict__ key_by_pos,
                 const u32 *__restrict__ val_by_key) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 q = (u64)blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += stride) {
        const u64 lo = offsets[q], hi = offsets[q + 1];
        if (hi == lo || hi > capacity) continue;
        u32 *a = ids + lo;
        const u64 n = hi - lo;
        if (n == 1) {
            a[0] = val_by_key[key_by_pos[a[0]]];
            continue;
        }
        for (u64 i = 0; i < n; ++i) a[i] = key_by_pos[a[i]];
        if (n <= 32) {
            for (u64 i = 1; i < n; ++i) {
                const u32 x = a[i];
                u64 j = i;
                while (j > 0 && a[j - 1] > x) {
                    a[j] = a[j - 1];
                    --j;
                }
                a[j] = x;
            }
        } else {
            auto sift = [&](u64 root, u64 end) {  // max-heap on a[0, end)
                const u32 x = a[root];
                for (;;) {
                    u64 ch = 2 * root + 1;
                    if (ch >= end) break;
                    if (ch + 1 < end && a[ch + 1] > a[ch]) ++ch;
                    if (a[ch] <= x) break;
                    a[root] = a[ch];
                    root = ch;
                }
                a[root] = x;
            };
            for (u64 i = n / 2; i > 0; --i) sift(i - 1, n);
            for (u64 end = n - 1; end > 0; --end) {
                const u32 t = a[0];
                a[0] = a[end];
                a[end] = t;
                sift(0, end);
            }
        }
        for (u64 i = 0; i < n; ++i) a[i] = val_by_key[a[i]];
    }
}

gtars_status launch_ailist_reorder(u32 *ids, const u64 *offsets, u64 nq, u64 capacity, const u32 *key_by_pos, const u32 *val_by_key,
                                   hipStream_t st) {
    if (nq == 0) return GTARS_OK;
    ProfScope p("k_ailist_reorder", st);
    hipLaunchKernelGGL(k_ailist_reorder, dim3(stream_grid(nq, 256)), dim3(256), 0, st, ids, offsets, nq, capacity, key_by_pos, val_by_key);
    GT_HIP(hipGetLastError());
    return GTARS_OK;
}

// ---------------------------------------------------------------- K5: IGD
//
// The reference walks nbp-sized tiles (igd.rs:753-847); for min_overlap >= 1
// that walk visits every stored interval of the chromosome that satisfies
//     min(r.end, qe) - max(r.start, qs) >= min_overlap
// exactly once (the first tile scans all of its records, later tiles only
// records that start inside them).  The device therefore keeps each interval
// once, sorted by start, and a query scans positions
// [lower_bound(qs - max_len), lower_bound(qe)).
// Query validation follows Igd::count_overlaps (igd.rs:514-517).

// min_overlap <= 0 (igd.rs accepts it): the walk then also admits records that do not overlap the query, and which ones
// depends on the tiling -- the first tile n1 = qs / nbp scans every record PRESENT in it (start / nbp <= n1 <= (end - 1) / nbp),
// a later tile j <= n2 = (qe - 1) / nbp only the records that START in it (igd.rs:772-846), each once, all with start < qe
// and min(end, qe) - max(start, qs) >= min_overlap.  For min_overlap >= 1 every overlapping record passes that tile test,
// which is why the flat formula suffices there.  nbp = 16384 (Igd::new, igd.rs:80-83).
constexpr i32 IGD_NBP_COUNT = 16384;
struct IgdQual {
    i32 qs, qe, mo, n1, n2;
    bool tiled;
    __device__ __forceinline__ bool operator()(i32 s, i32 e) const {
        const i32 ov = (e < qe ? e : qe) - (s > qs ? s : qs);
        if (ov < mo) return false;
        if (!tiled) return true;
        const i32 ts = s / IGD_NBP_COUNT;
        return ts <= n1 ? (e - 1) / IGD_NBP_COUNT >= n1 : ts <= n2;
    }
};
// -> false: no record can qualify (unknown / empty chromosome, or -- tiled -- the query starts past the contig's last tile);
// lo = where the scan starts
// pm (may be null): IgdTiles::pm, the prefix maximum of the ends over the chromosome's records -- with it (and min_overlap >= 1)
// the scan starts at the first record whose prefix-max end is > q_start, i.e. at the first record that CAN overlap, instead of
// at lower_bound(q_start - the chromosome's longest record): one multi-megabase record in a database would otherwise make
// every query of its chromosome walk through everything that starts within that distance before it.
__device__ __forceinline__ bool igd_walk_setup(const IgdView &v, u32 c, i32 qs, i32 qe, i32 min_overlap, IgdQual &q, u32 &lo,
                                               u32 &seg_hi, const i32 *__restrict__ pm = nullptr) {
    if (c >= v.n_chrom) return false;
    const u32 seg_lo = v.chrom_off[c];
    seg_hi = v.chrom_off[c + 1];
    if (seg_lo == seg_hi) return false;
    q.qs = qs;
    q.qe = qe;
    q.mo = min_overlap;
    q.tiled = min_overlap < 1;
    q.n1 = q.n2 = 0;
    i64 key = (i64)qs - (i64)v.chrom_maxlen[c];
    if (q.tiled) {
        const i32 nt = v.chrom_ntiles[c];
        q.n1 = qs / IGD_NBP_COUNT;
        if (q.n1 >= nt) return false;
        q.n2 = min((qe - 1) / IGD_NBP_COUNT, nt - 1);
        key += (i64)min_overlap;  // a record left of the query may end up to -min_overlap before it
    }
    if (pm && !q.tiled) {
        // first record with pm > qs (pm ascends inside a chromosome; a qualifying record ends at or after qs + min_overlap > qs)
        lo = lower_bound_i32(pm, seg_lo, seg_hi, qs < 0x7FFFFFFF ? qs + 1 : qs);
        return true;
    }
    lo = lower_bound_i32(v.starts, seg_lo, seg_hi, key > 0 ? (i32)key : 0);  // starts are >= 0
    return true;
}

template <class F>
__device__ __forceinline__ void igd_walk(const IgdView &v, u32 c, i32 qs, i32 qe, i32 min_overlap, F &&f, const i32 *__restrict__ pm = nullptr) {
    IgdQual q;
    u32 i, seg_hi;
    if (!igd_walk_setup(v, c, qs, qe, min_overlap, q, i, seg_hi, pm)) return;
    for (; i < seg_hi; ++i) {
        const i32 s = v.starts[i];
        if (s >= qe) break;
        if (q(s, v.ends[i])) f(i);
    }
}

constexpr int IGD_LDS_FILES = 8192;  // 32 KiB of u32 bins per workgroup

// BINARY with pme_file (min_overlap == 1): a hit is its file's first for the query iff no earlier record of the file
// ends after the query's start (see k_igd_sweep, igd_sweep.hip) -- no look at the scanned prefix at all.
template <bool BINARY, bool USE_LDS>
__global__ void __launch_bounds__(256)
k_igd_count(IgdView v, const i32 *__restrict__ pme_file, const i32 *__restrict__ pm, const u32 *__restrict__ qc, const u32 *__restrict__ qs,
            const u32 *__restrict__ qe, u64 nq, i32 min_overlap, unsigned long long *__restrict__ hits) {
    __shared__ u32 bins[USE_LDS ? IGD_LDS_FILES : 1];
    if (USE_LDS) {
        for (u32 i = threadIdx.x; i < v.n_files; i += blockDim.x) bins[i] = 0;
        __syncthreads();
    }
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 q = (u64)blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += stride) {
        i32 s = (i32)qs[q], e = (i32)qe[q];  // `as i32` (igd.rs:549-550)
        if (s >= e || e <= 0) continue;      // igd.rs:514-516
        if (s < 0) s = 0;                    // igd.rs:517
        const u32 c = qc[q];
        if (!BINARY) {
            igd_walk(v, c, s, e, min_overlap, [&](u32 i) {
                u32 f = v.files[i];
                if (f >> 31) {  // pieces view: a continuation piece counts only if it holds the query's start (IgdView::pieces)
                    if (v.starts[i] > s) return;
                    f &= IGD_FILE_MASK;
                }
                if (USE_LDS)
                    atomicAdd(&bins[f], 1u);
                else
                    atomicAdd(&hits[f], 1ull);
            }, pm);
        } else if (pme_file) {
            igd_walk(v, c, s, e, min_overlap, [&](u32 i) {
                if (pme_file[i] > s) return;
                u32 f = v.files[i];
                if (f >> 31) {
                    if (v.starts[i] > s) return;
                    f &= IGD_FILE_MASK;
                }
                if (USE_LDS)
                    atomicAdd(&bins[f], 1u);
                else
                    atomicAdd(&hits[f], 1ull);
            }, pm);
        } else {
            // count file f once per query: only at the first hit (in scan order) that belongs to f.
            // "Is there an earlier hit of the same file" looks at the file ids of the (short) scanned
            // prefix first and tests the overlap only for records of the same file.
            IgdQual qual;
            u32 lo = 0, seg_hi0 = 0;
            if (!igd_walk_setup(v, c, s, e, min_overlap, qual, lo, seg_hi0, pm)) continue;
            unsigned long long seen0 = 0, seen1 = 0;  // 128-bit filter of files already credited
            igd_walk(v, c, s, e, min_overlap, [&](u32 i) {
                const u32 f = v.files[i];
                const u32 hb = (f * 2654435761u) >> 25;
                const unsigned long long bit = 1ull << (hb & 63u);
                unsigned long long &word = (hb & 64u) ? seen1 : seen0;
                const bool maybe_dup = (word & bit) != 0;
                word |= bit;
                bool first = true;
                for (u32 k = lo; maybe_dup && k < i; ++k) {
                    if (v.files[k] == f && qual(v.starts[k], v.ends[k])) {
                        first = false;
                        break;
                    }
                }
                if (first) {
                    if (USE_LDS)
                        atomicAdd(&bins[f], 1u);
                    else
                        atomicAdd(&hits[f], 1ull);
                }
            }, pm);
        }
    }
    if (USE_LDS) {
        __syncthreads();
        for (u32 i = threadIdx.x; i < v.n_files; i += blockDim.x) {
            const u32 b = bins[i];
            if (b) atomicAdd(&hits[i], (unsigned long long)b);
        }
    }
}

gtars_status launch_igd_count(const IgdView &v, const i32 *pme_file, const u32 *qc, const u32 *qs, const u32 *qe, u64 nq,
                              i32 min_overlap, int binary, u64 *hits, hipStream_t st, const i32 *pm) {
    if (!binary || min_overlap != 1) pme_file = nullptr;
    if (min_overlap < 1 || cfg_get("GTARS_IGD_NO_PM_START")) pm = nullptr;  // (the environment switch: tests / A-B)
    if (v.pieces && (min_overlap != 1 || (binary && !pme_file)))
        return fail(GTARS_ERR_INTERNAL, "IGD count: a pieces view serves min_overlap == 1 (binary: the pme_file form) only");
    if (min_overlap < 1 && !v.chrom_ntiles) return fail(GTARS_ERR_INTERNAL, "IGD count with min_overlap < 1 needs the contigs' tile counts");
    GT_HIP(hipMemsetAsync(hits, 0, sizeof(u64) * (v.n_files ? v.n_files : 1), st));
    if (nq == 0 || v.n == 0) return GTARS_OK;
    // u32 LDS bins cannot overflow: a workgroup adds at most (its queries x hits);
    // cap the queries per workgroup so that even pathological inputs stay < 2^32
    // only matters in theory -- fall back to global atomics for huge batches.
    const bool lds = v.n_files <= IGD_LDS_FILES && nq < (1ull << 31);
    const unsigned grid = stream_grid(nq, 256);
    unsigned long long *h = (unsigned long long *)hits;
    ProfScope p(binary ? "k_igd_count<binary>" : "k_igd_count<pairwise>", st);
    if (binary) {
        if (lds)
            hipLaunchKernelGGL((k_igd_count<true, true>), dim3(grid), dim3(256), 0, st, v, pme_file, pm, qc, qs, qe, nq, min_overlap, h);
        else
            hipLaunchKernelGGL((k_igd_count<true, false>), dim3(grid), dim3(256), 0, st, v, pme_file, pm, qc, qs, qe, nq, min_overlap, h);
    } else {
        if (lds)
            hipLaunchKernelGGL((k_igd_count<false, true>), dim3(grid), dim3(256), 0, st, v, pme_file, pm, qc, qs, qe, nq, min_overlap, h);
        else
            hipLaunchKernelGGL((k_igd_count<false, false>), dim3(grid), dim3(256), 0, st, v, pme_file, pm, qc, qs, qe, nq, min_overlap, h);
    }
    GT_HIP(hipGetLastError());
    return GTARS_OK;
}

// Igd::count_overlaps_per_query / find_overlaps_regionset do NOT validate or
// clamp the query (igd.rs:645-722); hits are de-duplicated by record.value
// (first occurrence in walk order wins).  Walk order (igd.rs:772-846): the
// query's first tile n1 = qs/nbp holds every record starting before
// (n1+1)*nbp, scanned from the highest position down; then each later tile's
// own records, again downwards.
constexpr i32 IGD_NBP = 16384;

template <class F>
__device__ __forceinline__ void igd_walk_ref_order(const IgdView &v, u32 c, i32 qs, i32 qe, i32 min_overlap,
                                                   F &&f) {
    if (c >= v.n_chrom) return;
    const u32 seg_lo = v.chrom_off[c], seg_hi = v.chrom_off[c + 1];
    if (seg_lo == seg_hi) return;
    // A negative start (a u32 start >= 2^31 `as i32`, igd.rs:709) is not validated here: n1 = qs / nbp truncates towards zero, so
    // -nbp < qs < 0 walks from tile 0 with the raw start (the same hits as a start of 0); from -nbp down the reference indexes
    // tiles[n1] with n1 < 0 and panics -- no result defined, none returned.
    if (qs <= -IGD_NBP) return;
    const i32 n1 = qs / IGD_NBP;
    const bool tiled = min_overlap < 1;  // see IgdQual: the first tile then only serves the records PRESENT in it
    if (tiled && n1 >= v.chrom_ntiles[c]) return;
    i64 key = (i64)qs - (i64)v.chrom_maxlen[c] + (tiled ? (i64)min_overlap : 0);
    // min_overlap >= 1: nothing before the first record whose prefix-max end is > q_start can qualify (same records reported,
    // in the same order; one long record in the chromosome no longer makes every query walk back by its length)
    const u32 lo = (!tiled && v.pm) ? lower_bound_i32(v.pm, seg_lo, seg_hi, qs < 0x7FFFFFFF ? qs + 1 : qs)
                                    : lower_bound_i32(v.starts, seg_lo, seg_hi, key > 0 ? (i32)key : 0);
    const u32 hi = lower_bound_i32(v.starts, lo, seg_hi, qe);
    if (lo >= hi) return;
    // group boundaries: first group = starts < (n1+1)*nbp
    u32 g_lo = lo;
    i64 bd = (i64)IGD_NBP * ((i64)n1 + 1);
    bool first_group = true;
    while (g_lo < hi) {
        u32 g_hi = g_lo;
        // end of this group: first position with start >= bd
        {
            u32 a = g_lo, b = hi;
            while (a < b) {
                u32 m = a + ((b - a) >> 1);
                if ((i64)v.starts[m] < bd) a = m + 1; else b = m;
            }
            g_hi = a;
        }
        for (u32 i = g_hi; i > g_lo;) {
            --i;
            const i32 s = v.starts[i], e = v.ends[i];
            const i32 ov = (e < qe ? e : qe) - (s > qs ? s : qs);
            if (ov >= min_overlap && !(tiled && first_group && (e - 1) / IGD_NBP < n1)) f(i);
        }
        g_lo = g_hi;
        first_group = false;
        if (g_lo < hi) {
            // jump to the tile that holds the next record
            const i64 t = (i64)v.starts[g_lo] / IGD_NBP;
            bd = (t + 1) * (i64)IGD_NBP;
        }
    }
}

// de-duplicated walk: f(i) only for the first record (walk order) of each value.  UNIQUE: the database's values
// are all distinct (Igd::from_single_region_set: value = source index, the documented precondition of these
// queries, igd.rs:640-641), so every hit is a first occurrence; otherwise every hit re-walks the hits before it.
template <bool UNIQUE, class F>
__device__ __forceinline__ void igd_walk_unique(const IgdView &v, u32 c, i32 qs, i32 qe, i32 min_overlap,
                                                F &&f) {
    if (UNIQUE) {
        igd_walk_ref_order(v, c, qs, qe, min_overlap, f);
        return;
    }
    u32 ord = 0;
    igd_walk_ref_order(v, c, qs, qe, min_overlap, [&](u32 i) {
        const i32 val = v.values[i];
        bool first = true;
        u32 k_ord = 0;
        igd_walk_ref_order(v, c, qs, qe, min_overlap, [&](u32 k) {
            if (k_ord < ord && v.values[k] == val) first = false;
            ++k_ord;
        });
        if (first) f(i);
        ++ord;
    });
}

template <bool UNIQUE>
__global__ void __launch_bounds__(256)
k_igd_count_per_query(IgdView v, const u32 *__restrict__ qc, const u32 *__restrict__ qs,
                      const u32 *__restrict__ qe, u64 nq, i32 min_overlap, u32 *__restrict__ counts) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 q = (u64)blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += stride) {
        u32 n = 0;
        igd_walk_unique<UNIQUE>(v, qc[q], (i32)qs[q], (i32)qe[q], min_overlap, [&](u32) { ++n; });
        counts[q] = n;
    }
}

template <bool UNIQUE>
__global__ void __launch_bounds__(256)
k_igd_fill_pairs(IgdView v, const u32 *__restrict__ qc, const u32 *__restrict__ qs,
                 const u32 *__restrict__ qe, u64 nq, i32 min_overlap, const u64 *__restrict__ offsets,
                 u32 *__restrict__ out_q, u32 *__restrict__ out_s) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 q = (u64)blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += stride) {
        u64 o = offsets[q];
        igd_walk_unique<UNIQUE>(v, qc[q], (i32)qs[q], (i32)qe[q], min_overlap, [&](u32 i) {
            out_q[o] = (u32)q;
            out_s[o] = (u32)v.values[i];
            ++o;
        });
    }
}

gtars_status launch_igd_count_per_query(const IgdView &v, const u32 *qc, const u32 *qs, const u32 *qe,
                                        u64 nq, i32 min_overlap, u32 *counts, bool unique_values, hipStream_t st) {
    if (nq == 0) return GTARS_OK;
    ProfScope p("k_igd_count_per_query", st);
    if (unique_values)
        hipLaunchKernelGGL(k_igd_count_per_query<true>, dim3(stream_grid(nq, 256)), dim3(256), 0, st, v, qc, qs, qe, nq, min_overlap, counts);
    else
        hipLaunchKernelGGL(k_igd_count_per_query<false>, dim3(stream_grid(nq, 256)), dim3(256), 0, st, v, qc, qs, qe, nq, min_overlap, counts);
    GT_HIP(hipGetLastError());
    return GTARS_OK;
}

gtars_status launch_igd_fill_pairs(const IgdView &v, const u32 *qc, const u32 *qs, const u32 *qe, u64 nq,
                                   i32 min_overlap, const u64 *offsets, u32 *out_q, u32 *out_s,
                                   bool unique_values, hipStream_t st) {
    if (nq == 0) return GTARS_OK;
    ProfScope p("k_igd_fill_pairs", st);
    if (unique_values)
        hipLaunchKernelGGL(k_igd_fill_pairs<true>, dim3(stream_grid(nq, 256)), dim3(256), 0, st, v, qc, qs, qe, nq, min_overlap, offsets, out_q, out_s);
    else
        hipLaunchKernelGGL(k_igd_fill_pairs<false>, dim3(stream_grid(nq, 256)), dim3(256), 0, st, v, qc, qs, qe, nq, min_overlap, offsets, out_q, out_s);
    GT_HIP(hipGetLastError());
    return GTARS_OK;
}

// ---------------------------------------------------------------- CU hog (tests)
__global__ void __launch_bounds__(1024) k_occupy(unsigned long long ticks_100mhz, u32 lds_words, u32 *sink) {
    extern __shared__ u32 hog[];
    for (u32 i = threadIdx.x; i < lds_words; i += blockDim.x) hog[i] = i;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks_100mhz) __builtin_amdgcn_s_sleep(32);
    if (lds_words && hog[threadIdx.x % lds_words] == 0xFFFFFFFFu) *sink = 1;
}

gtars_status launch_occupy(u32 workgroups, u32 lds_bytes, u32 microseconds, hipStream_t st) {
    if (!workgroups) return GTARS_OK;
    static u32 *d_sink = nullptr;
    if (!d_sink) GT_HIP(hipMalloc((void **)&d_sink, 4));
    if (lds_bytes > 48 * 1024)
        GT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_occupy), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL(k_occupy, dim3(workgroups), dim3(1024), lds_bytes, st, (unsigned long long)microseconds * 100ull, lds_bytes / 4,
                       d_sink);
    GT_HIP(hipGetLastError());
    return GTARS_OK;
}

// ---------------------------------------------------------------- id histogram
// bins[id] += 1 for every id < n_bins: the scatter-add of gtars-scoring's count matrices (fragment_scoring.rs:88-105,
// CountMatrix::increment) -- one row of the matrix per call, the ids being the peaks hit by one file's probes.
__global__ void __launch_bounds__(256)
k_hist_u32(const u32 *__restrict__ ids, u64 n, u32 n_bins, u32 *__restrict__ bins) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const u32 k = ids[i];
        if (k < n_bins) atomicAdd(&bins[k], 1u);
    }
}

gtars_status launch_hist_u32(const u32 *ids, u64 n, u32 n_bins, u32 *bins, hipStream_t st) {
    if (n == 0 || n_bins == 0) return GTARS_OK;
    ProfScope p("k_hist_u32", st);
    hipLaunchKernelGGL(k_hist_u32, dim3(stream_grid(n, 256)), dim3(256), 0, st, ids, n, n_bins, bins);
    GT_HIP(hipGetLastError());
    return GTARS_OK;
}

// mat[(row[q] - row0) * n_cols + ids[h]] += 1 for every hit h of query q (CSR offsets / ids of a tokenization) whose row lies in
// [row0, row0 + n_rows): the scatter-add of gtars-scoring's barcode x peak counts (barcode_scoring_from_fragments,
// fragment_scoring.rs:125-155: one count per (barcode, overlapped peak)), a band of barcodes per call.  One thread per query.
__global__ void __launch_bounds__(256)
k_hist_rows(const u64 *__restrict__ offsets, const u32 *__restrict__ ids, const u32 *__restrict__ row, u64 nq, u32 row0, u32 n_rows,
            u32 n_cols, u32 *__restrict__ mat) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 q = (u64)blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += stride) {
        const u32 r = row[q] - row0;
        if (r >= n_rows) continue;
        for (u64 h = offsets[q]; h < offsets[q + 1]; ++h) {
            const u32 k = ids[h];
            if (k < n_cols) atomicAdd(&mat[(size_t)r * n_cols + k], 1u);
        }
    }
}

gtars_status launch_hist_rows(const u64 *offsets, const u32 *ids, const u32 *row, u64 nq, u32 row0, u32 n_rows, u32 n_cols, u32 *mat,
                              hipStream_t st) {
    if (nq == 0 || n_rows == 0 || n_cols == 0) return GTARS_OK;
    ProfScope p("k_hist_rows", st);
    hipLaunchKernelGGL(k_hist_rows, dim3(stream_grid(nq, 256)), dim3(256), 0, st, offsets, ids, row, nq, row0, n_rows, n_cols, mat);
    GT_HIP(hipGetLastError());
    return GTARS_OK;
}

// *dup = 1 if two neighbours of a SORTED array are equal
__global__ void __launch_bounds__(256)
k_has_adjacent_equal(const u32 *__restrict__ a, u64 n, u32 *__restrict__ dup) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    bool bad = false;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x + 1; i < n; i += stride) bad |= a[i] == a[i - 1];
    if (__any(bad) && (threadIdx.x & 63) == 0) *dup = 1u;
}

gtars_status launch_has_adjacent_equal(const u32 *a, u64 n, u32 *dup, hipStream_t st) {
    if (n < 2) return GTARS_OK;
    hipLaunchKernelGGL(k_has_adjacent_equal, dim3(stream_grid(n, 256)), dim3(256), 0, st, a, n, dup);
    GT_HIP(hipGetLastError());
    return GTARS_OK;
}

// out bit map[p] |= in bit p, p < n: a bitmap over one stored order carried to another (the flat companion of a nested
// AIList index marks hits by ITS positions; the caller wants the index's own: gtars_mark_overlapped_device).  `out` zeroed.
__global__ void __launch_bounds__(256)
k_permute_marks(const u32 *__restrict__ in, const u32 *__restrict__ map, u64 n, u32 *__restrict__ out) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 p = (u64)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += stride) {
        if (in[p >> 5] & (1u << (p & 31u))) {
            const u32 q = map[p];
            atomicOr(&out[q >> 5], 1u << (q & 31u));
        }
    }
}

gtars_status launch_permute_marks(const u32 *in, const u32 *map, u64 n, u32 *out, hipStream_t st) {
    if (!n) return GTARS_OK;
    hipLaunchKernelGGL(k_permute_marks, dim3(stream_grid(n, 256)), dim3(256), 0, st, in, map, n, out);
    GT_HIP(hipGetLastError());
    return GTARS_OK;
}

// ---------------------------------------------------------------- LOLA cells
__global__ void k_lola_contingency(const u64 *__restrict__ user_hits, const u64 *__restrict__ universe_hits,
                                   u64 n_files, i64 user_size, i64 universe_size, i64 *__restrict__ a,
                                   i64 *__restrict__ b, i64 *__restrict__ c, i64 *__restrict__ d) {
    const u64 f = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n_files) return;
    // enrichment.rs:214-220
    const i64 av = (i64)user_hits[f];
    const i64 bv = (i64)universe_hits[f] - av;
    const i64 cv = user_size - av;
    const i64 dv = universe_size - av - bv - cv;
    a[f] = av;
    b[f] = bv;
    c[f] = cv;
    d[f] = dv;
}

gtars_status launch_lola_contingency(const u64 *user_hits, const u64 *universe_hits, u64 n_files,
                                     i64 user_size, i64 universe_size, i64 *a, i64 *b, i64 *c, i64 *d,
                                     hipStream_t st) {
    if (n_files == 0) return GTARS_OK;
    ProfScope p("k_lola_contingency", st);
    hipLaunchKernelGGL(k_lola_contingency, dim3((unsigned)((n_files + 255) / 256)), dim3(256), 0, st,
                       user_hits, universe_hits, n_files, user_size, universe_size, a, b, c, d);
    GT_HIP(hipGetLastError());
    return GTARS_OK;
}

}  // namespace gtars
