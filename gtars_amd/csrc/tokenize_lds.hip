// tokenize_lds.hip -- the fast path of Tokenizer::tokenize / encode
// (gtars-tokenizers/src/tokenizer.rs:140-171 -> Bits::find, bits.rs:141-156,
// 433-446) for a Bits-kind index.
//
// Data layout (AccelView, common.h): the sorted index is cut into 8-interval
// blocks, one 128-byte record (= one L2 line) per block; the first start of
// every 2^top_shift-th block forms a small "top" array.
//
// Kernel: persistent workgroups copy `top` (+ the per-chromosome tables) into
// LDS once, then take tiles of TPB*4 consecutive queries through a ticket
// counter.  Per query:
//   1. LDS binary search of `top` (lock-step over the thread's 4 queries so the
//      4 searches overlap) for the last sampled block whose first start is
//      < q_start - max_len  -- Bits::find's lower_bound, bits.rs:144-147;
//      (+ a short search of blk_first in L2 when top_shift > 0);
//   2. ONE 128-byte block fetch (4 x dwordx4 per lane), overlap test of its 8
//      intervals in registers -> 8-bit hit mask; the scan continues into the
//      following blocks only while the block's last start is still < q_end
//      (iv.start >= stop ends the reference scan, bits.rs:441-443);
//   3. wave shuffles + one LDS word per wave scan the per-thread hit counts,
//      wave 0 resolves the tile's global base by chained look-back (scan.cuh);
//   4. CSR offsets (u64) and token ids (u32) are written once, in place.
// Starting the scan at a block boundary instead of the exact lower_bound only
// adds intervals with start < q_start - max_len, which cannot satisfy
// end > q_start, so the hit set and its order are exactly Bits::find's.
//
// Bound: HBM stream of queries in / offsets+ids out (23.5 B per query at
// config 2); the index itself stays L2/LDS resident.  No MFMA: integer search.
#include "common.h"
#include "scan.cuh"

namespace gtars {

constexpr int TOK_QPT = 4;

__device__ __forceinline__ i64 overlap_bp_tok(u32 as, u32 ae, u32 bs, u32 be) {
    u32 mn = ae < be ? ae : be;
    u32 mx = as > bs ? as : bs;
    return (i64)mn - (i64)mx;
}

// hit mask of one block for one query; *more = the scan must continue
template <bool FILTER>
__device__ __forceinline__ u32 block_mask(const uint4 *__restrict__ blk, u32 qs, u32 qe, i32 min_bp, bool &more) {
    const uint4 s0 = blk[0], s1 = blk[1], e0 = blk[2], e1 = blk[3];
    const u32 s[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
    const u32 e[8] = {e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w};
    u32 m = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        bool hit = (s[k] < qe) & (e[k] > qs);
        if (FILTER) hit = hit && overlap_bp_tok(qs, qe, s[k], e[k]) >= (i64)min_bp;
        m |= (hit ? 1u : 0u) << k;
    }
    more = s[7] < qe;  // starts ascend inside a block: no stop seen yet
    return m;
}

template <int TPB, bool FILTER>
__global__ void __launch_bounds__(TPB)
k_tok_lds(AccelView a, const u32 *__restrict__ qc, const u32 *__restrict__ qs, const u32 *__restrict__ qe,
          u64 nq, i32 min_bp, u64 *__restrict__ offsets, u32 *__restrict__ ovals, u32 *__restrict__ ostarts,
          u32 *__restrict__ oends, u64 cap, ScanWs *ws, u32 search_steps) {
    extern __shared__ __attribute__((aligned(16))) u32 smem[];
    __shared__ u32 s_tile;
    __shared__ u64 s_prefix;
    __shared__ u32 s_scan[TPB / 64];
    constexpr int TILE = TPB * TOK_QPT;

    const u32 n_top = a.n_top;
    const u32 n_top_pad = (n_top + 3u) & ~3u;
    u32 *s_top = smem;
    u32 *s_cboff = smem + n_top_pad;       // [n_chrom + 1]
    u32 *s_cmax = s_cboff + a.n_chrom + 1;  // [n_chrom]
    for (u32 i = threadIdx.x; i < n_top; i += TPB) s_top[i] = a.top[i];
    for (u32 i = threadIdx.x; i <= a.n_chrom; i += TPB) s_cboff[i] = a.chrom_blk_off[i];
    for (u32 i = threadIdx.x; i < a.n_chrom; i += TPB) s_cmax[i] = a.chrom_maxlen[i];
    __syncthreads();

    const u32 num_tiles = (u32)((nq + TILE - 1) / TILE);
    const int lane = threadIdx.x & 63;
    const bool vec_ok = ((((uintptr_t)qc) | ((uintptr_t)qs) | ((uintptr_t)qe)) & 15u) == 0;
    const u32 shift = a.top_shift;
    const u32 *blkw = reinterpret_cast<const u32 *>(a.blocks);

    for (;;) {
        if (threadIdx.x == 0) s_tile = atomicAdd(&ws->ticket, 1u);
        __syncthreads();
        const u32 tile = s_tile;
        if (tile >= num_tiles) break;

        const u64 q0 = (u64)tile * TILE + (u64)threadIdx.x * TOK_QPT;
        u32 c[TOK_QPT], s[TOK_QPT], e[TOK_QPT];
        if (vec_ok && q0 + TOK_QPT <= nq) {
            const uint4 c4 = *reinterpret_cast<const uint4 *>(qc + q0);
            const uint4 s4 = *reinterpret_cast<const uint4 *>(qs + q0);
            const uint4 e4 = *reinterpret_cast<const uint4 *>(qe + q0);
            c[0] = c4.x; c[1] = c4.y; c[2] = c4.z; c[3] = c4.w;
            s[0] = s4.x; s[1] = s4.y; s[2] = s4.z; s[3] = s4.w;
            e[0] = e4.x; e[1] = e4.y; e[2] = e4.z; e[3] = e4.w;
        } else {
#pragma unroll
            for (int j = 0; j < TOK_QPT; ++j) {
                const bool ok = q0 + j < nq;
                c[j] = ok ? qc[q0 + j] : GTARS_UNKNOWN_CHROM;
                s[j] = ok ? qs[q0 + j] : 0;
                e[j] = ok ? qe[q0 + j] : 0;
            }
        }

        // ---- 1. search: last (sampled) block whose first start < key ----------
        u32 key[TOK_QPT], lo[TOK_QPT], len[TOK_QPT], tbase[TOK_QPT], be[TOK_QPT];
#pragma unroll
        for (int j = 0; j < TOK_QPT; ++j) {
            const bool valid = c[j] < a.n_chrom;
            const u32 cc = valid ? c[j] : 0u;
            const u32 bb = s_cboff[cc];
            be[j] = valid ? s_cboff[cc + 1] : bb;  // invalid -> empty range
            const u32 ml = s_cmax[cc];
            key[j] = s[j] >= ml ? s[j] - ml : 0u;
            tbase[j] = bb >> shift;
            lo[j] = tbase[j];
            len[j] = (be[j] - bb) >> shift;
        }
        for (u32 it = 0; it < search_steps; ++it) {
#pragma unroll
            for (int j = 0; j < TOK_QPT; ++j) {
                const u32 half = len[j] >> 1;
                const u32 mid = lo[j] + half;
                const u32 v = s_top[mid < n_top ? mid : n_top - 1];
                const bool pred = (len[j] > 0) & (v < key[j]);
                lo[j] = pred ? mid + 1 : lo[j];
                len[j] = pred ? len[j] - half - 1 : half;
            }
        }
        u32 b0[TOK_QPT];
#pragma unroll
        for (int j = 0; j < TOK_QPT; ++j) {
            const u32 nlt = lo[j] - tbase[j];                 // sampled entries < key
            const u32 t0 = tbase[j] + (nlt ? nlt - 1 : 0u);
            u32 b = t0 << shift;
            if (shift) {
                // blocks [b, b + 2^shift): count blk_first < key (first entry is top[t0])
                u32 l2 = b, n2 = (be[j] > b) ? min(1u << shift, be[j] - b) : 0u;
                while (n2 > 0) {
                    const u32 half = n2 >> 1, mid = l2 + half;
                    const bool pred = a.blk_first[mid] < key[j];
                    l2 = pred ? mid + 1 : l2;
                    n2 = pred ? n2 - half - 1 : half;
                }
                const u32 nlt2 = l2 - b;
                b += nlt2 ? nlt2 - 1 : 0u;
            }
            b0[j] = b;
        }

        // ---- 2. first block of every query (independent 128-B fetches) --------
        u32 mask[TOK_QPT], cnt[TOK_QPT];
        bool more[TOK_QPT];
#pragma unroll
        for (int j = 0; j < TOK_QPT; ++j) {
            const bool act = b0[j] < be[j];
            const u32 b = act ? b0[j] : 0u;
            bool mr;
            const u32 m = block_mask<FILTER>(a.blocks + (size_t)b * 8, s[j], e[j], min_bp, mr);
            mask[j] = act ? m : 0u;
            more[j] = act && mr;
        }
        // continuation blocks (rare: the query reaches past the block's last start)
        u32 tsum = 0;
#pragma unroll
        for (int j = 0; j < TOK_QPT; ++j) {
            u32 n = __popc(mask[j]);
            if (more[j]) {
                u32 b = b0[j] + 1;
                bool mr = true;
                while (mr && b < be[j]) {
                    const u32 m = block_mask<FILTER>(a.blocks + (size_t)b * 8, s[j], e[j], min_bp, mr);
                    if (b == b0[j] + 1) mask[j] |= m << 8;
                    n += __popc(m);
                    ++b;
                }
                // more[j] stays true only if blocks beyond the second were visited
                more[j] = b > b0[j] + 2;
            }
            cnt[j] = n;
            tsum += n;
        }

        // ---- 3. scan ------------------------------------------------------------
        u32 block_total;
        const u32 excl = block_exclusive_scan<TPB>(tsum, s_scan, block_total);
        if (threadIdx.x < 64) {
            const u64 p = lookback(ws->state, tile, (u64)block_total, lane, &ws->err);
            if (lane == 0) s_prefix = p;
        }
        __syncthreads();
        const u64 prefix = s_prefix;
        u64 run = prefix + excl;
        if (tile == num_tiles - 1 && threadIdx.x == TPB - 1) {
            const u64 tot = prefix + (u64)block_total;
            offsets[nq] = tot;
            ws->total = tot;
        }

        // ---- 4. write offsets + payloads ----------------------------------------
        if (q0 + TOK_QPT <= nq && ((((uintptr_t)offsets) & 15u) == 0)) {
            const u64 o0 = run, o1 = o0 + cnt[0], o2 = o1 + cnt[1], o3 = o2 + cnt[2];
            ulonglong2 *dst = reinterpret_cast<ulonglong2 *>(offsets + q0);
            dst[0] = make_ulonglong2(o0, o1);
            dst[1] = make_ulonglong2(o2, o3);
        } else {
            u64 r = run;
#pragma unroll
            for (int j = 0; j < TOK_QPT; ++j) {
                if (q0 + j < nq) offsets[q0 + j] = r;
                r += cnt[j];
            }
        }
#pragma unroll
        for (int j = 0; j < TOK_QPT; ++j) {
            if (cnt[j]) {
                u64 o = run;
                u32 m = mask[j];
                while (m) {
                    const int k = __ffs((int)m) - 1;
                    m &= m - 1;
                    const u32 w = (b0[j] + (u32)(k >> 3)) * 32u + (u32)(k & 7);
                    if (o < cap) {
                        if (ovals) ovals[o] = blkw[w + 16];
                        if (ostarts) ostarts[o] = blkw[w];
                        if (oends) oends[o] = blkw[w + 8];
                    }
                    ++o;
                }
                if (more[j]) {
                    // blocks beyond the second: recompute their masks
                    u32 b = b0[j] + 2;
                    bool mr = true;
                    while (mr && b < be[j]) {
                        u32 mm = block_mask<FILTER>(a.blocks + (size_t)b * 8, s[j], e[j], min_bp, mr);
                        while (mm) {
                            const int k = __ffs((int)mm) - 1;
                            mm &= mm - 1;
                            const u32 w = b * 32u + (u32)k;
                            if (o < cap) {
                                if (ovals) ovals[o] = blkw[w + 16];
                                if (ostarts) ostarts[o] = blkw[w];
                                if (oends) oends[o] = blkw[w + 8];
                            }
                            ++o;
                        }
                        ++b;
                    }
                }
            }
            run += cnt[j];
        }
        __syncthreads();  // s_tile / s_prefix reuse
    }
}

// ---------------------------------------------------------------- launcher

static int env_int(const char *name, int dflt) {
    const char *v = getenv(name);
    return v && *v ? atoi(v) : dflt;
}

static size_t tok_lds_bytes(const AccelView &a) {
    const size_t n_top_pad = ((size_t)a.n_top + 3) & ~(size_t)3;
    return (n_top_pad + 2 * (size_t)a.n_chrom + 1) * sizeof(u32);
}

bool tokenize_lds_supported(const AccelView &a) {
    return a.n_blocks > 0 && a.n_top > 0 && tok_lds_bytes(a) <= 120 * 1024;
}

static int choose_tpb(u64 nq) {
    const int forced = env_int("GTARS_TOK_TPB", 0);
    if (forced == 256 || forced == 512 || forced == 1024) return forced;
    return nq >= (1ull << 22) ? 512 : 256;
}

size_t tokenize_lds_ws_bytes(u64 nq) {
    // sized for the smallest tile (256 threads x 4 queries)
    return scan_ws_bytes_for_tiles((nq + 256 * TOK_QPT - 1) / (256 * TOK_QPT));
}

template <int TPB, bool FILTER>
static gtars_status launch_tok_t(const AccelView &a, const u32 *qc, const u32 *qs, const u32 *qe, u64 nq,
                                 i32 min_bp, const EnumOut &out, ScanWs *ws, hipStream_t st) {
    const size_t lds = tok_lds_bytes(a);
    auto kern = k_tok_lds<TPB, FILTER>;
    static thread_local size_t attr_set = 0;
    if (lds > 48 * 1024 && lds > attr_set) {
        GT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = lds;
    }
    int per_cu = 0;
    GT_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, TPB, lds));
    if (per_cu < 1) return fail(GTARS_ERR_INTERNAL, "k_tok_lds does not fit on a CU");
    const int cap_per_cu = env_int("GTARS_TOK_WG_PER_CU", 0);
    if (cap_per_cu > 0 && per_cu > cap_per_cu) per_cu = cap_per_cu;
    int dev = 0, cus = 256;
    GT_HIP(hipGetDevice(&dev));
    GT_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const u64 tiles = (nq + (u64)TPB * TOK_QPT - 1) / ((u64)TPB * TOK_QPT);
    u64 grid = (u64)cus * per_cu;
    if (grid > tiles) grid = tiles;
    u32 steps = 0;
    while ((1u << steps) <= a.max_chrom_top) ++steps;  // iterations until len == 0
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(TPB), lds, st, a, qc, qs, qe, nq, min_bp, out.offsets,
                       out.vals, out.starts, out.ends, out.capacity, ws, steps);
    GT_HIP(hipGetLastError());
    return GTARS_OK;
}

gtars_status launch_tokenize_lds(const AccelView &a, const u32 *qc, const u32 *qs, const u32 *qe, u64 nq,
                                 int has_min, i32 min_overlap, const EnumOut &out, void *scan_ws,
                                 size_t scan_ws_bytes, hipStream_t st) {
    if (nq == 0) {
        GT_HIP(hipMemsetAsync(out.offsets, 0, sizeof(u64), st));
        GT_HIP(hipMemsetAsync(scan_ws, 0, sizeof(ScanWs), st));
        return GTARS_OK;
    }
    const int tpb = choose_tpb(nq);
    const u64 tiles = (nq + (u64)tpb * TOK_QPT - 1) / ((u64)tpb * TOK_QPT);
    if (tiles > 0xFFFFFFF0ull) return fail(GTARS_ERR_INVALID_ARG, "query batch too large for one launch");
    const size_t need = scan_ws_bytes_for_tiles(tiles);
    if (scan_ws_bytes < need) return fail(GTARS_ERR_INTERNAL, "fused scan workspace too small");
    GT_HIP(hipMemsetAsync(scan_ws, 0, need, st));
    const bool filter = has_min && min_overlap > 1;
    const i32 min_bp = has_min ? min_overlap : 0;
    ScanWs *ws = (ScanWs *)scan_ws;
    ProfScope p("k_tok_lds", st);
    if (tpb == 256)
        return filter ? launch_tok_t<256, true>(a, qc, qs, qe, nq, min_bp, out, ws, st)
                      : launch_tok_t<256, false>(a, qc, qs, qe, nq, min_bp, out, ws, st);
    if (tpb == 512)
        return filter ? launch_tok_t<512, true>(a, qc, qs, qe, nq, min_bp, out, ws, st)
                      : launch_tok_t<512, false>(a, qc, qs, qe, nq, min_bp, out, ws, st);
    return filter ? launch_tok_t<1024, true>(a, qc, qs, qe, nq, min_bp, out, ws, st)
                  : launch_tok_t<1024, false>(a, qc, qs, qe, nq, min_bp, out, ws, st);
}

}  // namespace gtars
