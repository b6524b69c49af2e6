// tokenize_lds.hip -- the fast path of Tokenizer::tokenize / encode
// (gtars-tokenizers/src/tokenizer.rs:140-171 -> Bits::find, bits.rs:141-156,
// 433-446) for a Bits-kind index.
//
// Data layout (AccelView, common.h): the sorted index is cut into 3-interval
// blocks, one 64-byte record per block that also carries a copy of the next
// block's first interval (the look-ahead); two levels of search keys over the
// blocks' last starts are small enough to live in LDS.
//
// Kernel: persistent workgroups copy the keys (+ the per-chromosome table) into
// LDS once, then take tiles of TPB*4 consecutive queries.  Per query:
//   1. LDS search for the first block that can hold a hit: the first whose key --
//      the largest end among all intervals up to and including the block -- is
//      > q_start.  (Bits::find starts at lower_bound(q_start - max_len),
//      bits.rs:144-147; every interval between that point and ours has
//      end <= q_start, so the hit set and its order are the same.)  A bucket
//      table over the one ascending key space, then a lock-step search with a
//      scalar step sequence over quantised u16 keys
//      (+ a short search of blk_first[] in L2 when top_shift > 0);
//   2. ONE burst of three 16-byte loads from the block's record: starts, ends and
//      token ids of its 3 intervals and of the look-ahead interval.  The overlap
//      test runs in registers -> 4-bit hit mask and the first two ids; only a
//      query that runs past the look-ahead start (iv.start >= stop ends the
//      reference scan, bits.rs:441-443) walks into the following blocks;
//   3. wave shuffles + one LDS word per wave scan the per-thread hit counts,
//      wave 0 resolves the tile's global base by chained look-back (scan.cuh);
//   4. CSR offsets (u64) and token ids (u32) are written once, in place.
// Starting the scan at a block boundary (or, after the quantised search, one
// block early) only adds intervals with end <= q_start, which cannot overlap,
// so the hit set and its order are exactly Bits::find's.
//
// Bound: HBM stream of queries in / offsets+ids out (23.5 B per query at
// config 2); the index itself stays L2/LDS resident.  What the count phase
// actually pays is the CU's vector-L1 fill rate for one random line per query
// (see AccelView).  No MFMA: integer search.
#include "common.h"
#include "scan.cuh"

namespace gtars {


// Timing experiments only (tools/ablate.sh builds a separate library with
// -DGTARS_ABLATE=<bits>; results are then WRONG by construction):
//  2: no record fetch   4: no id writes   8: no offset writes   16: no look-back
//  64: look-back never waits (scan.cuh)   128: per-phase cycle stamps printed by the launcher
#ifndef GTARS_ABLATE
#define GTARS_ABLATE 0
#endif
// Non-temporal streaming so that the query / result streams do not push the index (2.2 MB per XCD) out
// of the 4 MB L2 while a launch runs.  Bits: 1 query loads, 2 offset stores (full 32 B per lane), 4 id
// stores.  The id stores stay temporal: they are scattered 4-byte stores, and as non-temporal ones L2
// no longer merges them into full lines (WRITE_SIZE 16.2 MB instead of 10.35 MB per 1M queries).
#ifndef GTARS_TOK_NT
#define GTARS_TOK_NT 3
#endif
// 1: the next tile's queries are loaded right after the count phase; 0: at the start of its own iteration
// (measured: 1 is 15 % slower -- vmcnt retires in order, so every later wait also waits for that HBM stream)
#ifndef GTARS_TOK_PREFETCH
#define GTARS_TOK_PREFETCH 0
#endif

__device__ __forceinline__ i64 overlap_bp_tok(u32 as, u32 ae, u32 bs, u32 be) {
    u32 mn = ae < be ? ae : be;
    u32 mx = as > bs ? as : bs;
    return (i64)mn - (i64)mx;
}

typedef u32 u32x4 __attribute__((ext_vector_type(4)));
typedef u32 u32x2 __attribute__((ext_vector_type(2)));
typedef u64 u64x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u32x4 ld_stream4(const u32 *p) {
    return (GTARS_TOK_NT & 1) ? __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p)) : *reinterpret_cast<const u32x4 *>(p);
}
__device__ __forceinline__ u32x2 ld_stream2(const u32 *p) {
    return (GTARS_TOK_NT & 1) ? __builtin_nontemporal_load(reinterpret_cast<const u32x2 *>(p)) : *reinterpret_cast<const u32x2 *>(p);
}
__device__ __forceinline__ void st_stream(u32 *p, u32 v) {
    if (GTARS_TOK_NT & 4) __builtin_nontemporal_store(v, p); else *p = v;
}
__device__ __forceinline__ void st_stream2(u64 *p, u64 a, u64 b) {
    u64x2 v = {a, b};
    if (GTARS_TOK_NT & 2) __builtin_nontemporal_store(v, reinterpret_cast<u64x2 *>(p)); else *reinterpret_cast<u64x2 *>(p) = v;
}

template <bool FILTER>
__device__ __forceinline__ bool hit_test(u32 s, u32 e, u32 qs, u32 qe, i32 min_bp) {
    bool hit = (s < qe) & (e > qs);  // Interval::overlap, interval.rs:47-50
    if (FILTER) hit = hit && overlap_bp_tok(qs, qe, s, e) >= (i64)min_bp;
    return hit;
}

// Block record (16 words = 64 B), see AccelView:
//   quad 0: s0 s1 s2 ns0   quad 1: e0 e1 e2 ne0   quad 2: v0 v1 v2 nv0   quad 3: ns1 ne1 nv1 ns2
// 5-bit hit mask: the three own intervals (bits 0..2) and the two look-ahead intervals (bits 3,4);
// more = the reference scan would run past the look-ahead too.
template <bool FILTER>
__device__ __forceinline__ u32 block_mask5(const uint4 &S, const uint4 &E, const uint4 &L, u32 qs, u32 qe, i32 min_bp,
                                           bool &more) {
    u32 m = 0;
    m |= (hit_test<FILTER>(S.x, E.x, qs, qe, min_bp) ? 1u : 0u);
    m |= (hit_test<FILTER>(S.y, E.y, qs, qe, min_bp) ? 1u : 0u) << 1;
    m |= (hit_test<FILTER>(S.z, E.z, qs, qe, min_bp) ? 1u : 0u) << 2;
    m |= (hit_test<FILTER>(S.w, E.w, qs, qe, min_bp) ? 1u : 0u) << 3;
    m |= (hit_test<FILTER>(L.x, L.y, qs, qe, min_bp) ? 1u : 0u) << 4;
    more = L.w < qe;  // starts ascend: no stop seen yet (bits.rs:441-443)
    return m;
}

// word index (inside the block record) of the val of hit bit k (0..4)
__device__ __forceinline__ u32 val_word(int k) { return k < 4 ? 8u + (u32)k : 14u; }

// Per-query state kept between the count phase and the write phase, in ONE
// register: first block (22 bits) + 5-bit hit mask (3 own + 2 look-ahead).
constexpr u32 B0_BITS = 22;
constexpr u32 B0_MASK = (1u << B0_BITS) - 1u;

// Tail of a query that runs past block b0's look-ahead (rare): blocks b0+1..,
// skipping the first two intervals of block b0+1 (they were the look-ahead).
// f(block, k) is called for every hit in scan order; returns the hit count.
template <bool FILTER, class F>
__device__ __forceinline__ u32 walk_tail(const AccelView &a, u32 b0, u32 be, u32 qs, u32 qe, i32 min_bp, F &&f) {
    u32 n = 0;
    bool mr = true;
    for (u32 b = b0 + 1; mr && b < be; ++b) {
        const uint4 S = a.blocks[(size_t)b * 4], E = a.blocks[(size_t)b * 4 + 1];
        u32 m = 0;
        m |= (hit_test<FILTER>(S.x, E.x, qs, qe, min_bp) ? 1u : 0u);
        m |= (hit_test<FILTER>(S.y, E.y, qs, qe, min_bp) ? 1u : 0u) << 1;
        m |= (hit_test<FILTER>(S.z, E.z, qs, qe, min_bp) ? 1u : 0u) << 2;
        if (b == b0 + 1) m &= ~3u;
        mr = S.z < qe;
        n += __popc(m);
        while (m) {
            const int k = __ffs((int)m) - 1;
            m &= m - 1;
            f(b, k);
        }
    }
    return n;
}

// First block that can hold a hit, for SUB queries of one thread at once (their LDS round trips overlap).
// All unit keys live in one ascending key space (AccelView).  A direct-mapped bucket table narrows the
// range to a handful of units; the in-bucket search then runs the same scalar step sequence in every
// lane, clamped to the lane's own range: per step one add, one min, one LDS read, one compare, one select.
// b0[j] >= be[j] means "no candidate" (also for an unknown chromosome: be = 0).
template <int SUB>
__device__ __forceinline__ void search_blocks(const AccelView &a, const u32 *s_lut, const u32 *s_q, const uint4 *s_ctab,
                                              const u32 *c, const u32 *s, u32 *b0, u32 *be) {
    // LDS byte addresses (32-bit, address space 3) so that a step needs no address math
    typedef const __attribute__((address_space(3))) unsigned short *lds_cu16;
    const u32 lb = (u32)(uintptr_t)(lds_cu16)reinterpret_cast<const unsigned short *>(s_lut);
    const u32 qb = (u32)(uintptr_t)(lds_cu16)reinterpret_cast<const unsigned short *>(s_q);
    const u32 lsh = a.lut_shift, qsh = a.q_shift, shift = a.top_shift;
    const u32 wmask = (1u << lsh) - 1u;
    u32 pos[SUB], tq[SUB], last[SUB];
#pragma unroll
    for (int j = 0; j < SUB; ++j) {
        const bool valid = c[j] < a.n_chrom;
        const uint4 ct = s_ctab[valid ? c[j] : 0u];
        // target = q_start + 1 in the key space of prefix-max ends (first block with key > q_start);
        // at or beyond the chromosome's largest end: the sentinel key
        // (the key space is 64 bits wide: genomes beyond 4.29 Gbp; ct.z = high word of gbase)
        const u64 gkey = (((u64)ct.z << 32) | ct.x) + (s[j] < ct.y ? s[j] + 1u : ct.y);
        be[j] = valid ? ct.w : 0u;  // invalid -> empty range
        const u32 la = lb + ((u32)(gkey >> lsh) << 1);
        const u32 lo = *(lds_cu16)(uintptr_t)la, hi = *(lds_cu16)(uintptr_t)(la + 2u);
        tq[j] = hi > lo ? ((u32)gkey & wmask) >> qsh : 0u;  // empty bucket: no key is < 0
        pos[j] = qb + (lo << 1) - 2u;                  // &q[lo - 1]
        last[j] = qb + (hi << 1) - 2u;                 // &q[hi - 1]
    }
    for (u32 step = a.search_top << 1; step >= 2; step >>= 1) {  // byte steps
#pragma unroll
        for (int j = 0; j < SUB; ++j) {
            const u32 cand = min(pos[j] + step, last[j]);
            const u32 v = *(lds_cu16)(uintptr_t)cand;
            pos[j] = v < tq[j] ? cand : pos[j];
        }
    }
#pragma unroll
    for (int j = 0; j < SUB; ++j) {
        u32 b = ((pos[j] + 2u - qb) >> 1) << shift;  // first block of the first unit with key >= target
        if (shift) {
            // inside the unit: first block whose key (prefix-max end) is > q_start.  The keys ascend and a
            // chromosome's block range is padded to whole units (padding key 0xFFFFFFFF), so for small units
            // that is b + #{keys <= q_start}, counted on 16-byte loads that share ONE L2 line: a single
            // dependent round trip instead of `shift` of them (each a fresh L1 line fill).
            if (b < be[j]) {
                if (shift >= 2 && shift <= 4) {
                    u32 cnt = 0;
                    const uint4 *kp = reinterpret_cast<const uint4 *>(a.blk_first + b);
#pragma unroll 4
                    for (u32 k = 0; k < (1u << (shift - 2)); ++k) {
                        const uint4 kk = kp[k];
                        cnt += (kk.x <= s[j] ? 1u : 0u) + (kk.y <= s[j] ? 1u : 0u) + (kk.z <= s[j] ? 1u : 0u) +
                               (kk.w <= s[j] ? 1u : 0u);
                    }
                    b += cnt;
                } else {
                    u32 l2 = b, n2 = min(1u << shift, be[j] - b);
                    while (n2 > 0) {
                        const u32 half = n2 >> 1, mid = l2 + half;
                        const bool pred = a.blk_first[mid] <= s[j];
                        l2 = pred ? mid + 1 : l2;
                        n2 = pred ? n2 - half - 1 : half;
                    }
                    b = l2;
                }
            }
        }
        b0[j] = b;
    }
}

// copy the search structure into LDS: bucket table (u16) | unit keys (u16) | chromosome table
template <int TPB>
__device__ __forceinline__ void fill_search_lds(const AccelView &a, u32 *smem) {
    // 16-byte loads, 4 in flight.  Every workgroup copies the same arrays: start each one at a
    // different place so that they do not all queue on the same L2 channel at the same time.
    const u32 n4a = a.lut_words >> 2, n4 = n4a + (a.q_words >> 2);
    const uint4 *src_a = reinterpret_cast<const uint4 *>(a.lut);
    const uint4 *src_b = reinterpret_cast<const uint4 *>(a.qkeys);
    uint4 *dst = reinterpret_cast<uint4 *>(smem);
    const u32 rot = (u32)(((u64)blockIdx.x * 2654435761ull) % n4);
#pragma unroll 4
    for (u32 i = threadIdx.x; i < n4; i += TPB) {
        u32 k = i + rot;
        k = k >= n4 ? k - n4 : k;
        dst[k] = k < n4a ? src_a[k] : src_b[k - n4a];
    }
    uint4 *s_ctab = reinterpret_cast<uint4 *>(smem + a.lut_words + a.q_words);
    for (u32 i = threadIdx.x; i < a.n_chrom; i += TPB) s_ctab[i] = a.chrom_tab[i];
}

template <int TOK_QPT>
struct TileState {
    u32 st[TOK_QPT];  // b0 | mask5 << 22
    u32 v0[TOK_QPT];  // ids of the first two hits: loaded during the count phase,
    u32 v1[TOK_QPT];  //   stored one tile later (their latency is off the critical path)
    u32 excl;         // exclusive hit offset of the thread's queries inside the tile
    u32 more_bits;    // bit j: query j runs past its first block's look-ahead
    u32 total;        // hits of the whole tile
    u32 tile;
};

// TOK_QPT = queries per thread: 4 (one burst of 16 loads per round); 2 is kept for experiments.
template <int TPB, int TOK_QPT, bool FILTER>
__global__ void __launch_bounds__(TPB, (TOK_QPT == 2 ? 6 : 4))
k_tok_lds(AccelView a, const u32 *__restrict__ qc, const u32 *__restrict__ qs, const u32 *__restrict__ qe,
          u64 nq, i32 min_bp, u64 *__restrict__ offsets, u32 *__restrict__ ovals, u64 cap, ScanWs *ws,
          u32 epoch, u32 ticket_base) {
#if GTARS_ABLATE & 128
    const long long t_entry = clock64();
#endif
    extern __shared__ __attribute__((aligned(16))) u32 smem[];
    __shared__ u32 s_tile;
    __shared__ u64 s_prefix;
    constexpr int NW = TPB / 64;
    __shared__ u32 s_scan[NW];
    constexpr int TILE = TPB * TOK_QPT;

    const u32 num_tiles = (u32)((nq + TILE - 1) / TILE);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool vec_ok = ((((uintptr_t)qc) | ((uintptr_t)qs) | ((uintptr_t)qe)) & 15u) == 0;
    const bool off_vec_ok = (((uintptr_t)offsets) & 15u) == 0;
    const u32 *blkw = reinterpret_cast<const u32 *>(a.blocks);

    u32 c[TOK_QPT], s[TOK_QPT], e[TOK_QPT];
    auto load_queries = [&](u32 t) {
        const u64 q0 = (u64)t * TILE + (u64)threadIdx.x * TOK_QPT;
        if (t >= num_tiles) return;
        if (vec_ok && q0 + TOK_QPT <= nq) {
            if constexpr (TOK_QPT == 4) {
                const u32x4 c4 = ld_stream4(qc + q0), s4 = ld_stream4(qs + q0), e4 = ld_stream4(qe + q0);
                c[0] = c4.x; c[1] = c4.y; c[2] = c4.z; c[3] = c4.w;
                s[0] = s4.x; s[1] = s4.y; s[2] = s4.z; s[3] = s4.w;
                e[0] = e4.x; e[1] = e4.y; e[2] = e4.z; e[3] = e4.w;
            } else if constexpr (TOK_QPT == 8) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const u32x4 c4 = ld_stream4(qc + q0 + 4 * h), s4 = ld_stream4(qs + q0 + 4 * h), e4 = ld_stream4(qe + q0 + 4 * h);
                    c[4 * h] = c4.x; c[4 * h + 1] = c4.y; c[4 * h + 2] = c4.z; c[4 * h + 3] = c4.w;
                    s[4 * h] = s4.x; s[4 * h + 1] = s4.y; s[4 * h + 2] = s4.z; s[4 * h + 3] = s4.w;
                    e[4 * h] = e4.x; e[4 * h + 1] = e4.y; e[4 * h + 2] = e4.z; e[4 * h + 3] = e4.w;
                }
            } else {
                const u32x2 c2 = ld_stream2(qc + q0), s2 = ld_stream2(qs + q0), e2 = ld_stream2(qe + q0);
                c[0] = c2.x; c[1] = c2.y;
                s[0] = s2.x; s[1] = s2.y;
                e[0] = e2.x; e[1] = e2.y;
            }
        } else {
#pragma unroll
            for (int j = 0; j < TOK_QPT; ++j) {
                const bool ok = q0 + j < nq;
                c[j] = ok ? qc[q0 + j] : GTARS_UNKNOWN_CHROM;
                s[j] = ok ? qs[q0 + j] : 0;
                e[j] = ok ? qe[q0 + j] : 0;
            }
        }
    };
    // the first tile's queries come from HBM: issue their loads before the LDS fill so that both overlap
    load_queries(blockIdx.x);

    // LDS: bucket table (u16) | unit keys (u16) | chromosome table; both key arrays are padded to 16 bytes
    u32 *s_lut = smem;
    u32 *s_q = smem + a.lut_words;
    uint4 *s_ctab = reinterpret_cast<uint4 *>(smem + a.lut_words + a.q_words);  // [n_chrom] {gbase, span, 0, blk_end}
    fill_search_lds<TPB>(a, smem);
    __syncthreads();

    // Software pipeline across tiles: a tile's hit count (its "aggregate") is
    // published as soon as it is known, but its global base is resolved -- and
    // its outputs written -- only after the NEXT tile has been counted.  By then
    // every predecessor has had a whole tile time to publish, so the look-back
    // rarely waits and its latency is off the critical path.
    TileState<TOK_QPT> cur, prev;
    bool have_prev = false;
#if GTARS_ABLATE & 128
    long long t_ticket = 0, t_count = 0, t_scan = 0, t_resolve = 0, t_write = 0, t_c1 = 0, t_c2 = 0, t_c3 = 0, t_mark;
    const long long t_fill = clock64() - t_entry;
#define GT_STAMP(acc) do { const long long _n = clock64(); acc += _n - t_mark; t_mark = _n; } while (0)
    t_mark = clock64();
#else
#define GT_STAMP(acc) do { } while (0)
#endif

    // Tile assignment: the grid never exceeds what is resident at once (launcher), so every
    // workgroup's FIRST tile is simply its block index -- no atomic while the whole grid starts up.
    // Further tiles are drawn from a ticket counter, so a tile only ever waits on tiles that are
    // already running, whatever the dispatch order.
    //
    // The ticket of the NEXT tile is drawn (one lane) before the current tile is counted and handed
    // round through the scan's barrier, so no barrier is spent on it.  With GTARS_TOK_PREFETCH the next
    // tile's queries are also loaded right after the count phase (wave 0 only after its look-back
    // loads); that variant is kept for experiments only -- see the macro.
    const bool draw = num_tiles > gridDim.x;  // otherwise one tile per workgroup: nothing to draw
    u32 tile = blockIdx.x;
    bool loaded = true;  // the first tile's queries are already in flight
    for (;;) {
        const bool has_cur = tile < num_tiles;
        u32 next_tile = num_tiles;
        GT_STAMP(t_ticket);

        if (has_cur) {
            u32 ticket = 0;
            if (draw && threadIdx.x == 0) ticket = atomicAdd(&ws->ticket, 1u);  // consumed after the count phase
            if (!GTARS_TOK_PREFETCH && !loaded) load_queries(tile);
            loaded = false;
            // =============== count phase: 4 consecutive queries per thread ===============

            // A thread's TOK_QPT consecutive queries go through search + record fetch in rounds of SUB = 4
            // (the register budget of one burst); everything per tile -- ticket, scan, look-back,
            // barriers -- is paid once for all of them.
            constexpr int SUB = TOK_QPT < 4 ? TOK_QPT : 4;
            u32 tsum = 0;
            cur.more_bits = 0;
#pragma unroll
            for (int r0 = 0; r0 < TOK_QPT; r0 += SUB) {
                // ---- 1. search: first block whose key (prefix-max end) is > q_start ----
                u32 b0[SUB], be[SUB];
                search_blocks<SUB>(a, s_lut, s_q, s_ctab, &c[r0], &s[r0], b0, be);
                GT_STAMP(t_c1);

                // ---- 2. one burst of four 16-byte loads per query: starts, ends, ids (own + look-ahead) ----
                uint4 S[SUB], E[SUB], V[SUB], L[SUB];
                bool act[SUB];
#pragma unroll
                for (int j = 0; j < SUB; ++j) {
                    act[j] = b0[j] < be[j];
                    const uint4 *rec = a.blocks + (size_t)(act[j] ? b0[j] : 0u) * 4;
                    if (!(GTARS_ABLATE & 2)) {
                        S[j] = rec[0];
                        E[j] = rec[1];
                        V[j] = rec[2];
                        L[j] = rec[3];
                    } else {
                        S[j] = make_uint4(s[r0 + j] ^ 8u, ~0u, ~0u, ~0u);
                        E[j] = make_uint4(e[r0 + j], 0, 0, 0);
                        V[j] = make_uint4((u32)j, 0, 0, 0);
                        L[j] = make_uint4(~0u, 0, 0, ~0u);
                    }
                }
                GT_STAMP(t_c2);
#pragma unroll
                for (int j = 0; j < SUB; ++j) {
                    bool mr;
                    u32 m = block_mask5<FILTER>(S[j], E[j], L[j], s[r0 + j], e[r0 + j], min_bp, mr);
                    m = act[j] ? m : 0u;
                    const bool more2 = act[j] && mr && (b0[j] + 1 < be[j]);
                    u32 n = __popc(m);
                    if (more2) n += walk_tail<FILTER>(a, b0[j], be[j], s[r0 + j], e[r0 + j], min_bp, [](u32, int) {});
                    tsum += n;
                    cur.st[r0 + j] = (b0[j] & B0_MASK) | (m << B0_BITS);
                    cur.more_bits |= (more2 ? 1u : 0u) << (r0 + j);
                    // ids of the first two hits, picked out of the ids quad (stored one tile later)
                    cur.v0[r0 + j] = (m & 1u) ? V[j].x : (m & 2u) ? V[j].y : (m & 4u) ? V[j].z : (m & 8u) ? V[j].w : L[j].z;
                    const u32 m2 = m & (m - 1u);
                    cur.v1[r0 + j] = (m2 & 2u) ? V[j].y : (m2 & 4u) ? V[j].z : (m2 & 8u) ? V[j].w : L[j].z;
                }
            }
            GT_STAMP(t_c3);

            // =============== workgroup scan of the per-thread hit counts ===============
            const u32 inc = wave_inclusive_scan_u32(tsum, lane);
            if (lane == 63) s_scan[wave] = inc;
            if (draw && threadIdx.x == 0) s_tile = gridDim.x + (ticket - ticket_base);
            lds_barrier();
            if (draw) next_tile = s_tile;
            u32 wbase = 0, block_total = 0;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                const u32 v = s_scan[w];
                if (w < wave) wbase += v;
                block_total += v;
            }
            cur.excl = wbase + inc - tsum;
            cur.total = block_total;
            cur.tile = tile;
            if (threadIdx.x == 0 && !(GTARS_ABLATE & 16)) publish_aggregate(ws->state, tile, (u64)block_total, epoch);
            GT_STAMP(t_scan);
            if (GTARS_TOK_PREFETCH && wave != 0) load_queries(next_tile);  // c/s/e are dead from here on: prefetch the next tile
        }

        // =============== resolve + write the PREVIOUS tile ===============
        if (have_prev && threadIdx.x < 64) {
            const u64 p = (GTARS_ABLATE & 16) ? (u64)prev.tile * 600u
                                              : resolve_prefix(ws->state, prev.tile, (u64)prev.total, lane, &ws->err, epoch);
            if (lane == 0) s_prefix = p;
        }
        if (GTARS_TOK_PREFETCH && has_cur && wave == 0) load_queries(next_tile);  // after the look-back: vmcnt retires in order
        if (have_prev) {
            lds_barrier();
            GT_STAMP(t_resolve);
            const u64 prefix = s_prefix;
            if (prev.tile == num_tiles - 1 && threadIdx.x == TPB - 1) {
                const u64 tot = prefix + (u64)prev.total;
                offsets[nq] = tot;
                ws->total = tot;
            }
            const u64 q0 = (u64)prev.tile * TILE + (u64)threadIdx.x * TOK_QPT;
            u64 run = prefix + prev.excl;
            u64 o4[TOK_QPT];
            // sweep 1: offsets, plus the rare paths (3rd+ hit of a record, tails).  Their loads
            // wait on vmcnt, which also counts stores, so they go BEFORE the common stores.
#pragma unroll
            for (int j = 0; j < TOK_QPT; ++j) {
                o4[j] = run;
                const u32 b0 = prev.st[j] & B0_MASK;
                u32 m = prev.st[j] >> B0_BITS;
                u64 o = run + (u64)__popc(m);
                if (!(GTARS_ABLATE & 4)) {
                    m &= m - 1;
                    m &= m - 1;  // first two hits: v0 / v1, stored in sweep 2
                    u64 o3 = run + 2;
                    while (m) {
                        const int k = __ffs((int)m) - 1;
                        m &= m - 1;
                        if (o3 < cap) ovals[o3] = blkw[b0 * 16u + val_word(k)];
                        ++o3;
                    }
                    if (prev.more_bits & (1u << j)) {
                        const u32 cq = qc[q0 + j], sq = qs[q0 + j], eq = qe[q0 + j];
                        const u32 be = s_ctab[cq].w;
                        u64 ot = o;
                        o += walk_tail<FILTER>(a, b0, be, sq, eq, min_bp, [&](u32 b, int k) {
                            if (ot < cap) ovals[ot] = blkw[b * 16u + 8u + (u32)k];
                            ++ot;
                        });
                    }
                }
                run = o;
            }
            // sweep 2: the common stores
            if (!(GTARS_ABLATE & 4)) {
#pragma unroll
                for (int j = 0; j < TOK_QPT; ++j) {
                    const u32 n = __popc(prev.st[j] >> B0_BITS);
                    if (n >= 1 && o4[j] < cap) st_stream(&ovals[o4[j]], prev.v0[j]);
                    if (n >= 2 && o4[j] + 1 < cap) st_stream(&ovals[o4[j] + 1], prev.v1[j]);
                }
            }
            if (GTARS_ABLATE & 8) {
            } else if (off_vec_ok && q0 + TOK_QPT <= nq) {
#pragma unroll
                for (int h = 0; h < TOK_QPT / 2; ++h) st_stream2(offsets + q0 + 2 * h, o4[2 * h], o4[2 * h + 1]);
            } else {
#pragma unroll
                for (int j = 0; j < TOK_QPT; ++j)
                    if (q0 + j < nq) offsets[q0 + j] = o4[j];
            }
        }
        lds_barrier();  // s_tile / s_prefix / s_scan reuse
        GT_STAMP(t_write);
        if (!has_cur) break;
        prev = cur;
        have_prev = true;
        tile = next_tile;
    }
#if GTARS_ABLATE & 128
    if (threadIdx.x == 0) {
        unsigned long long *dbg = (unsigned long long *)(ws->state + num_tiles);  // 8 spare words
        atomicAdd(&dbg[0], (unsigned long long)t_ticket);
        atomicAdd(&dbg[1], (unsigned long long)t_count);
        atomicAdd(&dbg[2], (unsigned long long)t_scan);
        atomicAdd(&dbg[3], (unsigned long long)t_resolve);
        atomicAdd(&dbg[4], (unsigned long long)t_write);
        atomicAdd(&dbg[5], 1ull);
        atomicAdd(&dbg[6], (unsigned long long)t_fill);
        atomicAdd(&dbg[7], (unsigned long long)t_c1);
        atomicAdd(&dbg[8], (unsigned long long)t_c2);
        atomicAdd(&dbg[9], (unsigned long long)t_c3);
    }
#endif
}

static size_t tok_lds_bytes(const AccelView &a) {
    return ((size_t)a.lut_words + a.q_words + 4 * (size_t)a.n_chrom) * sizeof(u32);
}

// ---------------------------------------------------------------- K2 on the same structure
// count_overlaps / any_overlaps (multi_chrom_overlapper.rs:483-517) for a Bits-kind index: the search and
// the record burst of k_tok_lds without the scan -- counts do not depend on the result order.
// 4x the generic k_count (12 vs 48 us per 1M queries): no dependent chain of binary-search loads in L2.
template <int TPB, bool FILTER>
__global__ void __launch_bounds__(TPB, 4)
k_count_lds(AccelView a, const u32 *__restrict__ qc, const u32 *__restrict__ qs, const u32 *__restrict__ qe, u64 nq,
            i32 min_bp, u32 *__restrict__ counts, u8 *__restrict__ any) {
    extern __shared__ __attribute__((aligned(16))) u32 smem[];
    constexpr int QPT = 4;
    constexpr u64 TILE = (u64)TPB * QPT;
    const u32 *s_lut = smem, *s_q = smem + a.lut_words;
    const uint4 *s_ctab = reinterpret_cast<const uint4 *>(smem + a.lut_words + a.q_words);
    fill_search_lds<TPB>(a, smem);
    __syncthreads();
    const bool vec_ok = ((((uintptr_t)qc) | ((uintptr_t)qs) | ((uintptr_t)qe)) & 15u) == 0;
    const u64 num_tiles = (nq + TILE - 1) / TILE;
    for (u64 tile = blockIdx.x; tile < num_tiles; tile += gridDim.x) {
        const u64 q0 = tile * TILE + (u64)threadIdx.x * QPT;
        u32 c[QPT], s[QPT], e[QPT];
        if (vec_ok && q0 + QPT <= nq) {
            const u32x4 c4 = ld_stream4(qc + q0), s4 = ld_stream4(qs + q0), e4 = ld_stream4(qe + q0);
            c[0] = c4.x; c[1] = c4.y; c[2] = c4.z; c[3] = c4.w;
            s[0] = s4.x; s[1] = s4.y; s[2] = s4.z; s[3] = s4.w;
            e[0] = e4.x; e[1] = e4.y; e[2] = e4.z; e[3] = e4.w;
        } else {
#pragma unroll
            for (int j = 0; j < QPT; ++j) {
                const bool ok = q0 + j < nq;
                c[j] = ok ? qc[q0 + j] : GTARS_UNKNOWN_CHROM;
                s[j] = ok ? qs[q0 + j] : 0;
                e[j] = ok ? qe[q0 + j] : 0;
            }
        }
        u32 b0[QPT], be[QPT];
        search_blocks<QPT>(a, s_lut, s_q, s_ctab, c, s, b0, be);
        uint4 S[QPT], E[QPT], L[QPT];
        bool act[QPT];
#pragma unroll
        for (int j = 0; j < QPT; ++j) {
            act[j] = b0[j] < be[j];
            const uint4 *rec = a.blocks + (size_t)(act[j] ? b0[j] : 0u) * 4;
            S[j] = rec[0];
            E[j] = rec[1];
            L[j] = rec[3];
        }
#pragma unroll
        for (int j = 0; j < QPT; ++j) {
            bool mr;
            u32 m = block_mask5<FILTER>(S[j], E[j], L[j], s[j], e[j], min_bp, mr);
            u32 n = act[j] ? __popc(m) : 0u;
            if (act[j] && mr && b0[j] + 1 < be[j]) n += walk_tail<FILTER>(a, b0[j], be[j], s[j], e[j], min_bp, [](u32, int) {});
            if (q0 + j < nq) {
                if (counts) counts[q0 + j] = n;
                if (any) any[q0 + j] = n ? 1 : 0;
            }
        }
    }
}

gtars_status launch_count_lds(const AccelView &a, const u32 *qc, const u32 *qs, const u32 *qe, u64 nq, int has_min,
                              i32 min_overlap, u32 *counts, u8 *any, hipStream_t st) {
    if (nq == 0) return GTARS_OK;
    constexpr int TPB = 1024;
    const size_t lds = tok_lds_bytes(a);
    const bool filter = has_min && min_overlap > 1;
    const i32 min_bp = has_min ? min_overlap : 0;
    auto kern = filter ? k_count_lds<TPB, true> : k_count_lds<TPB, false>;
    struct Cfg {
        size_t lds = 0;
        int dev = -1, cus = 0;
        const void *fn = nullptr;
    };
    static thread_local Cfg cfg;
    int dev = 0;
    GT_HIP(hipGetDevice(&dev));
    if (cfg.dev != dev || cfg.lds != lds || cfg.fn != (const void *)kern) {
        if (lds > 48 * 1024)
            GT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        int cus = 256;
        GT_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        cfg.dev = dev;
        cfg.lds = lds;
        cfg.cus = cus;
        cfg.fn = (const void *)kern;
    }
    const u64 tiles = (nq + (u64)TPB * 4 - 1) / ((u64)TPB * 4);
    const unsigned grid = (unsigned)std::min<u64>(tiles, (u64)cfg.cus);
    ProfScope p("k_count_lds", st);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(TPB), lds, st, a, qc, qs, qe, nq, min_bp, counts, any);
    GT_HIP(hipGetLastError());
    return GTARS_OK;
}

// ---------------------------------------------------------------- launcher

static int env_int(const char *name, int dflt) {
    const char *v = getenv(name);
    return v && *v ? atoi(v) : dflt;
}


bool tokenize_lds_supported(const AccelView &a) {
    return a.n_blocks > 0 && a.n_blocks <= ((1u << 22) - 1u) && a.n_units > 0 && tok_lds_bytes(a) <= 150 * 1024;
}

// launch geometry: threads per workgroup and queries per thread (a tile is TPB * QPT queries)
static void choose_geometry(u64 nq, int &tpb, int &qpt) {
    // One 1024-thread workgroup per CU: one LDS copy of the search keys, 16 waves.  (Two 512-thread
    // workgroups were faster only before the query / result streams became non-temporal; since then
    // 1024 wins at every batch size: 727 vs 751 us at 64M queries, 18.4 vs 21.5 us at 1M.)
    (void)nq;
    tpb = 1024;
    qpt = 4;
    const int f_q = env_int("GTARS_TOK_QPT", 0);
    if (f_q == 2 || f_q == 4) qpt = f_q;
    const int f_tpb = env_int("GTARS_TOK_TPB", 0);
    if (f_tpb == 512 || f_tpb == 1024) tpb = f_tpb;
}

size_t tokenize_lds_ws_bytes(u64 nq) {
    // sized for the smallest tile (512 threads x 2 queries)
    return scan_ws_bytes_for_tiles((nq + 512 * 2 - 1) / (512 * 2));
}

template <int TPB, int TOK_QPT, bool FILTER>
static gtars_status launch_tok_t(const AccelView &a, const u32 *qc, const u32 *qs, const u32 *qe, u64 nq,
                                 i32 min_bp, const EnumOut &out, ScanWs *ws, ScanEpoch &ep, hipStream_t st) {
    const size_t lds = tok_lds_bytes(a);
    auto kern = k_tok_lds<TPB, TOK_QPT, FILTER>;
    // occupancy / attribute queries cost tens of microseconds of host time: do them once per
    // (kernel instantiation, LDS size, device) and cache the result
    struct Cfg {
        size_t lds = 0;
        int dev = -1, per_cu = 0, cus = 0;
    };
    static thread_local Cfg cfg;
    int dev = 0;
    GT_HIP(hipGetDevice(&dev));
    if (cfg.dev != dev || cfg.lds != lds) {
        if (lds > 48 * 1024)
            GT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        int per_cu_q = 0, cus_q = 256;
        GT_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_q, kern, TPB, lds));
        if (per_cu_q < 1) return fail(GTARS_ERR_INTERNAL, "k_tok_lds does not fit on a CU");
        GT_HIP(hipDeviceGetAttribute(&cus_q, hipDeviceAttributeMultiprocessorCount, dev));
        cfg.dev = dev;
        cfg.lds = lds;
        cfg.per_cu = per_cu_q;
        cfg.cus = cus_q;
    }
    int per_cu = cfg.per_cu;
    const int cus = cfg.cus;
    const int cap_per_cu = env_int("GTARS_TOK_WG_PER_CU", 0);
    if (cap_per_cu > 0 && per_cu > cap_per_cu) per_cu = cap_per_cu;
    const u64 tile_q = (u64)TPB * TOK_QPT;
    const u64 tiles = (nq + tile_q - 1) / tile_q;
    u64 grid = (u64)cus * per_cu;
    if (grid > tiles) grid = tiles;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(TPB), lds, st, a, qc, qs, qe, nq, min_bp, out.offsets,
                       out.vals, out.vals ? out.capacity : 0, ws, ep.epoch, ep.ticket_base);
    GT_HIP(hipGetLastError());
    // tickets drawn by this launch: the tiles beyond the first `grid`, plus one failing draw per workgroup
    if (tiles > grid) ep.ticket_base += (u32)tiles;
#if GTARS_ABLATE & 128
    {
        static int printed = 0;
        if (printed++ < 2) {
            unsigned long long h[10];
            (void)hipStreamSynchronize(st);
            (void)hipMemcpy(h, (char *)ws + sizeof(u64) * (2 + tiles), sizeof h, hipMemcpyDeviceToHost);
            fprintf(stderr, "[phase cycles per WG, %llu WGs, %llu tiles] fill %.0f ticket %.0f count %.0f (search %.0f fetch+mask %.0f lookahead+ids %.0f) scan %.0f resolve %.0f write %.0f\n",
                    h[5], (unsigned long long)tiles, (double)h[6] / h[5], (double)h[0] / h[5], (double)(h[7] + h[8] + h[9]) / h[5], (double)h[7] / h[5], (double)h[8] / h[5], (double)h[9] / h[5], (double)h[2] / h[5],
                    (double)h[3] / h[5], (double)h[4] / h[5]);
        }
    }
#endif
    return GTARS_OK;
}

gtars_status launch_tokenize_lds(const AccelView &a, const u32 *qc, const u32 *qs, const u32 *qe, u64 nq,
                                 int has_min, i32 min_overlap, const EnumOut &out, void *scan_ws,
                                 size_t scan_ws_bytes, ScanEpoch &ep, hipStream_t st) {
    if (out.starts || out.ends) return fail(GTARS_ERR_INTERNAL, "k_tok_lds writes vals only");
    if (nq == 0) {
        GT_HIP(hipMemsetAsync(out.offsets, 0, sizeof(u64), st));
        GT_HIP(hipMemsetAsync(&((ScanHead *)scan_ws)->total, 0, sizeof(u64), st));
        return GTARS_OK;
    }
    int tpb, qpt;
    choose_geometry(nq, tpb, qpt);
    const u64 tile_q = (u64)tpb * qpt;
    const u64 tiles = (nq + tile_q - 1) / tile_q;
    if (tiles > 0xFFFFFFF0ull) return fail(GTARS_ERR_INVALID_ARG, "query batch too large for one launch");
    const size_t need = scan_ws_bytes_for_tiles(tiles);
    if (scan_ws_bytes < need) return fail(GTARS_ERR_INTERNAL, "fused scan workspace too small");
    // The workspace is cleared only when it is new (ep.epoch == 0), when it has to grow past what
    // was cleared, or when the 14-bit epoch wraps; otherwise stale granules are told apart by epoch.
    if (ep.epoch == 0 || ep.epoch >= EP_MAX || need > ep.cleared_bytes) {
        GT_HIP(hipMemsetAsync(scan_ws, 0, scan_ws_bytes, st));
        ep.cleared_bytes = scan_ws_bytes;
        ep.epoch = 0;
        ep.ticket_base = 0;
    }
    ep.epoch += 1;
    const bool filter = has_min && min_overlap > 1;
    const i32 min_bp = has_min ? min_overlap : 0;
    ScanWs *ws = (ScanWs *)scan_ws;
    ProfScope p("k_tok_lds", st);
#define GT_TOK_CASE(T, Q)                                                                     \
    if (tpb == T && qpt == Q)                                                                 \
        return filter ? launch_tok_t<T, Q, true>(a, qc, qs, qe, nq, min_bp, out, ws, ep, st)      \
                      : launch_tok_t<T, Q, false>(a, qc, qs, qe, nq, min_bp, out, ws, ep, st);
    GT_TOK_CASE(512, 4)
    GT_TOK_CASE(1024, 4)
    GT_TOK_CASE(1024, 2)
    GT_TOK_CASE(512, 2)  // experiments (GTARS_TOK_QPT): 8 queries per thread spills, 2 under-uses the bursts
#undef GT_TOK_CASE
    return fail(GTARS_ERR_INTERNAL, "unsupported tokenizer launch geometry");
}

}  // namespace gtars
