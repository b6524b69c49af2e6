// tokenize_lds.hip -- the fast path of Tokenizer::tokenize / encode
// (gtars-tokenizers/src/tokenizer.rs:140-171 -> Bits::find, bits.rs:141-156,
// 433-446) and of count_overlaps / any_overlaps for a Bits-kind index.
//
// Data layout (AccelView, common.h): the sorted index is cut into blocks of 2 intervals; a block's record holds its
// own 2 intervals plus copies of the NEXT block's 2 (the look-ahead): starts | ends, 32 bytes, when token ids follow
// from the position (sorted universes); a 64-byte slot with an id quad otherwise.  One u16 search key per block (or
// per unit of 2^shift blocks) and a bucket table over the key space live in LDS.
//
// Per query:
//   1. LDS search for the first block that can hold a hit: the first whose key --
//      the largest end among all intervals up to and including the block -- is
//      > q_start.  (Bits::find starts at lower_bound(q_start - max_len),
//      bits.rs:144-147; every interval between that point and ours has
//      end <= q_start, so the hit set and its order are the same.)
//   2. ONE burst of two (three) 16-byte loads from the block's record; the overlap
//      test runs in registers -> 4-bit hit mask.  Only a query that reaches past BOTH
//      look-ahead intervals (iv.start >= stop ends the reference scan,
//      bits.rs:441-443) walks into the following blocks.
//   3. hit counts are scanned inside the wave, across the waves of a workgroup and
//      across workgroups (chained scan, scan.h);
//   4. CSR offsets (u64) are written in place; token ids (u32) are compacted in a
//      per-wave LDS buffer and leave as contiguous 256-byte stores.
//
// What bounds it (tools/ubench/ta.hip, 16 waves per CU): a CU's vector-memory path.
// A divergent 16-byte request costs 2.3 clk per lane, further 16-byte pieces of the same
// record ~1 clk each, a scattered dword store 2.5 clk per lane, while a contiguous
// wave-wide load or store of any width costs ~24 clk per instruction.  Hence 32-byte
// records (2.4 clk per query instead of 4.0 for 64 bytes) and LDS-compacted id stores
// (0.3 clk per query instead of 1.5).  No MFMA: integer search.
#include "common.h"
#include "scan.h"

#include <mutex>

namespace gtars {

// timing experiments only (tools/build_variant.sh; results are then WRONG by construction):
//   1 no look-back (a made-up base)   2 no id stores   4 no offset stores   8 no record burst (made-up records)
//   16 no LDS search (a made-up block)   32 no LDS fill (the search runs on whatever the LDS holds)
//   64 wide queries' wave-wide stores left out (coop_runs runs)   128 coop_runs left out
#ifndef GTARS_TOK_ABLATE
#define GTARS_TOK_ABLATE 0
#endif

__device__ __forceinline__ i64 overlap_bp_tok(u32 as, u32 ae, u32 bs, u32 be) {
    u32 mn = ae < be ? ae : be;
    u32 mx = as > bs ? as : bs;
    return (i64)mn - (i64)mx;
}

// The query stream in and the offsets out are touched once: non-temporal, so that they do not push
// the index out of the XCD's L2 while a launch runs.
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
typedef u32 u32x2 __attribute__((ext_vector_type(2)));
typedef u64 u64x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u32x4 ld_stream4(const u32 *p) { return __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p)); }
__device__ __forceinline__ u32x2 ld_stream2(const u32 *p) { return __builtin_nontemporal_load(reinterpret_cast<const u32x2 *>(p)); }
__device__ __forceinline__ void st_stream2(u64 *p, u64 a, u64 b) {
    u64x2 v = {a, b};
    __builtin_nontemporal_store(v, reinterpret_cast<u64x2 *>(p));
}

// diagnostic build (tools/build_variant.sh tokstamps "-DGTARS_TOK_STAMPS=1", tools/r03_tok_stamps.py): shader-clock totals per
// phase of the tile loop, for wave 0 (the look-back wave) and wave 1 of every workgroup, and the look-back's round counts
#ifndef GTARS_TOK_FILL_U
#define GTARS_TOK_FILL_U 4  // unroll factor of the copy of the search structure into LDS (16-byte loads in flight per thread)
#endif
#ifndef GTARS_TOK_PRIO
#define GTARS_TOK_PRIO 0  // 1: s_setprio 1 during the count phase, 2: during the write phase (A/B)
#endif
#ifndef GTARS_TOK_LB_W1
#define GTARS_TOK_LB_W1 1  // look-back windows per round in one-tile-per-group launches (see resolve_prefix_helping)
#endif
#ifndef GTARS_TOK_KEYS_U16
#define GTARS_TOK_KEYS_U16 1  // the in-bucket keys by eight 2-byte reads instead of one 16-byte read at a 2-byte-aligned address (1-2 % at the 100k universe)
#endif
#ifndef GTARS_TOK_SEARCH8
#define GTARS_TOK_SEARCH8 1  // in-bucket search by one 16-byte read of 8 keys instead of halving steps
#endif
#ifndef GTARS_TOK_STAMPS
#define GTARS_TOK_STAMPS 0
#endif
#ifndef GTARS_TOK_UNIT2_EXACT
#define GTARS_TOK_UNIT2_EXACT 1  // 1: the in-unit step of two-block units reads the first block.s key (a dependent load per query); 0: round-4 experiment, slower (profiles/r04)
#endif
#if GTARS_TOK_STAMPS
__device__ unsigned long long g_tok_stamps[2][12];
#define TSTAMP(k)                                         \
    do {                                                  \
        const u64 _t = __builtin_amdgcn_s_memtime();      \
        ts_acc[k] += _t - ts_last;                        \
        ts_last = _t;                                     \
    } while (0)
#else
#define TSTAMP(k) \
    do {          \
    } while (0)
#endif

template <bool FILTER>
__device__ __forceinline__ bool hit_test(u32 s, u32 e, u32 qs, u32 qe, i32 min_bp) {
    bool hit = (s < qe) & (e > qs);  // Interval::overlap, interval.rs:47-50
    if (FILTER) hit = hit && overlap_bp_tok(qs, qe, s, e) >= (i64)min_bp;
    return hit;
}

// 4-bit hit mask of a record: the two own intervals (bits 0, 1) and the two look-ahead intervals (bits 2, 3)
template <bool FILTER>
__device__ __forceinline__ u32 block_mask4(const uint4 &S, const uint4 &E, u32 qs, u32 qe, i32 min_bp) {
    u32 m = 0;
    m |= (hit_test<FILTER>(S.x, E.x, qs, qe, min_bp) ? 1u : 0u);
    m |= (hit_test<FILTER>(S.y, E.y, qs, qe, min_bp) ? 1u : 0u) << 1;
    m |= (hit_test<FILTER>(S.z, E.z, qs, qe, min_bp) ? 1u : 0u) << 2;
    m |= (hit_test<FILTER>(S.w, E.w, qs, qe, min_bp) ? 1u : 0u) << 3;
    return m;
}

// Per-query state kept between the count phase and the write phase, in ONE
// register: first block (22 bits) + 4-bit hit mask.
constexpr u32 B0_BITS = 22;
constexpr u32 B0_MASK = (1u << B0_BITS) - 1u;
// ... or, for a wide query in run form (tail_run; ids that follow from the position): its hit count n < 2^28 (low 22 bits | high 6 bits
// above the mask) + the mask -- a query's hits are bounded by the index's 2 * (2^22 - 1) intervals (tokenize_lds_supported)
__device__ __forceinline__ u32 run_state(u32 n, u32 m) { return (n & B0_MASK) | (m << B0_BITS) | ((n >> B0_BITS) << (B0_BITS + 4)); }
#ifndef GTARS_TOK_STAGE_RUNS
#define GTARS_TOK_STAGE_RUNS 1  // 1: the early staging of round 0 (stage_queries) emits run-form queries itself; 0: see there
#endif
// TileQ::more_bits: bit j: query j's scan runs past its first record; bit RUN_Q_BIT + j: it is in run form (its state word holds
// the run's length, not the block); 2 bits at STAB_W_BIT + 2j / 4 bits at STAB_N_BIT + 4j: records tested in front of the run
// (stab_walk) and their hits
constexpr u32 RUN_Q_BIT = 4, STAB_W_BIT = 8, STAB_N_BIT = 16;
__device__ __forceinline__ u32 run_state_n(u32 st) { return (st & B0_MASK) | ((st >> (B0_BITS + 4)) << B0_BITS); }

// Tail of a query whose scan runs past block b0's look-ahead intervals (rare).  Record b holds intervals
// ACC_OWN * b .. ACC_OWN * b + 3, so the walk goes on with records b0 + 2, b0 + 4, ... and uses all four slots.
// STRIDE = quads per record.  f(block, k) is called for every hit in scan order; returns the hit count.
template <bool FILTER, u32 STRIDE, class F>
__device__ __forceinline__ u32 walk_tail(const uint4 *__restrict__ recs, u32 b0, u32 be, u32 qs, u32 qe, i32 min_bp, F &&f) {
    static_assert(ACC_OWN == 2, "a record = two own intervals + two look-ahead intervals");
    u32 n = 0;
    bool mr = true;
    for (u32 b = b0 + 2; mr && b < be; b += 2) {
        const uint4 S = recs[(size_t)b * STRIDE], E = recs[(size_t)b * STRIDE + 1];
        u32 m = block_mask4<FILTER>(S, E, qs, qe, min_bp);
        mr = S.w < qe;  // starts ascend: an interval that starts at or after q_end ends the scan (bits.rs:441-443)
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
// IN_UNIT = false (unit records, AccelView::rec8): a two-block unit is not resolved to its block -- the caller reads the unit's record
template <int SUB, bool IN_UNIT = true>
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
#if GTARS_TOK_SEARCH8
    // The bucket's keys ascend, so "first key >= target" = number of leading keys below it.  A bucket holds a handful of
    // units (the table is as fine as the LDS budget allows: ~3 units per bucket for the 100k universe): ONE 16-byte LDS read
    // of the 8 keys at its start and a chain of eight compares replace the stepwise search (6 dependent LDS round trips
    // before).  Larger buckets are first narrowed by halving steps.  Keys past the bucket's end belong to the next bucket and
    // restart low, so only the LEADING run counts: c_i = c_(i-1) & (key_i < target), and the sum is clamped to the bucket.
    (void)last;
#if GTARS_TOK_KEYS_U16
    // (the eight keys by EIGHT 2-byte reads, all queries' reads in flight together: a 16-byte LDS read at a 2-byte-aligned address
    // is not a fast path -- found on k_igd_route, whose 4-byte reads at odd 2-byte offsets cost a fifth of that kernel)
    u32 lo_b[SUB], n[SUB], kk[SUB][8];
#pragma unroll
    for (int j = 0; j < SUB; ++j) {
        lo_b[j] = pos[j] + 2u, n[j] = (last[j] - pos[j]) >> 1;  // byte address of the bucket's first key, keys in the bucket
        while (n[j] > 8u) {
            const u32 half = n[j] >> 1, mid = lo_b[j] + (half << 1);
            const bool below = (u32) * (lds_cu16)(uintptr_t)mid < tq[j];
            lo_b[j] = below ? mid + 2u : lo_b[j];
            n[j] = below ? n[j] - half - 1u : half;
        }
    }
#pragma unroll
    for (int j = 0; j < SUB; ++j)
        asm volatile("ds_read_u16 %0, %8\n\tds_read_u16 %1, %8 offset:2\n\tds_read_u16 %2, %8 offset:4\n\tds_read_u16 %3, %8 offset:6\n\t"
                     "ds_read_u16 %4, %8 offset:8\n\tds_read_u16 %5, %8 offset:10\n\tds_read_u16 %6, %8 offset:12\n\tds_read_u16 %7, %8 offset:14"
                     : "=&v"(kk[j][0]), "=&v"(kk[j][1]), "=&v"(kk[j][2]), "=&v"(kk[j][3]), "=&v"(kk[j][4]), "=&v"(kk[j][5]), "=&v"(kk[j][6]),
                       "=&v"(kk[j][7])
                     : "v"(lo_b[j])
                     : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int j = 0; j < SUB; ++j) {
        asm volatile("" : "+v"(kk[j][0]), "+v"(kk[j][1]), "+v"(kk[j][2]), "+v"(kk[j][3]), "+v"(kk[j][4]), "+v"(kk[j][5]), "+v"(kk[j][6]),
                     "+v"(kk[j][7]));  // (the keys are only valid behind the wait)
        u32 cnt = 0;
        bool run = true;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            run = run && kk[j][i] < tq[j];
            cnt += run ? 1u : 0u;
        }
        pos[j] = lo_b[j] - 2u + (min(cnt, n[j]) << 1);
    }
#else
#pragma unroll
    for (int j = 0; j < SUB; ++j) {
        u32 lo_b = pos[j] + 2u, n = (last[j] - pos[j]) >> 1;  // byte address of the bucket's first key, keys in the bucket
        while (n > 8u) {
            const u32 half = n >> 1, mid = lo_b + (half << 1);
            const bool below = (u32) * (lds_cu16)(uintptr_t)mid < tq[j];
            lo_b = below ? mid + 2u : lo_b;
            n = below ? n - half - 1u : half;
        }
        typedef unsigned short us8 __attribute__((ext_vector_type(8)));
        typedef us8 us8_a2 __attribute__((aligned(2)));  // any key may start a bucket: a 2-byte-aligned 16-byte LDS read
        typedef const __attribute__((address_space(3))) us8_a2 *lds_k8;
        const us8 kk = *(lds_k8)(uintptr_t)lo_b;  // (the key array is padded by 8 entries)
        u32 cnt = 0;
        bool run = true;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            run = run && (u32)kk[i] < tq[j];
            cnt += run ? 1u : 0u;
        }
        pos[j] = lo_b - 2u + (min(cnt, n) << 1);  // &q[first key >= target] - 2, what the stepwise search leaves
    }
#endif
#else
    for (u32 step = a.search_top << 1; step >= 2; step >>= 1) {  // byte steps
#pragma unroll
        for (int j = 0; j < SUB; ++j) {
            const u32 cand = min(pos[j] + step, last[j]);
            const u32 v = *(lds_cu16)(uintptr_t)cand;
            pos[j] = v < tq[j] ? cand : pos[j];
        }
    }
#endif
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
                } else if (shift == 1) {
                    // two blocks per unit: the unit's last key is > q_start already (that is how the unit was found), so the
                    // first block's key decides -- ONE dependent load (a halving loop took two, the first of them for the key
                    // whose answer is known).  Round 4 tried NO in-unit step (start at the unit's first block: exact, because
                    // every interval in front of the unit ends at or before q_start, and the dead intervals just fail the overlap
                    // test): slower -- 941 vs 731 us per 64M queries at 140k regions -- because a first candidate in the record's
                    // later slots sends the scan into walk_tail, whose dependent loads cost more than the key load saved.
#if GTARS_TOK_UNIT2_EXACT
                    if constexpr (IN_UNIT) b += a.blk_first[b] <= s[j] ? 1u : 0u;
#endif
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

// LDS image of the search structure: bucket table (u16) | unit keys (u16) | chromosome table (16 B each) |
// id corrections (u32 each, padded to 16 B); both key arrays are padded to 16 bytes
struct SearchLds {
    const u32 *lut, *q;
    const uint4 *ctab;
    const u32 *idc;
};
__device__ __forceinline__ SearchLds search_lds_view(const AccelView &a, u32 *smem) {
    SearchLds v;
    v.lut = smem;
    v.q = smem + a.lut_words;
    v.ctab = reinterpret_cast<const uint4 *>(smem + a.lut_words + a.q_words);
    v.idc = smem + a.lut_words + a.q_words + 4 * a.n_chrom;
    return v;
}
__host__ __device__ __forceinline__ size_t tok_lds_bytes(const AccelView &a) {
    return ((size_t)a.lut_words + a.q_words + 4 * (size_t)a.n_chrom + (((size_t)a.n_chrom + 3) & ~(size_t)3)) * sizeof(u32);
}

// Tail of a wide query in RUN FORM (run_form below; AccelView::runs_ok / ends_mono): the query's hits are one run of stored
// positions, so the tail is MEASURED, not walked -- a second LDS search, for q_end, finds the first block whose key (the largest
// end so far) is >= q_end: every interval in front of it ends, hence starts (no interval is inverted), before q_end; from there
// the blocks' starts give the run's end.  On a universe of disjoint intervals that is the block found or its neighbour; a wide
// interval in front makes the key run ahead of the starts, and the run's end lies further on: the blocks are stepped through,
// one 16-byte load each (never more steps than the tail has blocks -- what the walk would load twice over -- and the write
// phase still needs none).
// Returns the number of intervals at padded position >= ACC_OWN * b0 + 4 that start before q_end -- what walk_tail counts for
// such a query.  33 ids per query: the walk was 8 dependent record loads per lane, twice (count phase and write phase).
template <u32 STRIDE>
__device__ __forceinline__ u32 tail_run(const AccelView &a, const SearchLds &L, const uint4 *__restrict__ recs, u32 c, u32 b0, u32 be,
                                        u32 qe, u32 skip = 0) {
    // (skip: records behind the first one that the caller has tested interval by interval -- the run starts behind them)
    const u32 sq = qe - 1u;  // (a query with a tail has q_end > the fourth start >= 0)
    u32 B, BE;
    search_blocks<1>(a, L.lut, L.q, L.ctab, &c, &sq, &B, &BE);
    const u32 bmin = b0 + 2u + 2u * skip;  // the block whose own two intervals are the run's first
    B = B > bmin ? B : bmin;
    u32 last;
    for (;;) {
        if (B >= be) {  // q_end lies beyond the chromosome's largest end
            last = a.chrom_iv_end[c];
            break;
        }
        const uint4 S = recs[(size_t)B * STRIDE];
        if (S.x == 0xFFFFFFFFu) {  // a padding block
            last = a.chrom_iv_end[c];
            break;
        }
        if (S.z >= qe) {  // the run ends inside this block's own two (sentinels never count)
            last = (u32)ACC_OWN * B + (S.x < qe ? 1u : 0u) + (S.y < qe ? 1u : 0u);
            break;
        }
        ++B;  // (floor-quantised keys, or a key that runs ahead of the starts)
    }
    const u32 first_tail = (u32)ACC_OWN * bmin;
    return last > first_tail ? last - first_tail : 0u;
}
// The records a query's scan crosses before its run starts (RUNS builds): behind the first record, every record whose FIRST
// interval still starts at or before q_start is tested interval by interval (a long interval in front -- its end is why the scan
// starts this early -- and short ones that end before q_start: hits with gaps); the first record that starts behind q_start
// opens the run.  At most STAB_MAX records (their number and their hits travel in TileQ::more_bits); a longer shadow is walked.
constexpr u32 STAB_MAX = 3;
struct Stab {
    u32 w, hits;       // records tested, hits among their intervals
    bool ok, ended;    // within STAB_MAX; the scan ended inside them (no run behind)
};
template <u32 STRIDE, class F>
__device__ __forceinline__ Stab stab_walk(const uint4 *__restrict__ recs, u32 b0, u32 be, u32 qs, u32 qe, F &&f) {
    Stab r{0u, 0u, true, false};
    for (u32 b = b0 + 2u;; b += 2u) {
        if (b >= be) {
            r.ended = true;
            break;
        }
        const uint4 S = recs[(size_t)b * STRIDE];
        if (S.x > qs) break;  // everything from here on starts, hence ends, behind q_start
        if (r.w == STAB_MAX) {
            r.ok = false;
            break;
        }
        const uint4 E = recs[(size_t)b * STRIDE + 1];
        u32 m = block_mask4<false>(S, E, qs, qe, 0);
        r.hits += __popc(m);
        while (m) {
            const int k = __ffs((int)m) - 1;
            m &= m - 1;
            f(r.w, k);
        }
        ++r.w;
        if (!(S.w < qe)) {  // (bits.rs:441-443: the scan ends at the first start >= q_end)
            r.ended = true;
            break;
        }
    }
    return r;
}
// Is a query with a tail, first-record hit mask m and fourth start s3 in run form?  Its first record's hits must reach up to the
// fourth interval without a gap (so that they and the tail are ONE run of ids), and every interval of the tail must end after
// q_start: because the fourth interval already starts after q_start (any universe without inverted intervals), or because
// the ends ascend with the starts and the record has a hit (disjoint universes).
__device__ __forceinline__ bool mask_joins_tail(u32 m) { return m && (m + (m & (0u - m))) == 16u; }
__device__ __forceinline__ bool run_form(const AccelView &a, u32 m, u32 s3, u32 qs) {
    return a.runs_ok && mask_joins_tail(m) && (s3 > qs || a.ends_mono);
}
#ifndef GTARS_TOK_RUNS
#define GTARS_TOK_RUNS 5  // 1: tails of wide queries measured (tail_run), 4: their ids leave by wave-wide stores (experiments: subsets)
#endif
constexpr u32 COOP_MIN = 16;  // ids of one query from which on they leave by wave-wide stores (write_queries)
constexpr u64 WIDE_IDS_PER_QUERY = 4;  // id slots per query from which on a launch gets the kernels with the run form (launch_tokenize_lds)

#ifndef GTARS_TOK_FILL_DMA
#define GTARS_TOK_FILL_DMA 0  // 1: the LDS image is filled by global_load_lds_dwordx4 (round-4 experiment: no gain, see profiles/r04); 0: loads + ds_write_b128
#endif
// The two key arrays by LDS-DMA: one wave-instruction moves 64 consecutive 16-byte vectors (the LDS destination is a wave-uniform
// base + lane x 16, the source address is per lane), no VGPR in between and none of the 13 LDS cycles a ds_write_b128 costs.
// Waves take 1-KiB pieces round-robin, starting at a different piece in every workgroup.  The caller waits (vmcnt) and barriers.
template <int TPB>
__device__ __forceinline__ void fill_search_lds_dma(const AccelView &a, u32 *smem) {
    constexpr u32 NW = TPB / 64;
    const u32 lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const u32 n4a = a.lut_words >> 2, n4b = a.q_words >> 2;
    const u32 pa = (n4a + 63u) >> 6, pb = (n4b + 63u) >> 6, np = pa + pb;  // 1-KiB pieces of the two arrays
    const u32 rot = (u32)(((u64)blockIdx.x * 2654435761ull) % np);
    for (u32 i = wave; i < np; i += NW) {
        u32 p = i + rot;
        p = p >= np ? p - np : p;
        const bool second = p >= pa;
        const u32 piece = second ? p - pa : p, n4 = second ? n4b : n4a, v = (piece << 6) + lane;
        const u32 *src = (second ? a.qkeys : a.lut) + (size_t)v * 4;
        u32 *dst = smem + (second ? a.lut_words : 0u) + ((size_t)piece << 8);  // (wave-uniform: the piece's first word)
        if (v < n4)
            __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
    }
    uint4 *s_ctab = reinterpret_cast<uint4 *>(smem + a.lut_words + a.q_words);
    u32 *s_idc = smem + a.lut_words + a.q_words + 4 * a.n_chrom;
    for (u32 i = threadIdx.x; i < a.n_chrom; i += TPB) {
        s_ctab[i] = a.chrom_tab[i];
        s_idc[i] = a.idc[i];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the DMA writes count as vector-memory operations)
}

template <int TPB>
__device__ __forceinline__ void fill_search_lds(const AccelView &a, u32 *smem) {
#if GTARS_TOK_FILL_DMA
    return fill_search_lds_dma<TPB>(a, smem);
#endif
    // 16-byte loads, GTARS_TOK_FILL_U in flight.  Every workgroup copies the same arrays: start each one at a
    // different place so that they do not all queue on the same L2 channel at the same time.
    // (A hand-batched form with the loads of a whole batch held in a register array went to scratch -- 176 bytes per lane --
    // and cost 6-10 us per launch; the compiler's own unrolling of this loop keeps them in registers.)
    const u32 n4a = a.lut_words >> 2, n4 = n4a + (a.q_words >> 2);
    const uint4 *src_a = reinterpret_cast<const uint4 *>(a.lut);
    const uint4 *src_b = reinterpret_cast<const uint4 *>(a.qkeys);
    uint4 *dst = reinterpret_cast<uint4 *>(smem);
    const u32 rot = (u32)(((u64)blockIdx.x * 2654435761ull) % n4);
#pragma unroll GTARS_TOK_FILL_U
    for (u32 i = threadIdx.x; i < n4; i += TPB) {
        u32 k = i + rot;
        k = k >= n4 ? k - n4 : k;
        dst[k] = k < n4a ? src_a[k] : src_b[k - n4a];
    }
    uint4 *s_ctab = reinterpret_cast<uint4 *>(smem + a.lut_words + a.q_words);
    u32 *s_idc = smem + a.lut_words + a.q_words + 4 * a.n_chrom;
    for (u32 i = threadIdx.x; i < a.n_chrom; i += TPB) {
        s_ctab[i] = a.chrom_tab[i];
        s_idc[i] = a.idc[i];
    }
}

// What a lane keeps about its QPT queries between the count phase and the write phase.
// IMPL (ids follow from the position: id = ACC_OWN * block + slot + idc[chrom]): aux[j] = ACC_OWN * b0 + idc;
// otherwise aux[2j], aux[2j+1] = ids of the first two hits, picked out of the record's id quad.
template <int QPT, bool IMPL>
struct TileQ {
    u32 st[QPT];                    // b0 | mask4 << 22
    u32 aux[IMPL ? QPT : 2 * QPT];
    u32 excl;                       // exclusive hit offset of the lane's first query inside its wave's part
    u32 wtotal;                     // hits of the wave's 64 * QPT queries
    u32 more_bits;                  // bit j: query j runs past its first block's look-ahead
};

template <int QPT>
__device__ __forceinline__ void load_queries(const u32 *__restrict__ qc, const u32 *__restrict__ qs, const u32 *__restrict__ qe,
                                             u64 nq, u64 q0, bool vec_ok, u32 (&c)[QPT], u32 (&s)[QPT], u32 (&e)[QPT]) {
    if (vec_ok && q0 + QPT <= nq) {
        if constexpr (QPT == 4) {
            const u32x4 c4 = ld_stream4(qc + q0), s4 = ld_stream4(qs + q0), e4 = ld_stream4(qe + q0);
            c[0] = c4.x; c[1] = c4.y; c[2] = c4.z; c[3] = c4.w;
            s[0] = s4.x; s[1] = s4.y; s[2] = s4.z; s[3] = s4.w;
            e[0] = e4.x; e[1] = e4.y; e[2] = e4.z; e[3] = e4.w;
        } else if constexpr (QPT == 2) {
            const u32x2 c2 = ld_stream2(qc + q0), s2 = ld_stream2(qs + q0), e2 = ld_stream2(qe + q0);
            c[0] = c2.x; c[1] = c2.y;
            s[0] = s2.x; s[1] = s2.y;
            e[0] = e2.x; e[1] = e2.y;
        } else {
            c[0] = __builtin_nontemporal_load(qc + q0);
            s[0] = __builtin_nontemporal_load(qs + q0);
            e[0] = __builtin_nontemporal_load(qe + q0);
        }
    } else {
#pragma unroll
        for (int j = 0; j < QPT; ++j) {
            const bool ok = q0 + j < nq;
            c[j] = ok ? qc[q0 + j] : GTARS_UNKNOWN_CHROM;
            s[j] = ok ? qs[q0 + j] : 0;
            e[j] = ok ? qe[q0 + j] : 0;
        }
    }
}

// Count phase of a lane's R rounds of QPT = 4 consecutive queries: search, record burst, hit masks; tsum[r] = the lane's
// hits in round r.  Software-pipelined over UNITS of two queries: unit u's LDS search runs while unit u-1's record loads
// are in flight, and unit u-1's masks are taken while unit u's loads are (searching all four queries of a round first
// and then waiting for all eight loads leaves the CU's vector-memory pipe idle during every search: 586 -> 555 us per 64M
// queries for the two-unit split alone).
// REV: the ids kept for the write phase are those of the LAST two hits (they are emitted first)
// U64 (unit records, AccelView::rec8: two-block units, ids that follow from the position, no run form): the burst is the UNIT's
// 64-byte record -- eight intervals, an 8-bit mask, the state word's block = the unit's first block -- and the in-unit key read is gone
template <int R, int QPT, bool FILTER, bool IMPL, bool REV, int RUNS, bool U64 = false>
__device__ __forceinline__ void count_rounds(const AccelView &a, const SearchLds &L, const u32 (&c)[R][QPT], const u32 (&s)[R][QPT],
                                             const u32 (&e)[R][QPT], i32 min_bp, TileQ<QPT, IMPL> (&t)[R], u32 (&tsum)[R]) {
    static_assert(QPT == 4, "two units of two queries per round");
    static_assert(!U64 || (IMPL && RUNS == 0 && R == 1), "unit records: position ids, the narrow build, one round");
    constexpr u32 STRIDE = IMPL ? 2 : 4;
    constexpr int UQ = 2, NU = R * QPT / UQ;  // queries per unit, units
    const uint4 *__restrict__ recs = IMPL ? a.rec2 : a.rec4;
    u32 b0[2][UQ], be[2][UQ];
    uint4 S[2][UQ], E[2][UQ], V[2][IMPL ? 1 : UQ];
    uint4 S2[2][U64 ? UQ : 1], E2[2][U64 ? UQ : 1];  // (unit records: slots 4-7)
    bool act[2][UQ];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        tsum[r] = 0;
        t[r].more_bits = 0;
    }
    u32 pend = 0;  // bit r * QPT + j: a wide query in run form whose tail is still to be measured
    auto issue = [&](int u) {  // search + record loads of unit u
        const int r = u / 2, j0 = (u & 1) * UQ, p = u & 1;
        if (GTARS_TOK_ABLATE & 16) {
#pragma unroll
            for (int k = 0; k < UQ; ++k) {
                b0[p][k] = (s[r][j0 + k] * 2654435761u) % a.n_blocks;
                be[p][k] = c[r][j0 + k] < a.n_chrom ? a.n_blocks : 0u;
            }
        } else {
            search_blocks<UQ, !U64>(a, L.lut, L.q, L.ctab, c[r] + j0, s[r] + j0, b0[p], be[p]);
            if (GTARS_TOK_ABLATE & 32) {  // the LDS image was not filled: keep the made-up blocks inside the record array
#pragma unroll
                for (int k = 0; k < UQ; ++k) {
                    b0[p][k] %= a.n_blocks;
                    be[p][k] = a.n_blocks;
                }
            }
        }
#pragma unroll
        for (int k = 0; k < UQ; ++k) {
            act[p][k] = b0[p][k] < be[p][k];
            if constexpr (U64) {
                const uint4 *ur = a.rec8 + (size_t)((act[p][k] ? b0[p][k] : 0u) >> 1) * 4;
                S[p][k] = ur[0];
                S2[p][k] = ur[1];
                E[p][k] = ur[2];
                E2[p][k] = ur[3];
                continue;
            }
            const uint4 *rec = recs + (size_t)(act[p][k] ? b0[p][k] : 0u) * STRIDE;
            if (GTARS_TOK_ABLATE & 8) {
                S[p][k] = make_uint4(s[r][j0 + k] ^ 8u, ~0u, ~0u, ~0u);
                E[p][k] = make_uint4(e[r][j0 + k], 0, 0, 0);
                if constexpr (!IMPL) V[p][k] = make_uint4((u32)k, 0, 0, 0);
            } else {
                S[p][k] = rec[0];
                E[p][k] = rec[1];
                if constexpr (!IMPL) V[p][k] = rec[2];
            }
        }
    };
    auto finish = [&](int u) {  // masks, tails and write-phase state of unit u
        const int r = u / 2, j0 = (u & 1) * UQ, p = u & 1;
#pragma unroll
        for (int k = 0; k < UQ; ++k) {
            const int j = j0 + k;
            const u32 qs_ = s[r][j], qe_ = e[r][j];
            u32 m = block_mask4<FILTER>(S[p][k], E[p][k], qs_, qe_, min_bp);
            if constexpr (U64) m |= block_mask4<FILTER>(S2[p][k], E2[p][k], qs_, qe_, min_bp) << 4;
            m = act[p][k] ? m : 0u;
            const bool more = U64 ? act[p][k] && (S2[p][k].w < qe_) && (b0[p][k] + 4 < be[p][k])
                                  : act[p][k] && (S[p][k].w < qe_) && (b0[p][k] + 2 < be[p][k]);
            u32 n = __popc(m);
            if (more) {
                if ((RUNS & 1) && !FILTER && a.runs_ok)  // run form: the tail is measured below, by ONE copy of the code (bit 16 + ..:
                    pend |= (1u | (run_form(a, m, S[p][k].w, qs_) ? 0x10000u : 0u)) << (r * QPT + j);  // nothing to test in front)
                else  // (unit records: the walk goes on behind the record's eight intervals, at block b0 + 4)
                    n += walk_tail<FILTER, STRIDE>(recs, b0[p][k] + (U64 ? 2u : 0u), be[p][k], qs_, qe_, min_bp, [](u32, int) {});
            }
            tsum[r] += n;
            t[r].st[j] = (b0[p][k] & B0_MASK) | (m << B0_BITS);
            t[r].more_bits |= (more ? 1u : 0u) << j;
            if constexpr (IMPL) {
                t[r].aux[j] = (u32)ACC_OWN * b0[p][k] + L.idc[act[p][k] ? c[r][j] : 0u];
            } else {
                const uint4 v = V[p][k];
                if (REV) {
                    t[r].aux[2 * j] = (m & 8u) ? v.w : (m & 4u) ? v.z : (m & 2u) ? v.y : v.x;
                    const u32 m2 = m ? m & ~(0x80000000u >> __clz(m)) : 0u;  // without its highest bit
                    t[r].aux[2 * j + 1] = (m2 & 4u) ? v.z : (m2 & 2u) ? v.y : v.x;
                } else {
                    t[r].aux[2 * j] = (m & 1u) ? v.x : (m & 2u) ? v.y : (m & 4u) ? v.z : v.w;
                    const u32 m2 = m & (m - 1u);
                    t[r].aux[2 * j + 1] = (m2 & 2u) ? v.y : (m2 & 4u) ? v.z : v.w;
                }
            }
        }
    };
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        issue(u);
        __builtin_amdgcn_sched_barrier(0);  // keep the order: search and loads of u, THEN the wait for u - 1
        if (u > 0) finish(u - 1);
    }
    finish(NU - 1);
    // Wide queries on an index whose ends ascend with the starts (run form): a lane measures its tails one after the other, the
    // query picked out of the register arrays by select chains -- inlined into finish() above, tail_run's search came eight times
    // per kernel and pushed the two-round kernels into scratch memory (60 bytes per lane; 64M C2 queries 578 -> 627 us).
    if constexpr (!FILTER && (RUNS & 1) != 0) {
        static_assert(R * QPT <= 16, "the two halves of `pend`");
        while (pend & 0xFFFFu) {
            const u32 k = (u32)__ffs((int)(pend & 0xFFFFu)) - 1u;
            const bool simple = ((pend >> (16 + k)) & 1u) != 0;
            pend &= ~(1u << k);
            u32 cq = c[0][0], sq = s[0][0], eq = e[0][0], stv = t[0].st[0];
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int j = 0; j < QPT; ++j) {
                    const bool me = k == (u32)(r * QPT + j);
                    cq = me ? c[r][j] : cq;
                    sq = me ? s[r][j] : sq;
                    eq = me ? e[r][j] : eq;
                    stv = me ? t[r].st[j] : stv;
                }
            const u32 m = (stv >> B0_BITS) & 15u, bq = stv & B0_MASK, beq = L.ctab[cq].w;
            Stab sb{0u, 0u, true, false};
            if (!simple) sb = stab_walk<STRIDE>(recs, bq, beq, sq, eq, [](u32, int) {});
            u32 nt, n_run = 0;
            if (sb.ok) {
                if (!sb.ended) n_run = tail_run<STRIDE>(a, L, recs, cq, bq, beq, eq, sb.w);
                nt = sb.hits + n_run;
            } else {
                nt = walk_tail<FILTER, STRIDE>(recs, bq, beq, sq, eq, min_bp, [](u32, int) {});  // (a long shadow: walked)
            }
            // ids that follow from the position: the write phase needs the run's length, not the block (its ids follow from aux,
            // the mask and the number of records in front) -- no search there, and loads only for the records in front
            const bool keep = IMPL && sb.ok;
            const u32 nst = keep ? run_state(n_run, m) : stv;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                tsum[r] += (k / QPT == (u32)r) ? nt : 0u;
#pragma unroll
                for (int j = 0; j < QPT; ++j) {
                    const bool me = k == (u32)(r * QPT + j);
                    t[r].st[j] = me ? nst : t[r].st[j];
                    t[r].more_bits |= me && keep ? (1u << (RUN_Q_BIT + j)) | (sb.w << (STAB_W_BIT + 2 * j)) | (sb.hits << (STAB_N_BIT + 4 * j)) : 0u;
                }
            }
        }
    }
}

// Emits the hits of the lane's QPT queries (first query q0) in result order: put(position, id) for every hit when
// `want_ids`, o4[j] = position of query j's first hit; `run` = position of the lane's first hit.
// REV: a query's hits leave in DESCENDING stored order (AIList::find, ailist.rs:238-263): the i-th hit of the forward
// scan goes to slot n - 1 - i of the query's n.
// Run form (ids that follow from the position, no min-overlap filter; run_form above): a wide query's ids are id0, id0 + 1,
// ... -- nothing is walked; with `defer` (the wave's ids go straight to memory) a query of >= COOP_MIN ids is not emitted here:
// the caller writes it with the whole wave (coop_runs).
template <int QPT, bool FILTER, bool IMPL, bool REV, int RUNS, bool U64 = false, class Put>
__device__ __forceinline__ void emit_queries(const AccelView &a, const SearchLds &L, const u32 *__restrict__ qc,
                                             const u32 *__restrict__ qs, const u32 *__restrict__ qe, i32 min_bp,
                                             const TileQ<QPT, IMPL> &t, u64 q0, u64 run, bool want_ids, u64 (&o4)[QPT], bool defer,
                                             Put &&put) {
    constexpr u32 STRIDE = IMPL ? 2 : 4;
    const uint4 *__restrict__ recs = IMPL ? a.rec2 : a.rec4;
    const u32 *recw = reinterpret_cast<const u32 *>(a.rec4);
#pragma unroll
    for (int j = 0; j < QPT; ++j) {
        o4[j] = run;
        const u32 b0 = t.st[j] & B0_MASK;
        u32 m = (t.st[j] >> B0_BITS) & (U64 ? 255u : 15u);
        const u32 b_tail = b0 + (U64 ? 2u : 0u);  // walk_tail goes on two blocks behind its argument
        const bool more = (t.more_bits & (1u << j)) != 0;
        if constexpr (IMPL && !FILTER && (RUNS & 1) != 0) {
            if (t.more_bits & (1u << (RUN_Q_BIT + j))) {  // run form (count_rounds): the state word holds the run's length
                // forward order of the ids: the first record's hits (mask m), the hits of the w records tested in front of the run
                // (stab_walk), the run aux + 4 + 4 w, ...  With no record in front and a mask that joins the tail, mask and run are
                // ONE run from aux + (first bit of m).
                const u32 n_run = run_state_n(t.st[j]), w = (t.more_bits >> (STAB_W_BIT + 2 * j)) & 3u,
                          n_stab = (t.more_bits >> (STAB_N_BIT + 4 * j)) & 15u, pm = __popc(m);
                const bool joined = w == 0 && mask_joins_tail(m);
                const u32 n_all = pm + n_stab + n_run;
                if (want_ids) {
                    auto slot = [&](u32 i) -> u64 { return REV ? run + (n_all - 1u - i) : run + i; };
                    u32 i = 0;
                    if (!joined) {
                        u32 mm = m;
                        while (mm) {
                            const int k = __ffs((int)mm) - 1;
                            mm &= mm - 1;
                            put(slot(i++), t.aux[j] + (u32)k);
                        }
                        if (w) {  // the records in front once more (rare: a start in a long interval's shadow)
                            const u32 cq = qc[q0 + j], sq = qs[q0 + j], eq = qe[q0 + j];
                            const u32 b0r = (t.aux[j] - L.idc[cq]) / (u32)ACC_OWN;
                            stab_walk<STRIDE>(recs, b0r, L.ctab[cq].w, sq, eq, [&](u32 x, int k) { put(slot(i++), t.aux[j] + 4u + 4u * x + (u32)k); });
                        }
                    }
                    const u32 n_here = joined ? pm + n_run : n_run;  // the run (with the mask's ids when they join it)
                    if (!((RUNS & 4) && defer && n_here >= COOP_MIN)) {  // (else deferred: coop_runs)
                        const u32 id0 = joined ? t.aux[j] + (u32)(__ffs((int)m) - 1) : t.aux[j] + 4u + 4u * w;
                        for (u32 x = 0; x < n_here; ++x) put(slot(i++), id0 + x);
                    }
                }
                run += n_all;
                continue;
            }
        }
        u32 cq = 0, sq = 0, eq = 0, be = 0;
        u32 n_all = __popc(m);  // the query's hits
        bool tail_counted = false;
        if (more) {
            cq = qc[q0 + j];
            sq = qs[q0 + j];
            eq = qe[q0 + j];
            be = L.ctab[cq].w;
            if (REV || !want_ids) {
                n_all += walk_tail<FILTER, STRIDE>(recs, b_tail, be, sq, eq, min_bp, [](u32, int) {});
                tail_counted = true;
            }
        }
        // slot of the i-th hit of the forward scan
        auto slot = [&](u32 i) -> u64 { return REV ? run + (n_all - 1u - i) : run + i; };
        u32 i = 0;
        if (want_ids) {
            if constexpr (IMPL) {
                while (m) {
                    const int k = __ffs((int)m) - 1;
                    m &= m - 1;
                    put(slot(i++), t.aux[j] + (u32)k);
                }
            } else if (REV) {
                // aux holds the ids of the record's LAST two hits; earlier ones (3rd and 4th from the end) are rare loads
                const u32 pm = __popc(m);
                u32 r = 0;  // rank from the end
                while (m) {
                    const int k = 31 - __clz(m);
                    m &= ~(1u << k);
                    const u32 id = r == 0 ? t.aux[2 * j] : r == 1 ? t.aux[2 * j + 1] : recw[b0 * 16u + 8u + (u32)k];
                    put(slot(pm - 1u - r), id);
                    ++r;
                }
                i = pm;
            } else {
                if (m) put(slot(i++), t.aux[2 * j]);
                m &= m - 1;
                if (m) put(slot(i++), t.aux[2 * j + 1]);
                m &= m - 1;
                while (m) {  // 3rd and 4th hit of a record: rare, a dependent load
                    const int k = __ffs((int)m) - 1;
                    m &= m - 1;
                    put(slot(i++), recw[b0 * 16u + 8u + (u32)k]);
                }
            }
        } else {
            i = __popc(m);
        }
        if (more && want_ids) {
            const u32 n_tail = walk_tail<FILTER, STRIDE>(recs, b_tail, be, sq, eq, min_bp, [&](u32 b, int k) {
                put(slot(i++), IMPL ? t.aux[IMPL ? j : 0] + (u32)ACC_OWN * (b - b0) + (u32)k : recw[b * 16u + 8u + (u32)k]);
            });
            if (!tail_counted) n_all += n_tail;
        }
        run += n_all;
    }
}

// Hit-heavy batches (the wave's ids do not fit its LDS buffer): a wide query in run form has the ids id0, id0 + 1, ..., which the
// whole wave writes instead of one scattered 4-byte store per lane and id (2.5 clk of the CU's vector-memory path each: 1 Mbp
// queries, 33 ids each, ran at 0.07 of the HBM roofline).
// What bounds this path is the INSTRUCTION COUNT, not memory: with the stores left out the first form below took as long.  A
// wave-wide instruction occupies its SIMD for 4 cycles whether it serves 64 queries or one run, so one run's scalar bookkeeping
// (~60 instructions: state by v_readlane, window masks, branches) cost ~47 cycles of the CU per query -- more than the stores.
// Measured forms, 16M queries x 33 ids / 1M x 311 ids (profiles/r04/tok_experiments.txt): run by run, 256-byte-aligned dword
// stores 1366 / 347 us; runs laid end to end in an LDS window, 16-byte flushes 2036 / 321; in a register window 1484 / 560.
// This form: a source lane's FOUR queries at once.  Their runs [S_j, S_j + n_j) come by v_readlane (n_j = 0 for a query the lane
// emitted itself); the lanes then walk the 256-byte-aligned pieces of the lane's region, position x = piece + lane picks its
// run by compares (four sub / compare / add / select groups -- lane-parallel work), and a piece leaves as one store when the
// walk moves past it; a piece that straddles two source lanes is carried over and stored once.
template <int QPT, bool REV>
__device__ __forceinline__ void coop_runs(const TileQ<QPT, true> &t, const u64 (&o4)[QPT], u64 wave_base, u32 *__restrict__ ovals,
                                          u64 cap, int lane) {
    static_assert(QPT == 4, "four runs per source lane");
    // per lane: start, length and id offset of its queries' runs (length 0: not a deferred run)
    u32 S[QPT], N[QPT], K[QPT];
    u32 any_big = 0;
#pragma unroll
    for (int j = 0; j < QPT; ++j) {
        const u32 m = (t.st[j] >> B0_BITS) & 15u, pm = __popc(m), w = (t.more_bits >> (STAB_W_BIT + 2 * j)) & 3u,
                  n_stab = (t.more_bits >> (STAB_N_BIT + 4 * j)) & 15u;
        const bool joined = w == 0 && mask_joins_tail(m);
        const u32 n = run_state_n(t.st[j]) + (joined ? pm : 0u);  // the run (emit_queries: with the mask's ids when they join it)
        const bool big = ((t.more_bits >> (RUN_Q_BIT + j)) & 1u) && n >= COOP_MIN;
        // forward order: [mask ids, ids of the records in front] run; reversed: run [...]
        S[j] = (u32)(o4[j] - wave_base) + (REV || joined ? 0u : pm + n_stab);
        N[j] = big ? n : 0u;
        const u32 id0 = joined ? t.aux[j] + (u32)(__ffs((int)m) - 1) : t.aux[j] + 4u + 4u * w;
        K[j] = REV ? id0 + (n - 1u) + S[j] : id0 - S[j];  // id of output position x: K + x, reversed K - x
        any_big |= N[j];
    }
    unsigned long long todo = __ballot(any_big != 0);
    if (!todo || wave_base >= cap) return;
    u32 *__restrict__ out = ovals + wave_base;                                             // (wave-uniform)
    const u32 room = cap - wave_base > 0xFFFFFFFFull ? 0xFFFFFFFFu : (u32)(cap - wave_base);  // ids the caller's buffer still takes
    // the piece being filled: output positions [w0, w0 + 64); lane l holds the id of position w0 + l when `has`
    u32 w0 = 0, val = 0;
    bool has = false;
    bool open = false;  // (wave-uniform) some lane of the piece is filled
    while (todo) {
        const int src = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        u32 s_[QPT], n_[QPT], k_[QPT];
#pragma unroll
        for (int j = 0; j < QPT; ++j) {
            s_[j] = (u32)__builtin_amdgcn_readlane((int)S[j], src);
            n_[j] = (u32)__builtin_amdgcn_readlane((int)N[j], src);
            k_[j] = (u32)__builtin_amdgcn_readlane((int)K[j], src);
        }
        // the lane's region: from its first run's start to its last run's end (runs ascend with j)
        u32 lo = 0xFFFFFFFFu, hi = 0;
#pragma unroll
        for (int j = 0; j < QPT; ++j)
            if (n_[j]) {
                lo = lo < s_[j] ? lo : s_[j];
                hi = s_[j] + n_[j];
            }
        hi = hi < room ? hi : room;
        for (u32 w = lo & ~63u; w < hi; w += 64) {
            if (w != w0) {  // (w > w0: the regions come in ascending order)
                if (open && has) __builtin_nontemporal_store(val, &out[w0 + (u32)lane]);
                has = false;
                w0 = w;
            }
            open = true;
            const u32 x = w + (u32)lane;
#pragma unroll
            for (int j = 0; j < QPT; ++j) {
                const bool in = x - s_[j] < n_[j];
                val = in ? (REV ? k_[j] - x : k_[j] + x) : val;
                has = has || in;
            }
            has = has && x < room;
        }
    }
    if (open && has) __builtin_nontemporal_store(val, &out[w0 + (u32)lane]);
}

// write phase: CSR offsets and token ids of the lane's QPT queries.  wave_base = global offset
// of the wave's first id.  When the wave's ids fit the wave's LDS buffer (`stage`, stage_cap words) they are
// compacted there and leave as contiguous stores; otherwise (and beyond the caller's capacity) one by one.
template <int QPT, bool FILTER, bool IMPL, bool REV, int RUNS, bool U64 = false>
__device__ __forceinline__ void write_queries(const AccelView &a, const SearchLds &L, const u32 *__restrict__ qc,
                                              const u32 *__restrict__ qs, const u32 *__restrict__ qe, u64 nq, i32 min_bp,
                                              const TileQ<QPT, IMPL> &t, u64 q0, u64 wave_base, u64 *__restrict__ offsets,
                                              u32 *__restrict__ ovals, u64 cap, bool off_vec_ok, u32 *stage, u32 stage_cap,
                                              int lane) {
    const bool staged = cap && t.wtotal <= stage_cap && wave_base + t.wtotal <= cap;
    u64 o4[QPT];
    // ids go either to the wave's LDS buffer (index relative to wave_base) or straight to memory
    emit_queries<QPT, FILTER, IMPL, REV, RUNS, U64>(a, L, qc, qs, qe, min_bp, t, q0, wave_base + t.excl, cap != 0, o4, !staged, [&](u64 pos, u32 id) {
        if (GTARS_TOK_ABLATE & 2) return;
        if (staged)
            stage[(u32)(pos - wave_base)] = id;
        else if (pos < cap)
            ovals[pos] = id;
    });
    if constexpr (IMPL && !FILTER && (RUNS & 4) != 0) {
        if (!staged && cap && a.runs_ok && !(GTARS_TOK_ABLATE & (2 | 128))) coop_runs<QPT, REV>(t, o4, wave_base, ovals, cap, lane);
    }
    if (staged && !(GTARS_TOK_ABLATE & 2)) {
        // the wave's ids, contiguous: 256 bytes per store instruction (LDS operations of a wave execute in order)
        for (u32 i = (u32)lane; i < t.wtotal; i += 64) __builtin_nontemporal_store(stage[i], &ovals[wave_base + i]);
    }
    if (GTARS_TOK_ABLATE & 4) {
        if (o4[0] == 0xFFFFFFFFFFFFFFFFull) offsets[q0] = o4[QPT - 1];
    } else if (QPT >= 2 && off_vec_ok && q0 + QPT <= nq) {
#pragma unroll
        for (int h = 0; h < QPT / 2; ++h) st_stream2(offsets + q0 + 2 * h, o4[2 * h], o4[2 * h + 1]);
    } else {
#pragma unroll
        for (int j = 0; j < QPT; ++j)
            if (q0 + j < nq) offsets[q0 + j] = o4[j];
    }
}

// The write phase in two steps, for the round whose ids can be staged BEFORE the tile's global base is known: the ids
// go to the wave's LDS buffer at wave-relative positions while wave 0 is still busy with the look-back (the other
// waves would only wait for it), and once the base is there the buffer is flushed and the offsets are stored.
// stage_queries returns false when the wave's ids do not fit the buffer (write_queries then serves the round).
template <int QPT, bool FILTER, bool IMPL, bool REV, int RUNS, bool U64 = false>
__device__ __forceinline__ bool stage_queries(const AccelView &a, const SearchLds &L, const u32 *__restrict__ qc,
                                              const u32 *__restrict__ qs, const u32 *__restrict__ qe, i32 min_bp,
                                              const TileQ<QPT, IMPL> &t, u64 q0, u64 cap, u32 *stage, u32 stage_cap,
                                              u32 (&orel)[QPT]) {
    if (cap && t.wtotal > stage_cap) return false;
    // (GTARS_TOK_STAGE_RUNS = 0, measured and rejected: leave a wave-round that holds a query in run form to write_queries, so that
    // the two copies of this early staging carry no run-form code -- 4 KB less code, but the wave-wide test in front of the
    // staging cost the C2 batches 8 %: 1M 16.6 -> 18.0 us, 64M 567 -> 598)
#if !GTARS_TOK_STAGE_RUNS
    if (IMPL && !FILTER && (RUNS & 1) && __ballot((t.more_bits >> RUN_Q_BIT) & 15u)) return false;
#endif
    u64 o4[QPT];
    emit_queries<QPT, FILTER, IMPL, REV, (GTARS_TOK_STAGE_RUNS != 0 ? RUNS : 0), U64>(a, L, qc, qs, qe, min_bp, t, q0, (u64)t.excl, cap != 0, o4, false, [&](u64 pos, u32 id) {
        if (GTARS_TOK_ABLATE & 2) return;
        stage[(u32)pos] = id;
    });
#pragma unroll
    for (int j = 0; j < QPT; ++j) orel[j] = (u32)o4[j];
    return true;
}
template <int QPT>
__device__ __forceinline__ void flush_queries(u64 nq, u32 wtotal, const u32 (&orel)[QPT], u64 q0, u64 wave_base,
                                              u64 *__restrict__ offsets, u32 *__restrict__ ovals, u64 cap, bool off_vec_ok,
                                              const u32 *stage, int lane) {
    if (cap && !(GTARS_TOK_ABLATE & 2)) {
        // within the caller's capacity; 256 bytes per store instruction
        const u32 n = wave_base >= cap ? 0u : (cap - wave_base < (u64)wtotal ? (u32)(cap - wave_base) : wtotal);
        for (u32 i = (u32)lane; i < n; i += 64) __builtin_nontemporal_store(stage[i], &ovals[wave_base + i]);
    }
    if (GTARS_TOK_ABLATE & 4) {
        if (orel[0] == 0xFFFFFFFFu) offsets[q0] = orel[QPT - 1];
    } else if (QPT >= 2 && off_vec_ok && q0 + QPT <= nq) {
#pragma unroll
        for (int h = 0; h < QPT / 2; ++h) st_stream2(offsets + q0 + 2 * h, wave_base + orel[2 * h], wave_base + orel[2 * h + 1]);
    } else {
#pragma unroll
        for (int j = 0; j < QPT; ++j)
            if (q0 + j < nq) offsets[q0 + j] = wave_base + orel[j];
    }
}

// ---------------------------------------------------------------- k_tok_lds
// Persistent workgroups take tiles of TPB * QPT consecutive queries.  Per tile: count phase (search, record
// burst, hit masks), workgroup scan of the hit counts, aggregate published; then -- one tile later, so that the
// predecessors have a tile time to publish theirs -- wave 0 resolves the tile's global base and every wave
// writes its offsets and ids.  The phases are separated by LDS-only barriers (a barrier that also drained the
// vector-memory queue would put the latency of the result stores on the critical path of every tile).
//
// Tile assignment: the grid never exceeds the tile count and every workgroup's FIRST tile is its block index
// (no atomic while the whole grid starts up); later tiles come from a ticket counter, so a tile only ever waits
// on tiles that have been handed out.  Forward progress does not depend on residency or dispatch order either:
// a look-back that finds a predecessor unpublished for `spin_limit` polls COUNTS THAT TILE ITSELF
// (help_count_tile) and publishes the aggregate on the missing workgroup's behalf (granules are idempotent: the
// owner will store the same value).  A launch therefore always completes with the right offsets and ids, also
// when other streams hold CUs.
//
// Measured and rejected on this structure (64M queries, 100k universe; profiles/r02/README.md): per-wave pipelines
// without workgroup barriers (waves poll a control block in LDS: 855 us vs 619); prefetching the next tile's
// queries, wherever the loads are issued -- behind the record burst (772-781 us) or a phase later, after the scan
// barrier, as raw vectors unpacked at the next count (669 vs 620 us for the same build without the prefetch): with
// the HBM stream in flight the tile's own look-back, tail and store traffic queues behind it in the CU's in-order
// vector-memory pipe; a service wave that owns look-back and all stores while 15 waves only load (825 us); a dedicated
// scan wave with prefetching workers (815 us); resolving a tile's base from the workgroup's own previous inclusive
// prefix plus ONE round of 256 granule loads instead of chained 64-granule windows (658 us; 18.6 vs 17.2 us per 1M);
// four wave groups (633 vs 560 us); for 1M-query launches two groups of 2048-query tiles with the second group's search
// held back until the first group's record loads are in flight (20.1 vs 16.8 us: eight waves hide a burst's latency
// worse than sixteen do, and the look-back chain doubles).
// hits of queries [q_begin, q_end), counted by one wave (the look-back's self-service path: rare)
template <bool FILTER>
__device__ __forceinline__ u64 help_count_tile(const AccelView &a, const SearchLds &L, const u32 *qc, const u32 *qs,
                                               const u32 *qe, u64 q_begin, u64 q_end, i32 min_bp) {
    u64 n = 0;
    for (u64 q = q_begin + (threadIdx.x & 63); q < q_end; q += 64) {
        const u32 c = qc[q], s = qs[q], e = qe[q];
        u32 b0, be;
        search_blocks<1>(a, L.lut, L.q, L.ctab, &c, &s, &b0, &be);
        if (b0 < be) {
            const uint4 S = a.rec2[(size_t)b0 * 2], E = a.rec2[(size_t)b0 * 2 + 1];
            n += __popc(block_mask4<FILTER>(S, E, s, e, min_bp));
            if (S.w < e && b0 + 2 < be) n += walk_tail<FILTER, 2>(a.rec2, b0, be, s, e, min_bp, [](u32, int) {});  // (rare path: walked)
        }
    }
    return wave_reduce_sum_u64(n);
}

// resolve_prefix (scan.h) that never gives up: the tile's exclusive prefix by chained look-back; help(t) returns tile t's
// hit count.  Publishes the inclusive prefix.  W windows of 64 predecessor granules are read per round (independent loads:
// one memory round trip) and consumed nearest first; the lanes' contributions are summed once, at the end.
// W = 1 everywhere: wider rounds measured SLOWER both for launches whose groups take many tiles (558 / 600 / 615 / 672 us per
// 64M queries for W = 1 / 2 / 4 / 8) and for one-tile-per-workgroup launches, where all <= 256 tiles publish at about the same
// time (16.5 / 17.3 / 18.4 us per 1M queries for W = 1 / 2 / 4 when every round re-reads all W windows, 16.5 / 16.6 / 17.3 when only
// the incomplete window is polled again): the look-back of a 1M-query launch is the wait for the slowest predecessor
// workgroup, not a chain of round trips (profiles/r03/README.md).
template <int W, class Help>
__device__ __forceinline__ u64 resolve_prefix_helping(u64 *state, u32 tile, u64 agg, int lane, u32 epoch, u32 spin_limit,
                                                      u64 base, Help &&help) {
    const u64 ep = (u64)epoch << EP_SHIFT;
    u64 part = 0;  // this lane's share of the exclusive prefix
    i64 pred = (i64)tile - 1;
    u32 spins = 0;
    bool done = pred < 0;
    auto load_window = [&](i64 first) -> u64 {
        const i64 idx = first - lane;
        u64 v = ST_INC;  // before tile 0: inclusive 0
        if (idx >= 0) {
            v = ld_state(&state[idx]);
            if ((u32)((v >> EP_SHIFT) & EP_MAX) != epoch) v = 0;  // left over from an earlier launch
        }
        return v;
    };
    while (!done) {
#if GTARS_TOK_STAMPS
        if (lane == 0) atomicAdd(&g_tok_stamps[1][spins ? 9 : 8], 1ull);  // rounds: first tries / retries
#endif
        // W windows in one round trip; a window that is not complete yet is polled ON ITS OWN afterwards (the snapshots of the
        // windows behind it stay valid: a granule only ever goes from aggregate to inclusive prefix, both usable)
        u64 v[W];
#pragma unroll
        for (int w = 0; w < W; ++w) v[w] = load_window(pred - 64 * w);
#pragma unroll
        for (int w = 0; w < W; ++w) {
            if (done) continue;
            for (;;) {
                const u64 status = v[w] & ST_MASK;
                const unsigned long long b_inc = __ballot(status == ST_INC);
                const unsigned long long b_inv = __ballot(status == 0);
                const int first_inc = b_inc ? __ffsll((long long)b_inc) - 1 : 64;
                const unsigned long long need = first_inc >= 63 ? ~0ull : ((1ull << (first_inc + 1)) - 1ull);
                if (!(b_inv & need)) {
                    part += lane <= first_inc ? (v[w] & VAL_MASK) : 0ull;
                    if (first_inc < 64) done = true;
                    break;
                }
                // an unpublished predecessor in front of the nearest inclusive prefix: wait for it (or count it ourselves)
                if (++spins > spin_limit) {
                    const i64 missing = pred - 64 * w - (__ffsll((long long)(b_inv & need)) - 1);
                    const u64 h = help((u32)missing) + (missing == 0 ? base : 0ull);
                    if (lane == 0) st_state(&state[missing], (missing == 0 ? ST_INC : ST_AGG) | ep | h);
                    spins = 0;
                } else {
                    __builtin_amdgcn_s_sleep(1);
                }
                v[w] = load_window(pred - 64 * w);
            }
        }
        pred -= 64 * W;
        if (pred < 0) done = true;
    }
    // `base`: what precedes tile 0 (chained launches); later tiles get it through tile 0's inclusive prefix
    const u64 excl = wave_reduce_sum_u48(part) + (tile == 0 ? base : 0ull);
    if (lane == 0) st_state(&state[tile], ST_INC | ep | (excl + agg));
    return excl;
}

// G = 2: the workgroup's 16 waves form two GROUPS of 8 that take tiles independently of each other and
// synchronize through LDS counters instead of s_barrier, so that one group's search, scan, look-back and LDS
// compaction (no vector-memory traffic) run while the other group's record burst and stores keep the CU's
// vector-memory path busy.  Both groups search the same LDS copy of the keys.
__device__ __forceinline__ void group_barrier(u32 *ctr, u32 &phase, u32 members, int lane) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's LDS writes have landed
    phase += members;
    if (lane == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    for (;;) {
        const u32 v = __builtin_amdgcn_readfirstlane(__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
        if ((i32)(v - phase) >= 0) break;
        __builtin_amdgcn_s_sleep(1);
    }
    asm volatile("" ::: "memory");
}

template <int TPB, int QPT, int R, int G, bool FILTER, bool IMPL, bool REV, int RUNS, bool U64 = false>
__global__ void __launch_bounds__(TPB, TPB / 256)
k_tok_lds(AccelView a, const u32 *__restrict__ qc, const u32 *__restrict__ qs, const u32 *__restrict__ qe,
          u64 nq, i32 min_bp, u64 *__restrict__ offsets, u32 *__restrict__ ovals, u64 cap, ScanWs *ws,
          u32 epoch, u32 ticket_base, u32 stage_cap, u32 spin_limit, const u64 *__restrict__ d_base,
          u64 *__restrict__ d_total_out) {
    extern __shared__ __attribute__((aligned(16))) u32 smem[];
    constexpr int NW = TPB / 64, GW = NW / G;  // waves per workgroup / per group
    static_assert(GW * R <= 64, "one lane per wave part in the group's scan");
    __shared__ u32 s_tile[G];
    __shared__ u64 s_prefix[G];
    __shared__ u32 s_scan[G][GW * R];
    __shared__ u32 s_bar[G];
    // a tile is R rounds of GW * 64 * QPT queries: round r of the group's lane t holds queries tile * TILE + r * ROUND + t * QPT ...
    constexpr u32 ROUND = GW * 64 * QPT;
    constexpr u32 TILE = ROUND * R;

    const u32 num_tiles = (u32)((nq + TILE - 1) / TILE);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = wave / GW, gwave = wave % GW;   // group, wave within the group
    const u32 gtid = threadIdx.x - (u32)grp * GW * 64;  // thread within the group
    const u32 first_tiles = gridDim.x * G;            // tiles dealt by position; the others by ticket
    const bool vec_ok = ((((uintptr_t)qc) | ((uintptr_t)qs) | ((uintptr_t)qe)) & 15u) == 0;
    const bool off_vec_ok = (((uintptr_t)offsets) & 15u) == 0;
    const SearchLds L = search_lds_view(a, smem);
    u32 *stage = smem + tok_lds_bytes(a) / 4 + (size_t)wave * stage_cap;
    // Chained launches (the host pipeline feeds a batch in chunks): every offset of this launch starts at *d_base, the
    // running hit count of the chunks before.  It enters through tile 0 -- its inclusive prefix carries the base, and
    // with it every later look-back sum.
    const u64 base = d_base ? *d_base : 0ull;

    u32 c[R][QPT], s[R][QPT], e[R][QPT];
    auto load_tile = [&](u32 t) {
#pragma unroll
        for (int r = 0; r < R; ++r)
            load_queries<QPT>(qc, qs, qe, nq, (u64)t * TILE + (u64)r * ROUND + (u64)gtid * QPT, vec_ok, c[r], s[r], e[r]);
    };
    // the first tile's queries come from HBM: issue their loads before the LDS fill so that both overlap
    u32 tile = blockIdx.x * G + (u32)grp;
    load_tile(tile);
    if (G > 1 && threadIdx.x < G) s_bar[threadIdx.x] = 0;
    if (!(GTARS_TOK_ABLATE & 32)) fill_search_lds<TPB>(a, smem);
    __syncthreads();

    u32 phase = 0;
    auto bar = [&]() {
        if constexpr (G == 1)
            lds_barrier();
        else
            group_barrier(&s_bar[grp], phase, (u32)GW, lane);
    };
    auto help = [&](u32 t) -> u64 {
        const u64 qb = (u64)t * TILE, qn = qb + TILE < nq ? qb + TILE : nq;
        return help_count_tile<FILTER>(a, L, qc, qs, qe, qb, qn, min_bp);
    };

    struct Prev {
        TileQ<QPT, IMPL> q[R];
        u32 wbase[R], total, tile;
    } cur, prev;
    bool have_prev = false, loaded = true;
    const bool draw = num_tiles > first_tiles;  // otherwise one tile per group: nothing to draw
#if GTARS_TOK_STAMPS
    u64 ts_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, ts_last = __builtin_amdgcn_s_memtime();
#endif
    for (;;) {
        const bool has_cur = tile < num_tiles;
        u32 next_tile = num_tiles;
        if (has_cur) {
            // the ticket of the NEXT tile is drawn (one lane) before this tile is counted and handed round through
            // the scan's barrier: its latency is off the critical path and no barrier is spent on it
            u32 ticket = 0;
            if (draw && gtid == 0) ticket = atomicAdd(&ws->ticket, 1u);
            if (!loaded) load_tile(tile);
            loaded = false;
#if GTARS_TOK_STAMPS
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (diagnostic build only: separates the query-load wait)
#endif
            TSTAMP(0);
            u32 tsum[R], inc[R];
#if GTARS_TOK_PRIO & 1
            __builtin_amdgcn_s_setprio(1);  // the group that feeds the vector-memory path goes first
#endif
            count_rounds<R, QPT, FILTER, IMPL, REV, RUNS, U64>(a, L, c, s, e, min_bp, cur.q, tsum);
#if GTARS_TOK_PRIO & 1
            __builtin_amdgcn_s_setprio(0);
#endif
            TSTAMP(1);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                inc[r] = wave_inclusive_scan_u32(tsum[r], lane);
                if (lane == 63) s_scan[grp][r * GW + gwave] = inc[r];
            }
            if (draw && gtid == 0) s_tile[grp] = first_tiles + (ticket - ticket_base);
            bar();
            TSTAMP(2);
            if (draw) next_tile = s_tile[grp];
            // the group's wave parts, scanned by every wave for itself: one LDS read and six DPP adds
            const u32 v = lane < GW * R ? s_scan[grp][lane] : 0u;
            const u32 vinc = wave_inclusive_scan_u32(v, lane);
            cur.total = __builtin_amdgcn_readlane(vinc, 63);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                cur.q[r].wtotal = __builtin_amdgcn_readlane(v, r * GW + gwave);
                cur.wbase[r] = __builtin_amdgcn_readlane(vinc, r * GW + gwave) - cur.q[r].wtotal;
                cur.q[r].excl = inc[r] - tsum[r];
            }
            cur.tile = tile;
            if (gtid == 0) publish_aggregate(ws->state, tile, (u64)cur.total + (tile == 0 ? base : 0ull), epoch);
        }
        // resolve + write the PREVIOUS tile.  Round 0's ids are staged in LDS before the base is known: by waves 1.. while
        // wave 0 looks back, by wave 0 right after.
        u32 orel0[QPT];
        bool pre0 = false;
        const u64 prev_q0 = (u64)prev.tile * TILE + (u64)gtid * QPT;
        if (have_prev && gwave != 0)
            pre0 = stage_queries<QPT, FILTER, IMPL, REV, RUNS, U64>(a, L, qc, qs, qe, min_bp, prev.q[0], prev_q0, cap, stage, stage_cap, orel0);
        if (have_prev && gwave == 0) {
            const u64 excl = (GTARS_TOK_ABLATE & 1) ? (u64)prev.tile * 2400u
                                                    : (draw ? resolve_prefix_helping<1>(ws->state, prev.tile, (u64)prev.total, lane, epoch, spin_limit, base, help)
                                                            : resolve_prefix_helping<GTARS_TOK_LB_W1>(ws->state, prev.tile, (u64)prev.total, lane, epoch,
                                                                                                      spin_limit, base, help));
            if (lane == 0) {
                s_prefix[grp] = excl;
                if (prev.tile == num_tiles - 1) {
                    offsets[nq] = excl + (u64)prev.total;
                    ws->total = excl + (u64)prev.total;
                    if (d_total_out) *d_total_out = excl + (u64)prev.total;
                }
            }
        }
        TSTAMP(7);
        if (have_prev) {
            if (gwave == 0)
                pre0 = stage_queries<QPT, FILTER, IMPL, REV, RUNS, U64>(a, L, qc, qs, qe, min_bp, prev.q[0], prev_q0, cap, stage, stage_cap, orel0);
            TSTAMP(3);
            bar();
            TSTAMP(4);
#if GTARS_TOK_PRIO & 2
            __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const u64 q0 = (u64)prev.tile * TILE + (u64)r * ROUND + (u64)gtid * QPT;
                if (r == 0 && pre0)
                    flush_queries<QPT>(nq, prev.q[0].wtotal, orel0, q0, s_prefix[grp] + prev.wbase[0], offsets, ovals, cap, off_vec_ok,
                                       stage, lane);
                else
                    write_queries<QPT, FILTER, IMPL, REV, RUNS, U64>(a, L, qc, qs, qe, nq, min_bp, prev.q[r], q0, s_prefix[grp] + prev.wbase[r],
                                                          offsets, ovals, cap, off_vec_ok, stage, stage_cap, lane);
            }
        }
#if GTARS_TOK_PRIO & 2
        __builtin_amdgcn_s_setprio(0);
#endif
        TSTAMP(5);
        bar();  // s_tile / s_prefix / s_scan reuse
        TSTAMP(6);
        if (!has_cur) break;
        prev = cur;
        have_prev = true;
        tile = next_tile;
    }
#if GTARS_TOK_STAMPS
    if (lane == 0 && wave < 2) {
        for (int k = 0; k < 11; ++k) atomicAdd(&g_tok_stamps[wave][k], ts_acc[k]);
        atomicAdd(&g_tok_stamps[wave][11], 1ull);
    }
#endif
}

// ---------------------------------------------------------------- k_tok_sweep: batches in (chromosome, start) order
// What Tokenizer.tokenize(path) always delivers: a file-loaded RegionSet is stably sorted by (chr, start)
// (gtars-core/src/models/region_set.rs:182, 502-505), and sorted BED input is the norm.  k_tok_lds serves such a batch like any
// other: 132 KB of search keys copied into every workgroup's LDS, one record request per query.  But 256 CONSECUTIVE sorted queries
// -- a wave's share of a round -- need a CONTIGUOUS slice of the blocked records (13 blocks at config 2's density, less than one in
// a 64M-query batch) and nothing else of the index: no search image, no prologue, and universes of any size (the 16-bit key image
// of k_tok_lds holds ~65k keys: beyond ~130k regions it pays a dependent global read per query).
// Every WAVE works for itself until the tile's scan -- no workgroup barrier in the count phase, so the sixteen waves' dependent
// round trips overlap each other (a first cut with workgroup-wide runs, ranges and staging behind four barriers per tile ran at half
// the speed of k_tok_lds: profiles/r06).  Per wave and round of 256 queries:
//   1. RUNS.  A query opens a run when its chromosome differs from its predecessor's or its start is smaller (DPP; the wave's first
//      query always opens one).  A wave inside one chromosome of a sorted batch has ONE run, a chromosome boundary makes two.
//   2. WINDOW.  64 probes per round trip narrow the chromosome's blocks to a window of <= 64 that holds the first block whose key
//      (the prefix maximum of the ends, blk_first) is > the run's first start: nothing in front of that block can overlap any
//      query of the run -- the argument of k_tok_lds' search (Bits::find starts at lower_bound(start - max_len), bits.rs:141-156:
//      same hits, same order).  One round for a chromosome of <= 4096 blocks.
//   3. STAGE.  From the window's start on, 64 blocks per round trip (own two intervals + key, 20 bytes per block) go to the wave's
//      LDS region until a block starts at or beyond the run's largest end (bits.rs:441-443 ends every scan there): one round trip
//      for a run that spans <= ~60 blocks.
//   4. COUNT.  Per query a binary search of the staged keys and a forward walk of the staged intervals, all in LDS: the first
//      candidate block and a 32-bit hit mask over the 32 intervals from it on travel to the write phase (a query with hits beyond
//      them walks the rest from global memory, like k_tok_lds' walk_tail: rare).
// Then, per tile: scan, publish, chained look-back, id compaction (in the wave's region, dead by then) and stores as in k_tok_lds,
// one tile behind.  Anything else still gets the right answer, slowly: a third run in a wave's round (a shuffled batch), a run
// that outgrows the region -- those queries search and walk in global memory (sweep_query_global).  The caller chooses this
// kernel (GTARS_TOK_SORTED; the host-buffer entry points probe the batch's order on the host first).
constexpr u32 SWP_RUNS = 2;      // runs per wave and round that are staged
constexpr u32 SWP_MASK_IV = 32;  // intervals covered by a query's hit mask, from its first candidate block on

// a query on the global arrays: first candidate block, hit mask over the first SWP_MASK_IV intervals, hit count, "hits beyond"
template <bool FILTER>
__device__ __forceinline__ void sweep_query_global(const AccelView &a, u32 c, u32 s, u32 e, i32 min_bp, u32 &b0, u32 &mask, u32 &n,
                                                   bool &more) {
    const u32 bb = c ? a.chrom_tab[c - 1].w : 0u, be = a.chrom_tab[c].w;
    u32 lo = bb, hi = be;
    while (lo < hi) {
        const u32 mid = lo + ((hi - lo) >> 1);
        if (a.blk_first[mid] <= s)
            lo = mid + 1;
        else
            hi = mid;
    }
    b0 = lo;
    mask = 0;
    n = 0;
    more = false;
    u32 idx = 0;
    for (u32 b = lo; b < be; ++b, idx += 2) {
        const uint4 S = a.rec2[(size_t)b * 2], E = a.rec2[(size_t)b * 2 + 1];
        if (!(S.x < e)) break;  // starts ascend: the scan ends at the first start >= q_end (bits.rs:441-443)
        const bool h0 = hit_test<FILTER>(S.x, E.x, s, e, min_bp);
        const bool go = S.y < e;
        const bool h1 = go && hit_test<FILTER>(S.y, E.y, s, e, min_bp);
        if (idx < SWP_MASK_IV)
            mask |= ((h0 ? 1u : 0u) | (h1 ? 2u : 0u)) << idx;
        else
            more = more || h0 || h1;
        n += (h0 ? 1u : 0u) + (h1 ? 1u : 0u);
        if (!go) break;
    }
}
// the hits of a query from block `from` on (global memory): f(block, slot) in scan order
template <bool FILTER, class F>
__device__ __forceinline__ void sweep_walk_from(const AccelView &a, u32 from, u32 be, u32 s, u32 e, i32 min_bp, F &&f) {
    for (u32 b = from; b < be; ++b) {
        const uint4 S = a.rec2[(size_t)b * 2], E = a.rec2[(size_t)b * 2 + 1];
        if (!(S.x < e)) break;
        if (hit_test<FILTER>(S.x, E.x, s, e, min_bp)) f(b, 0u);
        if (!(S.y < e)) break;
        if (hit_test<FILTER>(S.y, E.y, s, e, min_bp)) f(b, 1u);
    }
}
template <bool FILTER>
__device__ __forceinline__ u64 help_count_tile_sweep(const AccelView &a, const u32 *qc, const u32 *qs, const u32 *qe, u64 q_begin,
                                                     u64 q_end, i32 min_bp) {
    u64 n = 0;
    for (u64 q = q_begin + (threadIdx.x & 63); q < q_end; q += 64) {
        const u32 c = qc[q];
        if (c >= a.n_chrom) continue;
        u32 b0, m, k;
        bool more;
        sweep_query_global<FILTER>(a, c, qs[q], qe[q], min_bp, b0, m, k, more);
        n += k;
    }
    return wave_reduce_sum_u64(n);
}
// largest value of the wave (every lane gets it): the DPP steps of wave_inclusive_scan_u32 with an unsigned maximum
__device__ __forceinline__ u32 wave_max_u32(u32 x) {
    x = max(x, (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false));
    x = max(x, (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false));
    x = max(x, (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false));
    x = max(x, (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false));
    x = max(x, (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false));
    x = max(x, (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false));
    return (u32)__builtin_amdgcn_readlane((int)x, 63);
}

// what a lane keeps about its four queries of a round between the count phase and the write phase
struct SweepQ {
    u32 st[4];      // ids that follow from the position: ACC_OWN * b0 + idc[chrom]; otherwise the first candidate block b0
    u32 mask[4];    // hits among the SWP_MASK_IV intervals from block b0 on
    u32 excl, wtotal;
    u32 more_bits;  // bit j: query j has hits beyond the mask
};

// count phase of one wave's round: runs, window, staging into the wave's region `reg` (cap_w blocks: starts | ends | keys), counts.
template <bool FILTER, bool IMPL>
__device__ __forceinline__ u32 sweep_count_round(const AccelView &a, const u32 (&c)[4], const u32 (&s)[4], const u32 (&e)[4], i32 min_bp,
                                                 u32 *reg, u32 cap_w, u32 max_runs, int lane, SweepQ &t, u64 *ts_acc, u64 &ts_last) {
    constexpr int QPT = 4;
    (void)ts_acc;
    (void)ts_last;
#if GTARS_TOK_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (diagnostic build only: separates the query-load wait)
#endif
    TSTAMP(0);
    // the region: {start, end} of the blocks' own intervals (16 bytes per block: one write per block, one 8-byte read per interval) | keys
    uint2 *r_iv = reinterpret_cast<uint2 *>(reg);
    u32 *r_key = reg + 4u * cap_w;
    // ---- 1. runs
    bool valid[QPT], nr[QPT];
    {
        // the previous lane's last query (wave_shr:1; lane 0 reads `old`: an unknown chromosome, so its first query opens a run)
        const u32 upc = (u32)__builtin_amdgcn_update_dpp((int)GTARS_UNKNOWN_CHROM, (int)c[QPT - 1], 0x138, 0xf, 0xf, false);
        const u32 ups = (u32)__builtin_amdgcn_update_dpp(0, (int)s[QPT - 1], 0x138, 0xf, 0xf, false);
        u32 lc = upc, ls = ups;
#pragma unroll
        for (int j = 0; j < QPT; ++j) {
            valid[j] = c[j] < a.n_chrom;
            nr[j] = valid[j] && (c[j] != lc || s[j] < ls);  // (an invalid predecessor never equals a valid chromosome)
            lc = c[j];
            ls = s[j];
        }
    }
    u32 cnt = 0;
#pragma unroll
    for (int j = 0; j < QPT; ++j) cnt += nr[j] ? 1u : 0u;
    const u32 inc_r = wave_inclusive_scan_u32(cnt, lane);
    const u32 n_runs = (u32)__builtin_amdgcn_readlane((int)inc_r, 63);
    u32 rid[QPT];
    {
        u32 r = inc_r - cnt;
#pragma unroll
        for (int j = 0; j < QPT; ++j) {
            r += nr[j] ? 1u : 0u;
            rid[j] = r - 1u;  // (a valid query always has a run: the wave's first valid query opens one)
        }
    }
    // ---- 2. + 3. per run: window, staging.  Wave-uniform results: first staged block, staged blocks, region offset, id correction
    u32 run_lo[SWP_RUNS], run_len[SWP_RUNS], run_off[SWP_RUNS], run_idc[SWP_RUNS], run_be[SWP_RUNS];
    u32 used = 0;
#pragma unroll
    for (u32 r = 0; r < SWP_RUNS; ++r) {
        run_lo[r] = run_len[r] = run_idc[r] = run_be[r] = 0;
        run_off[r] = 0xFFFFFFFFu;  // not staged
        if (r >= n_runs || r >= max_runs) continue;  // (wave-uniform)
        // the run's first query and largest end
        u32 cc = 0, ss = 0, me = 0;
        bool has = false;
#pragma unroll
        for (int j = QPT - 1; j >= 0; --j) {
            const bool mine = valid[j] && rid[j] == r;
            me = mine ? max(me, e[j]) : me;
            if (nr[j] && rid[j] == r) {
                has = true;
                cc = c[j];
                ss = s[j];
            }
        }
        const unsigned long long hb = __ballot(has);
        const int src = __ffsll((long long)hb) - 1;  // (exactly one lane opens run r)
        const u32 cr = (u32)__builtin_amdgcn_readlane((int)cc, src), s0 = (u32)__builtin_amdgcn_readlane((int)ss, src);
        const u32 max_e = wave_max_u32(me);
        const u32 bb = cr ? a.chrom_tab[cr - 1].w : 0u, be = a.chrom_tab[cr].w;
        run_idc[r] = a.idc[cr];
        run_be[r] = be;
        // window: [lo, hi] holds the first block with key > s0 (hi == be: possibly none)
        u32 lo = bb, hi = be;
        bool hi_holds = false;  // blk_first[hi] > s0 is known (hi < be)
        while (hi - lo > 64u) {
            const u32 n = hi - lo, chunk = (n + 63u) >> 6;
            const u32 last = lo + min(n - 1u, (u32)lane * chunk + chunk - 1u);  // the last block of the lane's chunk
            const bool p = (u32)lane * chunk < n && a.blk_first[last] > s0;
            const unsigned long long b = __ballot(p);
            if (!b) {  // nothing in [lo, hi): the block at hi if it is known to hold, otherwise no key of the chromosome is > s0
                lo = hi = hi_holds ? hi : be;
                break;
            }
            const u32 f = (u32)__ffsll((long long)b) - 1u;
            const u32 nlo = lo + f * chunk;
            hi = lo + min(n - 1u, f * chunk + chunk - 1u);
            hi_holds = true;
            lo = nlo;
        }
        TSTAMP(1);
        // stage from `lo` on, 64 blocks per round trip, until a block starts at or beyond the run's largest end
        run_lo[r] = lo;
        run_off[r] = used;
        u32 len = 0;
        bool fits = true;
        for (u32 b0 = lo; b0 < be;) {
            const u32 nb = min(64u, be - b0);
            if (used + len + nb > cap_w) {
                fits = false;
                break;
            }
            const bool in = (u32)lane < nb;
            uint4 S = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0, 0), E = make_uint4(0, 0, 0, 0);
            u32 key = 0xFFFFFFFFu;
            if (in) {
                const u32 b = b0 + (u32)lane;
                S = a.rec2[(size_t)b * 2];
                E = a.rec2[(size_t)b * 2 + 1];
                key = a.blk_first[b];
                const u32 w = used + len + (u32)lane;
                *reinterpret_cast<uint4 *>(r_iv + 2u * w) = make_uint4(S.x, E.x, S.y, E.y);
                r_key[w] = key;
            }
            len += nb;
            b0 += nb;
            if (__ballot(in && S.x >= max_e)) break;  // (padding blocks start at 0xFFFFFFFF)
        }
        if (fits) {
            run_len[r] = len;
            used += len;
        } else {
            run_off[r] = 0xFFFFFFFFu;  // the run outgrows the region: its queries go to global memory
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#if GTARS_TOK_STAMPS
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
    TSTAMP(2);
    auto run_be_of = [&](int j) -> u32 {
        u32 b = 0;
#pragma unroll
        for (u32 r = 0; r < SWP_RUNS; ++r) b = valid[j] && rid[j] == r ? run_be[r] : b;
        return b;
    };
    // ---- 4. count
    // Everything below is written so that a lane's four queries have their LDS reads in flight TOGETHER and no read sits behind
    // a branch: reads at clamped addresses, selects instead of conditions.  (The first cut's `if (left) { read; compare }` came out
    // as four guarded reads per step, each with its own wait: 28 dependent LDS round trips per search, 6000 cycles per round and
    // wave by the in-kernel stamps -- a quarter of the kernel.)
    u32 tsum = 0;
    t.more_bits = 0;
    u32 off[QPT], len[QPT], pos[QPT], left[QPT], lo_b[QPT], idc[QPT];
    bool lds_q[QPT];
#pragma unroll
    for (int j = 0; j < QPT; ++j) {
        u32 o = 0xFFFFFFFFu, l = 0, lb = 0, ic = 0;
#pragma unroll
        for (u32 r = 0; r < SWP_RUNS; ++r) {
            const bool mine = valid[j] && rid[j] == r;
            o = mine ? run_off[r] : o;
            l = mine ? run_len[r] : l;
            lb = mine ? run_lo[r] : lb;
            ic = mine ? run_idc[r] : ic;
        }
        lds_q[j] = o != 0xFFFFFFFFu;
        off[j] = lds_q[j] ? o : 0u;
        len[j] = lds_q[j] ? l : 0u;
        lo_b[j] = lb;
        idc[j] = ic;
        pos[j] = 0;
        left[j] = len[j];
    }
    // lockstep binary searches of the staged keys: first key > q_start (the trip count is the wave's: the runs' lengths are uniform)
    u32 steps = 0;
#pragma unroll
    for (u32 r = 0; r < SWP_RUNS; ++r) steps = max(steps, 32u - (u32)__clz((int)run_len[r]));
    for (u32 it = 0; it < steps; ++it) {
        u32 key[QPT], half[QPT], mid[QPT];
#pragma unroll
        for (int j = 0; j < QPT; ++j) {
            half[j] = left[j] >> 1;
            mid[j] = pos[j] + half[j];
            key[j] = r_key[off[j] + (mid[j] < len[j] ? mid[j] : 0u)];  // (an address inside the region whatever the lane's state)
        }
#pragma unroll
        for (int j = 0; j < QPT; ++j) {
            const bool below = (left[j] != 0u) & (key[j] <= s[j]);
            pos[j] = below ? mid[j] + 1u : pos[j];
            left[j] = below ? left[j] - half[j] - 1u : half[j];
        }
    }
    TSTAMP(3);
    // lockstep walks of the staged intervals from the block found
    u32 m[QPT], n[QPT], i[QPT], i0[QPT], i1[QPT];
    bool act[QPT], more[QPT], redo[QPT];
    bool any = false;
#pragma unroll
    for (int j = 0; j < QPT; ++j) {
        m[j] = n[j] = 0;
        more[j] = redo[j] = false;
        i0[j] = 2u * (off[j] + pos[j]);
        i1[j] = 2u * (off[j] + len[j]);
        i[j] = i0[j];
        act[j] = lds_q[j] & (i0[j] < i1[j]);
        any = any | act[j];
    }
    while (__ballot(any)) {
        uint2 iv[QPT];
#pragma unroll
        for (int j = 0; j < QPT; ++j) iv[j] = r_iv[i[j] < i1[j] ? i[j] : 2u * off[j]];
        any = false;
#pragma unroll
        for (int j = 0; j < QPT; ++j) {
            const bool go = act[j] & (iv[j].x < e[j]);  // (bits.rs:441-443: the scan ends at the first start >= q_end)
            const bool h = go & hit_test<FILTER>(iv[j].x, iv[j].y, s[j], e[j], min_bp);
            const u32 rel = i[j] - i0[j];
            m[j] |= (h & (rel < SWP_MASK_IV)) ? 1u << (rel & 31u) : 0u;
            more[j] = more[j] | (h & (rel >= SWP_MASK_IV));
            n[j] += h ? 1u : 0u;
            i[j] += go ? 1u : 0u;
            act[j] = go & (i[j] < i1[j]);
            any = any | act[j];
        }
    }
#pragma unroll
    for (int j = 0; j < QPT; ++j) {
        u32 b0 = lo_b[j] + pos[j];
        // the walk ran off the staged blocks before a start >= q_end (cannot happen: staging ends behind the run's largest end or at
        // the chromosome's end -- kept as a guard): the whole query again on the global arrays
        const bool ran_off = lds_q[j] && i[j] == i1[j] && i1[j] > i0[j] && r_iv[i1[j] - 1u].x < e[j] && lo_b[j] + len[j] < run_be_of(j);
        if ((valid[j] && !lds_q[j]) || ran_off) {
            bool mo = false;
            sweep_query_global<FILTER>(a, c[j], s[j], e[j], min_bp, b0, m[j], n[j], mo);
            more[j] = mo;
            idc[j] = a.idc[c[j]];
        }
        t.mask[j] = m[j];
        t.st[j] = IMPL ? (u32)ACC_OWN * b0 + idc[j] : b0;
        t.more_bits |= (more[j] ? 1u : 0u) << j;
        tsum += n[j];
    }
    TSTAMP(4);
    return tsum;
}

// Emits the hits of a lane's four queries of a round in result order (emit_queries' contract): put(position, id) when want_ids,
// o4[j] = position of query j's first hit.
template <bool FILTER, bool IMPL, bool REV, class Put>
__device__ __forceinline__ void emit_sweep(const AccelView &a, const u32 *__restrict__ qc, const u32 *__restrict__ qs,
                                           const u32 *__restrict__ qe, i32 min_bp, const SweepQ &t, u64 q0, u64 run, bool want_ids,
                                           u64 (&o4)[4], Put &&put) {
    const u32 *recw = reinterpret_cast<const u32 *>(a.rec4);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        o4[j] = run;
        u32 m = t.mask[j];
        u32 n_all = __popc(m);
        const bool more = (t.more_bits & (1u << j)) != 0;
        u32 cq = 0, sq = 0, eq = 0, b0 = IMPL ? 0u : t.st[j], be = 0;
        if (more) {  // hits beyond the mask's intervals: counted, then walked (rare)
            cq = qc[q0 + j];
            sq = qs[q0 + j];
            eq = qe[q0 + j];
            be = a.chrom_tab[cq].w;
            if (IMPL) b0 = (t.st[j] - a.idc[cq]) / (u32)ACC_OWN;
            sweep_walk_from<FILTER>(a, b0 + SWP_MASK_IV / 2u, be, sq, eq, min_bp, [&](u32, u32) { ++n_all; });
        }
        if (want_ids && n_all) {
            auto slot = [&](u32 i) -> u64 { return REV ? run + (n_all - 1u - i) : run + i; };
            u32 i = 0;
            while (m) {
                const u32 k = (u32)__ffs((int)m) - 1u;
                m &= m - 1u;
                put(slot(i++), IMPL ? t.st[j] + k : recw[(size_t)(b0 + (k >> 1)) * 16u + 8u + (k & 1u)]);
            }
            if (more)
                sweep_walk_from<FILTER>(a, b0 + SWP_MASK_IV / 2u, be, sq, eq, min_bp, [&](u32 b, u32 k) {
                    put(slot(i++), IMPL ? t.st[j] + (u32)ACC_OWN * (b - b0) + k : recw[(size_t)b * 16u + 8u + k]);
                });
        }
        run += n_all;
    }
}

template <int TPB, int R, bool FILTER, bool IMPL, bool REV>
__global__ void __launch_bounds__(TPB, 4)
k_tok_sweep(AccelView a, const u32 *__restrict__ qc, const u32 *__restrict__ qs, const u32 *__restrict__ qe, u64 nq, i32 min_bp,
            u64 *__restrict__ offsets, u32 *__restrict__ ovals, u64 cap, ScanWs *ws, u32 epoch, u32 ticket_base, u32 region_words,
            u32 spin_limit, const u64 *__restrict__ d_base, u64 *__restrict__ d_total_out, u32 cap_w, u32 max_runs) {
    extern __shared__ __attribute__((aligned(16))) u32 smem[];
    constexpr int QPT = 4, NW = TPB / 64;
    constexpr u32 ROUND = TPB * QPT, TILE = ROUND * R;
    static_assert(NW * R <= 64, "one lane per wave part in the tile's scan");
    __shared__ u32 s_tile, s_scan[NW * R];
    __shared__ u64 s_prefix;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    u32 *reg = smem + (size_t)wave * region_words;  // the wave's region: staged blocks in the count phase, id staging in the write phase
    const u32 num_tiles = (u32)((nq + TILE - 1) / TILE);
    const bool vec_ok = ((((uintptr_t)qc) | ((uintptr_t)qs) | ((uintptr_t)qe)) & 15u) == 0;
    const bool off_vec_ok = (((uintptr_t)offsets) & 15u) == 0;
    const u64 base = d_base ? *d_base : 0ull;
    const u32 first_tiles = gridDim.x;
    const bool draw = num_tiles > first_tiles;
    auto help = [&](u32 t) -> u64 {
        const u64 qb = (u64)t * TILE, qn = qb + TILE < nq ? qb + TILE : nq;
        return help_count_tile_sweep<FILTER>(a, qc, qs, qe, qb, qn, min_bp);
    };
    struct Prev {
        SweepQ q[R];
        u32 wbase[R], total, tile;
    } cur, prev;
    bool have_prev = false;
    u32 tile = blockIdx.x;
    u64 ts_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, ts_last = 0;
#if GTARS_TOK_STAMPS
    ts_last = __builtin_amdgcn_s_memtime();
#endif
    for (;;) {
        const bool has_cur = tile < num_tiles;
        u32 next_tile = num_tiles;
        if (has_cur) {
            u32 ticket = 0;
            if (draw && threadIdx.x == 0) ticket = atomicAdd(&ws->ticket, 1u);
            u32 tsum[R], inc[R];
            // a wave's rounds one after the other, the next round's queries in flight while the current one is served
            u32 c[QPT], s[QPT], e[QPT];
            load_queries<QPT>(qc, qs, qe, nq, (u64)tile * TILE + (u64)threadIdx.x * QPT, vec_ok, c, s, e);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                u32 c2[QPT], s2[QPT], e2[QPT];
                if (r + 1 < R) load_queries<QPT>(qc, qs, qe, nq, (u64)tile * TILE + (u64)(r + 1) * ROUND + (u64)threadIdx.x * QPT, vec_ok, c2, s2, e2);
                tsum[r] = sweep_count_round<FILTER, IMPL>(a, c, s, e, min_bp, reg, cap_w, max_runs, lane, cur.q[r], ts_acc, ts_last);
                if (r + 1 < R) {
#pragma unroll
                    for (int j = 0; j < QPT; ++j) {
                        c[j] = c2[j];
                        s[j] = s2[j];
                        e[j] = e2[j];
                    }
                }
                inc[r] = wave_inclusive_scan_u32(tsum[r], lane);
                if (lane == 63) s_scan[r * NW + wave] = inc[r];
            }
            if (draw && threadIdx.x == 0) s_tile = first_tiles + (ticket - ticket_base);
            lds_barrier();
            TSTAMP(5);
            if (draw) next_tile = s_tile;
            const u32 v = lane < NW * R ? s_scan[lane] : 0u;
            const u32 vinc = wave_inclusive_scan_u32(v, lane);
            cur.total = __builtin_amdgcn_readlane(vinc, 63);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                cur.q[r].wtotal = __builtin_amdgcn_readlane(v, r * NW + wave);
                cur.wbase[r] = __builtin_amdgcn_readlane(vinc, r * NW + wave) - cur.q[r].wtotal;
                cur.q[r].excl = inc[r] - tsum[r];
            }
            cur.tile = tile;
            if (threadIdx.x == 0) publish_aggregate(ws->state, tile, (u64)cur.total + (tile == 0 ? base : 0ull), epoch);
        }
        // ---- resolve + write the PREVIOUS tile (round 0's ids are staged before the base is known: by waves 1.. while wave 0 looks
        // back, by wave 0 right after)
        u32 orel[QPT];
        auto stage_round = [&](int r) -> bool {
            if (cap && prev.q[r].wtotal > region_words) return false;
            u64 o4[QPT];
            emit_sweep<FILTER, IMPL, REV>(a, qc, qs, qe, min_bp, prev.q[r], (u64)prev.tile * TILE + (u64)r * ROUND + (u64)threadIdx.x * QPT,
                                          (u64)prev.q[r].excl, cap != 0, o4, [&](u64 pos, u32 id) { reg[(u32)pos] = id; });
#pragma unroll
            for (int j = 0; j < QPT; ++j) orel[j] = (u32)o4[j];
            return true;
        };
        bool pre0 = false;
        if (have_prev && wave != 0) pre0 = stage_round(0);
        if (have_prev && wave == 0) {
            const u64 excl = resolve_prefix_helping<1>(ws->state, prev.tile, (u64)prev.total, lane, epoch, spin_limit, base, help);
            if (lane == 0) {
                s_prefix = excl;
                if (prev.tile == num_tiles - 1) {
                    offsets[nq] = excl + (u64)prev.total;
                    ws->total = excl + (u64)prev.total;
                    if (d_total_out) *d_total_out = excl + (u64)prev.total;
                }
            }
            pre0 = stage_round(0);
        }
        TSTAMP(6);
        if (have_prev) {
            lds_barrier();  // s_prefix
            TSTAMP(7);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const u64 q0 = (u64)prev.tile * TILE + (u64)r * ROUND + (u64)threadIdx.x * QPT;
                const u64 wave_base = s_prefix + prev.wbase[r];
                const bool staged = r == 0 ? pre0 : stage_round(r);
                if (staged) {
                    flush_queries<QPT>(nq, prev.q[r].wtotal, orel, q0, wave_base, offsets, ovals, cap, off_vec_ok, reg, lane);
                } else {
                    u64 o4[QPT];
                    emit_sweep<FILTER, IMPL, REV>(a, qc, qs, qe, min_bp, prev.q[r], q0, wave_base + prev.q[r].excl, cap != 0, o4,
                                                  [&](u64 pos, u32 id) {
                                                      if (pos < cap) ovals[pos] = id;
                                                  });
                    if (off_vec_ok && q0 + QPT <= nq) {
#pragma unroll
                        for (int h = 0; h < QPT / 2; ++h) st_stream2(offsets + q0 + 2 * h, o4[2 * h], o4[2 * h + 1]);
                    } else {
#pragma unroll
                        for (int j = 0; j < QPT; ++j)
                            if (q0 + j < nq) offsets[q0 + j] = o4[j];
                    }
                }
                // (the region is reused by the next round's ids / the next tile's blocks: LDS operations of a wave execute in order)
            }
        }
        TSTAMP(8);
        lds_barrier();  // s_tile / s_prefix / s_scan reuse
        TSTAMP(9);
        if (!has_cur) break;
        prev = cur;
        have_prev = true;
        tile = next_tile;
    }
#if GTARS_TOK_STAMPS
    if (lane == 0 && wave < 2) {
        for (int k = 0; k < 11; ++k) atomicAdd(&g_tok_stamps[wave][k], ts_acc[k]);
        atomicAdd(&g_tok_stamps[wave][11], 1ull);
    }
#endif
}

// branch-free form of load_queries for 16-byte-aligned arrays: lanes past the end load element 0 and are
// masked afterwards; a lane's 16 bytes never leave the array's last 16-byte chunk
__device__ __forceinline__ void load_queries_bf4(const u32 *__restrict__ qc, const u32 *__restrict__ qs, const u32 *__restrict__ qe,
                                                 u64 nq, u64 q0, u32 (&c)[4], u32 (&s)[4], u32 (&e)[4]) {
    const u64 qa = q0 < nq ? q0 : 0;
    const u32x4 c4 = ld_stream4(qc + qa), s4 = ld_stream4(qs + qa), e4 = ld_stream4(qe + qa);
    c[0] = q0 + 0 < nq ? c4.x : GTARS_UNKNOWN_CHROM; c[1] = q0 + 1 < nq ? c4.y : GTARS_UNKNOWN_CHROM;
    c[2] = q0 + 2 < nq ? c4.z : GTARS_UNKNOWN_CHROM; c[3] = q0 + 3 < nq ? c4.w : GTARS_UNKNOWN_CHROM;
    s[0] = s4.x; s[1] = s4.y; s[2] = s4.z; s[3] = s4.w;
    e[0] = e4.x; e[1] = e4.y; e[2] = e4.z; e[3] = e4.w;
}


// ---------------------------------------------------------------- K2 on the same structure
// count_overlaps / any_overlaps (multi_chrom_overlapper.rs:483-517) for a Bits-kind index: the search and
// the record burst without the scan -- counts do not depend on the result order, nor on the ids (always the
// 32-byte records).
// PF (16-byte-aligned query arrays): the next tile's queries are loaded right behind the record burst (branch-free,
// so that the wait for the records is a counted one that leaves them in flight): 570 -> 532 us per 64M queries.
// (The same prefetch makes the tokenizer SLOWER -- 632 -> 772 us per 64M queries -- and is not used there.)
// MARK (index-side subset, multi_chrom_overlapper.rs:454-478 / indexed_region_set.rs:201-230): instead of counting, every hit
// sets the bit of its STORED POSITION in `mark` (launched on the position view of the structure, AccelView::idc = idc_pos:
// position of (block b, slot k) = ACC_OWN * b + k + idc[chrom]).  A bit already seen set is not set again (a stale read only
// costs a redundant atomic), so a dense batch does not hammer the same words.
template <int TPB, bool FILTER, bool PF, bool MARK = false>
__global__ void __launch_bounds__(TPB, 4)
k_count_lds(AccelView a, const u32 *__restrict__ qc, const u32 *__restrict__ qs, const u32 *__restrict__ qe, u64 nq,
            i32 min_bp, u32 *__restrict__ counts, u8 *__restrict__ any, u32 *mark = nullptr) {
    extern __shared__ __attribute__((aligned(16))) u32 smem[];
    constexpr int QPT = 4;
    constexpr u64 TILE = (u64)TPB * QPT;
    const SearchLds L = search_lds_view(a, smem);
    const u64 num_tiles = (nq + TILE - 1) / TILE;
    u32 c[QPT], s[QPT], e[QPT];
    if constexpr (PF) load_queries_bf4(qc, qs, qe, nq, (u64)blockIdx.x * TILE + (u64)threadIdx.x * QPT, c, s, e);
    fill_search_lds<TPB>(a, smem);
    __syncthreads();
    for (u64 tile = blockIdx.x; tile < num_tiles; tile += gridDim.x) {
        const u64 q0 = tile * TILE + (u64)threadIdx.x * QPT;
        if constexpr (!PF) load_queries<QPT>(qc, qs, qe, nq, q0, false, c, s, e);
        u32 b0[QPT], be[QPT];
        uint4 S[QPT], E[QPT];
        bool act[QPT];
        // two units of two queries: the second unit's LDS search runs while the first unit's record loads are in flight
#pragma unroll
        for (int h0 = 0; h0 < QPT; h0 += 2) {
            search_blocks<2>(a, L.lut, L.q, L.ctab, c + h0, s + h0, b0 + h0, be + h0);
#pragma unroll
            for (int j = h0; j < h0 + 2; ++j) {
                act[j] = b0[j] < be[j];
                const uint4 *rec = a.rec2 + (size_t)(act[j] ? b0[j] : 0u) * 2;
                S[j] = rec[0];
                E[j] = rec[1];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        u32 c2[QPT], s2[QPT], e2[QPT];
        if constexpr (PF) {
            __builtin_amdgcn_sched_barrier(0);
            load_queries_bf4(qc, qs, qe, nq, (tile + gridDim.x) * TILE + (u64)threadIdx.x * QPT, c2, s2, e2);
            __builtin_amdgcn_sched_barrier(0);
        }
        u32 n[QPT];
        if constexpr (MARK) {
#pragma unroll
            for (int j = 0; j < QPT; ++j) {
                if (!act[j]) continue;  // also every query past the end of the batch (unknown chromosome)
                const u32 pbase = L.idc[c[j]];
                auto set_bit = [&](u32 b, int k) {
                    const u32 pos = (u32)ACC_OWN * b + (u32)k + pbase, w = pos >> 5, bit = 1u << (pos & 31u);
                    if (!(mark[w] & bit)) atomicOr(&mark[w], bit);
                };
                u32 m = block_mask4<FILTER>(S[j], E[j], s[j], e[j], min_bp);
                while (m) {
                    const int k = __ffs((int)m) - 1;
                    m &= m - 1;
                    set_bit(b0[j], k);
                }
                if (S[j].w < e[j] && b0[j] + 2 < be[j]) walk_tail<FILTER, 2>(a.rec2, b0[j], be[j], s[j], e[j], min_bp, set_bit);
            }
        } else {
#pragma unroll
            for (int j = 0; j < QPT; ++j) {
                const u32 m = act[j] ? block_mask4<FILTER>(S[j], E[j], s[j], e[j], min_bp) : 0u;
                n[j] = __popc(m);
                if (act[j] && S[j].w < e[j] && b0[j] + 2 < be[j])
                    n[j] += (!FILTER && run_form(a, m, S[j].w, s[j])) ? tail_run<2>(a, L, a.rec2, c[j], b0[j], be[j], e[j])
                                                          : walk_tail<FILTER, 2>(a.rec2, b0[j], be[j], s[j], e[j], min_bp, [](u32, int) {});
            }
#pragma unroll
            for (int j = 0; j < QPT; ++j) {
                if (q0 + j < nq) {
                    if (counts) counts[q0 + j] = n[j];
                    if (any) any[q0 + j] = n[j] ? 1 : 0;
                }
            }
        }
        if constexpr (PF) {
#pragma unroll
            for (int j = 0; j < QPT; ++j) {
                c[j] = c2[j];
                s[j] = s2[j];
                e[j] = e2[j];
            }
        }
    }
}

// ---------------------------------------------------------------- launchers

static int env_int(const char *name, int dflt) {
    const char *v = cfg_get(name);
    return v && *v ? atoi(v) : dflt;
}

constexpr size_t TOK_LDS_MAX = 150 * 1024;    // search structure
constexpr size_t TOK_LDS_TOTAL = 158 * 1024;  // search structure + id staging (static LDS comes on top)
bool tokenize_lds_supported(const AccelView &a) {
    return a.n_blocks > 0 && a.n_blocks <= ((1u << 22) - 1u) && a.n_units > 0 && tok_lds_bytes(a) <= TOK_LDS_MAX &&
           (u64)a.max_chrom_n * 2048ull <= 0xFFFFFFFFull;  // the smallest tile's 32-bit hit total
}

// The dynamic-LDS attribute belongs to the function (per device), not to the calling thread: it is raised once
// to the largest size any launch uses and never lowered again.  Also caches the CU count.
struct KernelSetup {
    std::mutex mu;
    bool done[64] = {};
    int cus[64] = {};
    gtars_status get(const void *fn, int &dev, int &n_cus) {
        GT_HIP(hipGetDevice(&dev));
        if (dev < 0 || dev >= 64) return fail(GTARS_ERR_INTERNAL, "device ordinal out of range");
        std::lock_guard<std::mutex> g(mu);
        if (!done[dev]) {
            GT_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)TOK_LDS_TOTAL));
            int c = 256;
            GT_HIP(hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev));
            cus[dev] = c;
            done[dev] = true;
        }
        n_cus = cus[dev];
        return GTARS_OK;
    }
};

gtars_status launch_count_lds(const AccelView &a, const u32 *qc, const u32 *qs, const u32 *qe, u64 nq, int has_min,
                              i32 min_overlap, u32 *counts, u8 *any, hipStream_t st) {
    if (nq == 0) return GTARS_OK;
    constexpr int TPB = 1024;
    const size_t lds = tok_lds_bytes(a);
    const bool filter = has_min && min_overlap > 1;
    const i32 min_bp = has_min ? min_overlap : 0;
    const bool pf = ((((uintptr_t)qc) | ((uintptr_t)qs) | ((uintptr_t)qe)) & 15u) == 0;
    auto kern = filter ? (pf ? k_count_lds<TPB, true, true> : k_count_lds<TPB, true, false>)
                       : (pf ? k_count_lds<TPB, false, true> : k_count_lds<TPB, false, false>);
    static KernelSetup setup[4];
    int dev = 0, cus = 256;
    gtars_status s0 = setup[(filter ? 2 : 0) + (pf ? 1 : 0)].get(reinterpret_cast<const void *>(kern), dev, cus);
    if (s0) return s0;
    const u64 tiles = (nq + (u64)TPB * 4 - 1) / ((u64)TPB * 4);
    const unsigned grid = (unsigned)std::min<u64>(tiles, (u64)cus);
    ProfScope p("k_count_lds", st);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(TPB), lds, st, a, qc, qs, qe, nq, min_bp, counts, any, (u32 *)nullptr);
    GT_HIP(hipGetLastError());
    return GTARS_OK;
}

// index-side subset: bit p of `mark` (zeroed by the caller, ceil(n / 32) words) = stored position p is hit by some query.
// `a` must be the POSITION view of the structure (gtars_index::accel_pos()).
gtars_status launch_mark_lds(const AccelView &a, const u32 *qc, const u32 *qs, const u32 *qe, u64 nq, int has_min, i32 min_overlap,
                             u32 *mark, hipStream_t st) {
    if (nq == 0) return GTARS_OK;
    constexpr int TPB = 1024;
    const size_t lds = tok_lds_bytes(a);
    const bool filter = has_min && min_overlap > 1;
    const i32 min_bp = has_min ? min_overlap : 0;
    auto kern = filter ? k_count_lds<TPB, true, false, true> : k_count_lds<TPB, false, false, true>;
    static KernelSetup setup[2];
    int dev = 0, cus = 256;
    gtars_status s0 = setup[filter ? 1 : 0].get(reinterpret_cast<const void *>(kern), dev, cus);
    if (s0) return s0;
    const u64 tiles = (nq + (u64)TPB * 4 - 1) / ((u64)TPB * 4);
    const unsigned grid = (unsigned)std::min<u64>(tiles, (u64)cus);
    ProfScope p("k_mark_lds", st);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(TPB), lds, st, a, qc, qs, qe, nq, min_bp, (u32 *)nullptr, (u8 *)nullptr, mark);
    GT_HIP(hipGetLastError());
    return GTARS_OK;
}

static u64 tok_tile_queries(int tpb, int qpt) { return (u64)tpb * (u64)qpt; }

// Queries per lane and round: 4 (one burst of eight 16-byte loads per lane; 2 was slower at every batch size).
// Rounds per tile: 2 once every CU has an 8192-query tile -- the query loads of both rounds are in flight together and
// the per-tile costs (barriers, ticket, look-back) are paid half as often (2M queries: 26.6 -> 25.5 us, 4M: 46.1 -> 43.3);
// smaller batches keep 4096-query tiles so that every CU gets one.
static int choose_rounds(u64 nq, int cus) {
    const int forced = env_int("GTARS_TOK_ROUNDS", 0);
    if (forced == 1 || forced == 2) return forced;
    return nq >= (u64)cus * 8192ull ? 2 : 1;
}
// wave groups per workgroup (k_tok_lds): 2 once every group has several tiles
static int choose_groups(u64 nq, int cus) {
    const int forced = env_int("GTARS_TOK_GROUPS", 0);
    if (forced == 1 || forced == 2) return forced;
    return nq >= (u64)cus * 8192ull * 4ull ? 2 : 1;
}

size_t tokenize_lds_ws_bytes(u64 nq) {
    // sized for the smallest tile (k_tok_sweep: 256 threads x 4 queries)
    return scan_ws_bytes_for_tiles((nq + 1023) / 1024);
}

// words of id staging per wave: what is left of the LDS budget, at most 512 (a wave-tile of 256 queries rarely has more hits)
static u32 stage_words(const AccelView &a, int tpb, int per_cu) {
    const int forced = env_int("GTARS_TOK_STAGE", -1);
    const size_t search = tok_lds_bytes(a);
    const size_t budget = TOK_LDS_TOTAL / (size_t)per_cu;
    if (search >= budget) return 0;
    size_t w = (budget - search) / 4 / (size_t)(tpb / 64);
    w = std::min<size_t>(w, forced >= 0 ? (size_t)forced : 512) / 64 * 64;
    return w >= 128 ? (u32)w : 0u;
}

template <int TPB, int QPT, int R, int G, bool FILTER, bool IMPL, bool REV, int RUNS, bool U64 = false>
static gtars_status launch_tok_t(const AccelView &a, const u32 *qc, const u32 *qs, const u32 *qe, u64 nq,
                                 i32 min_bp, const EnumOut &out, ScanWs *ws, ScanEpoch &ep, const u64 *d_base, u64 *d_total_out,
                                 hipStream_t st) {
    static KernelSetup setup;
    int dev = 0, cus = 256;
    gtars_status s0 = setup.get(reinterpret_cast<const void *>(k_tok_lds<TPB, QPT, R, G, FILTER, IMPL, REV, RUNS, U64>), dev, cus);
    if (s0) return s0;
    // one 1024-thread workgroup per CU: one LDS copy of the search keys, 16 waves -- what 128 VGPRs admit
    const u32 stage = stage_words(a, TPB, 1);
    const size_t lds = tok_lds_bytes(a) + (size_t)stage * 4 * (TPB / 64);
    const u32 spin_limit = (u32)env_int("GTARS_TOK_SPIN_LIMIT", 4096);
    const u64 tile_q = tok_tile_queries(TPB / G, QPT * R);
    const u64 tiles = (nq + tile_q - 1) / tile_q;
    const u64 grid = std::min<u64>((u64)cus, (tiles + G - 1) / G);
    const u64 cap = out.vals ? out.capacity : 0;
    hipLaunchKernelGGL((k_tok_lds<TPB, QPT, R, G, FILTER, IMPL, REV, RUNS, U64>), dim3((unsigned)grid), dim3(TPB), lds, st, a, qc, qs, qe, nq,
                       min_bp, out.offsets, out.vals, cap, ws, ep.epoch, ep.ticket_base, stage, spin_limit, d_base, d_total_out);
    GT_HIP(hipGetLastError());
    // tickets drawn by this launch: one per tile beyond the first `grid * G`, plus one failing draw per group
    if (tiles > grid * G) ep.ticket_base += (u32)tiles;
    return GTARS_OK;
}

gtars_status launch_tokenize_lds(const AccelView &a, const u32 *qc, const u32 *qs, const u32 *qe, u64 nq,
                                 int has_min, i32 min_overlap, const EnumOut &out, void *scan_ws,
                                 size_t scan_ws_bytes, ScanEpoch &ep, hipStream_t st, const u64 *d_base, u64 *d_total_out, bool reverse) {
    if (out.starts || out.ends) return fail(GTARS_ERR_INTERNAL, "the LDS tokenizer writes vals only");
    if (nq == 0) {
        if (d_base) {
            GT_HIP(hipMemcpyAsync(out.offsets, d_base, sizeof(u64), hipMemcpyDeviceToDevice, st));
            GT_HIP(hipMemcpyAsync(&((ScanHead *)scan_ws)->total, d_base, sizeof(u64), hipMemcpyDeviceToDevice, st));
            if (d_total_out) GT_HIP(hipMemcpyAsync(d_total_out, d_base, sizeof(u64), hipMemcpyDeviceToDevice, st));
        } else {
            GT_HIP(hipMemsetAsync(out.offsets, 0, sizeof(u64), st));
            GT_HIP(hipMemsetAsync(&((ScanHead *)scan_ws)->total, 0, sizeof(u64), st));
            if (d_total_out) GT_HIP(hipMemsetAsync(d_total_out, 0, sizeof(u64), st));
        }
        return GTARS_OK;
    }
    constexpr int TPB = 1024;
    int cus = 256;
    {
        int dev = 0;
        GT_HIP(hipGetDevice(&dev));
        static int cached[64] = {};
        if (dev >= 0 && dev < 64) {
            if (!cached[dev]) {
                int c = 256;
                GT_HIP(hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev));
                cached[dev] = c;
            }
            cus = cached[dev];
        }
    }
    const bool impl = a.ids_affine != 0;
    const bool filter = has_min && min_overlap > 1;
    // Two builds of every kernel without a min-overlap filter: with the run form of wide queries (tail_run, coop_runs) and without.
    // The run form costs a C2 launch 2-3.5 % (registers of the two-round kernels, code in the tile loop), and a launch whose id
    // buffer cannot hold WIDE_IDS_PER_QUERY ids per query cannot be a hit-heavy batch that completes -- it would end in
    // GTARS_ERR_CAPACITY -- so it runs the build without; an offsets-only launch (capacity 0: the sizing pass of a two-pass caller)
    // keeps the run form, which is what makes a hit-heavy batch's count cheap.  Both builds give the same offsets and ids.
    const u64 cap_ids = out.vals ? out.capacity : 0;
    // The caller may say which it is (EnumOut::hint <- gtars_tokenize_device_ex / gtars_fill_device_n: a two-pass caller knows
    // its hits per query from the sizing pass); the capacity rule is only the default.
    const bool by_rule = cap_ids == 0 || cap_ids / WIDE_IDS_PER_QUERY >= nq;
    const bool want_wide = out.hint == GTARS_TOK_WIDE || (out.hint != GTARS_TOK_NARROW && by_rule);
    const bool wide = GTARS_TOK_RUNS != 0 && !filter && a.runs_ok && !cfg_flag("GTARS_TOK_NARROW") &&
                      (want_wide || cfg_flag("GTARS_TOK_WIDE"));  // (switches: tests, A/B)
    prof_note_fact(wide ? "tok_build_wide" : "tok_build_narrow");
    // two rounds keep 8 queries' state per lane across the scan: with the id pairs of explicit-id records that spills; the kernels
    // with the run form gain nothing from the second round (16M queries of 33 ids: 1125 us with one round, 1139 with two)
    int rounds = impl && !wide ? choose_rounds(nq, cus) : 1;
    int groups = choose_groups(nq, cus);
    // unit records (AccelView::rec8: universes of ~130k-260k regions, two blocks per LDS key): one 64-byte request per query instead
    // of the block key and then a 32-byte record; the narrow build, one round (eight more registers of record per pipelined query)
    const bool unit_rec = a.rec8 && a.top_shift == 1 && impl && !filter && !reverse && !wide;
    if (unit_rec) rounds = 1;
    prof_note_fact(unit_rec ? "tok_unit_records" : "tok_block_records");
    // a tile's hits are summed in 32 bits: tile queries x (most intervals on one chromosome) must fit
    const u64 dense = std::max<u64>(a.max_chrom_n, 1);
    if (dense * tok_tile_queries(TPB / groups, 4 * rounds) > 0xFFFFFFFFull) rounds = 1;
    if (dense * tok_tile_queries(TPB / groups, 4 * rounds) > 0xFFFFFFFFull) groups = 2;
    if (dense * tok_tile_queries(TPB / groups, 4 * rounds) > 0xFFFFFFFFull)
        return fail(GTARS_ERR_INVALID_ARG, "index too dense: a tile of queries could have more than 2^32 - 1 hits");
    const u64 tile_q = tok_tile_queries(TPB / groups, 4 * rounds);
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
    const i32 min_bp = has_min ? min_overlap : 0;
    ScanWs *ws = (ScanWs *)scan_ws;
    if (!impl && !a.rec4) return fail(GTARS_ERR_INTERNAL, "index has no id records");
    ProfScope p("k_tok_lds", st);
#define GT_TOK_CASE(N, GG, F, I, V, W)                                                                                      \
    if (rounds == N && groups == GG && filter == F && impl == I && reverse == V && wide == (W != 0))                        \
        return launch_tok_t<TPB, 4, N, GG, F, I, V, W>(a, qc, qs, qe, nq, min_bp, out, ws, ep, d_base, d_total_out, st);
#define GT_TOK_GEOM(N, GG, I)                                                                                               \
    GT_TOK_CASE(N, GG, false, I, false, 0)                                                                                  \
    GT_TOK_CASE(N, GG, true, I, false, 0)                                                                                   \
    GT_TOK_CASE(N, GG, false, I, true, 0)                                                                                   \
    GT_TOK_CASE(N, GG, true, I, true, 0)
#define GT_TOK_WIDE(GG, I)                                                                                                  \
    GT_TOK_CASE(1, GG, false, I, false, GTARS_TOK_RUNS)                                                                     \
    GT_TOK_CASE(1, GG, false, I, true, GTARS_TOK_RUNS)
    if (unit_rec && groups == 1) return launch_tok_t<TPB, 4, 1, 1, false, true, false, 0, true>(a, qc, qs, qe, nq, min_bp, out, ws, ep, d_base, d_total_out, st);
    if (unit_rec && groups == 2) return launch_tok_t<TPB, 4, 1, 2, false, true, false, 0, true>(a, qc, qs, qe, nq, min_bp, out, ws, ep, d_base, d_total_out, st);
    GT_TOK_GEOM(1, 1, true)
    GT_TOK_GEOM(2, 1, true)
    GT_TOK_GEOM(1, 2, true)
    GT_TOK_GEOM(2, 2, true)
    GT_TOK_GEOM(1, 1, false)
    GT_TOK_GEOM(1, 2, false)
#if GTARS_TOK_RUNS
    GT_TOK_WIDE(1, true)
    GT_TOK_WIDE(2, true)
    GT_TOK_WIDE(1, false)
    GT_TOK_WIDE(2, false)
#endif
#undef GT_TOK_WIDE
#undef GT_TOK_GEOM
#undef GT_TOK_CASE
    return fail(GTARS_ERR_INTERNAL, "unsupported tokenizer launch geometry");
}

// ---- the sweep form (k_tok_sweep): batches in (chromosome, start) order
bool tokenize_sweep_supported(const AccelView &a) {
    return a.n_blocks > 0 && a.rec2 && a.blk_first && a.chrom_tab && a.idc && (a.ids_affine || a.rec4) &&
           (u64)a.max_chrom_n * 4096ull <= 0xFFFFFFFFull;  // a tile's 32-bit hit total
}

// Geometry: workgroups of 256 threads, FOUR per CU (16 waves per CU either way: the kernel sits at ~115 VGPRs).  A tile's phases are a
// chain of dependent round trips -- queries, window probes, staging, then scan, look-back, stores -- and the waves of ONE workgroup
// go through them together, so a CU with one 1024-thread workgroup idles at every link; four small workgroups are in different
// phases of different tiles (1024-thread workgroups: 661 us per 64M queries in order; GTARS_TOK_SWEEP_TPB=1024 is the A/B switch
// of the main variant).
template <int TPB, int R, bool FILTER, bool IMPL, bool REV>
static gtars_status launch_sweep_t(const AccelView &a, const u32 *qc, const u32 *qs, const u32 *qe, u64 nq, i32 min_bp, const EnumOut &out,
                                   ScanWs *ws, ScanEpoch &ep, const u64 *d_base, u64 *d_total_out, hipStream_t st) {
    static KernelSetup setup;
    int dev = 0, cus = 256;
    gtars_status s0 = setup.get(reinterpret_cast<const void *>(k_tok_sweep<TPB, R, FILTER, IMPL, REV>), dev, cus);
    if (s0) return s0;
    // LDS: one region per wave -- 20 bytes per staged block in the count phase, the wave's id staging in the write phase.  9.5 KB per
    // wave = 486 blocks = 972 intervals under a wave's 256 queries.  (GTARS_TOK_SWEEP_BLOCKS / _RUNS: tests -- small budgets force
    // the global path.)
    const u32 cap_w = (u32)std::min(486, std::max(1, env_int("GTARS_TOK_SWEEP_BLOCKS", 486)));
    const u32 max_runs = (u32)std::min<int>((int)SWP_RUNS, std::max(0, env_int("GTARS_TOK_SWEEP_RUNS", (int)SWP_RUNS)));
    const u32 region_words = 2432;  // (>= 5 * 486, a multiple of 4)
    const size_t lds = (size_t)region_words * (TPB / 64) * 4;
    const u32 spin_limit = (u32)env_int("GTARS_TOK_SPIN_LIMIT", 4096);
    const u64 tile_q = (u64)TPB * 4 * R;
    const u64 tiles = (nq + tile_q - 1) / tile_q;
    const u64 grid = std::min<u64>((u64)cus * (1024 / TPB), tiles);
    const u64 cap = out.vals ? out.capacity : 0;
    hipLaunchKernelGGL((k_tok_sweep<TPB, R, FILTER, IMPL, REV>), dim3((unsigned)grid), dim3(TPB), lds, st, a, qc, qs, qe, nq, min_bp,
                       out.offsets, out.vals, cap, ws, ep.epoch, ep.ticket_base, region_words, spin_limit, d_base, d_total_out, cap_w, max_runs);
    GT_HIP(hipGetLastError());
    if (tiles > grid) ep.ticket_base += (u32)tiles;
    return GTARS_OK;
}

gtars_status launch_tokenize_sweep(const AccelView &a, const u32 *qc, const u32 *qs, const u32 *qe, u64 nq, int has_min, i32 min_overlap,
                                   const EnumOut &out, void *scan_ws, size_t scan_ws_bytes, ScanEpoch &ep, hipStream_t st,
                                   const u64 *d_base, u64 *d_total_out, bool reverse) {
    if (out.starts || out.ends) return fail(GTARS_ERR_INTERNAL, "the sweep tokenizer writes vals only");
    if (nq == 0)  // (the empty batch: offsets[0] and the totals, no kernel)
        return launch_tokenize_lds(a, qc, qs, qe, nq, has_min, min_overlap, out, scan_ws, scan_ws_bytes, ep, st, d_base, d_total_out, reverse);
    const bool impl = a.ids_affine != 0;
    const bool filter = has_min && min_overlap > 1;
    int cus = 256;
    {
        int dev = 0;
        GT_HIP(hipGetDevice(&dev));
        GT_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    }
    // rounds per tile: 2 once every workgroup has several two-round tiles (the per-tile costs -- scan barrier, ticket, look-back --
    // paid half as often, a wave's rounds pipelined; 4 rounds keep 88 registers of query state across the scan: scratch);
    // GTARS_TOK_SWEEP_ROUNDS / _TPB: tests, A/B
    const int tpb = env_int("GTARS_TOK_SWEEP_TPB", 256) == 1024 && !filter && impl && !reverse ? 1024 : 256;
    int rounds = !filter && nq >= (u64)cus * 8192ull * 2ull ? 2 : 1;
    {
        const int forced = env_int("GTARS_TOK_SWEEP_ROUNDS", 0);
        if (forced == 1 || (forced == 2 && !filter)) rounds = forced;
    }
    if ((u64)std::max<u32>(a.max_chrom_n, 1) * 4ull * (u64)tpb * (u64)rounds > 0xFFFFFFFFull) rounds = 1;  // a tile's 32-bit hit total
    const u64 tile_q = 4ull * (u64)tpb * (u64)rounds;
    const u64 tiles = (nq + tile_q - 1) / tile_q;
    if (tiles > 0xFFFFFFF0ull) return fail(GTARS_ERR_INVALID_ARG, "query batch too large for one launch");
    const size_t need = scan_ws_bytes_for_tiles(tiles);
    if (scan_ws_bytes < need) return fail(GTARS_ERR_INTERNAL, "fused scan workspace too small");
    if (ep.epoch == 0 || ep.epoch >= EP_MAX || need > ep.cleared_bytes) {
        GT_HIP(hipMemsetAsync(scan_ws, 0, scan_ws_bytes, st));
        ep.cleared_bytes = scan_ws_bytes;
        ep.epoch = 0;
        ep.ticket_base = 0;
    }
    ep.epoch += 1;
    const i32 min_bp = has_min ? min_overlap : 0;
    ScanWs *ws = (ScanWs *)scan_ws;
    prof_note_fact("tok_build_sweep");
    ProfScope p("k_tok_sweep", st);
#define GT_SWEEP_CASE(T, N, F, I, V) \
    if (tpb == T && rounds == N && filter == F && impl == I && reverse == V) \
        return launch_sweep_t<T, N, F, I, V>(a, qc, qs, qe, nq, min_bp, out, ws, ep, d_base, d_total_out, st);
    GT_SWEEP_CASE(256, 1, false, true, false)
    GT_SWEEP_CASE(256, 1, false, true, true)
    GT_SWEEP_CASE(256, 1, true, true, false)
    GT_SWEEP_CASE(256, 1, true, true, true)
    GT_SWEEP_CASE(256, 1, false, false, false)
    GT_SWEEP_CASE(256, 1, false, false, true)
    GT_SWEEP_CASE(256, 1, true, false, false)
    GT_SWEEP_CASE(256, 1, true, false, true)
    GT_SWEEP_CASE(256, 2, false, true, false)
    GT_SWEEP_CASE(256, 2, false, true, true)
    GT_SWEEP_CASE(256, 2, false, false, false)
    GT_SWEEP_CASE(256, 2, false, false, true)
    GT_SWEEP_CASE(1024, 1, false, true, false)
    GT_SWEEP_CASE(1024, 2, false, true, false)
#undef GT_SWEEP_CASE
    return fail(GTARS_ERR_INTERNAL, "unsupported sweep tokenizer variant");
}

}  // namespace gtars

#if GTARS_TOK_STAMPS
extern "C" int gtars_debug_tok_stamps(unsigned long long *out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(gtars::g_tok_stamps), 192) != hipSuccess) return 1;
    if (reset) {
        unsigned long long z[24] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(gtars::g_tok_stamps), z, 192) != hipSuccess) return 1;
    }
    return 0;
}
#endif
