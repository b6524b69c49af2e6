/*
 * gtars_oracle.c -- TEST INFRASTRUCTURE ONLY (see gtars_oracle.h).
 *
 * CPU restatement of the reference algorithms, written to follow the
 * reference control flow step by step (same searches, same scans, same break
 * conditions) rather than to be fast.  Single-threaded, plain C99.
 */
#include "gtars_oracle.h"

#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ utils */

static void *xmalloc(size_t n) {
    void *p = malloc(n ? n : 1);
    if (!p) abort();
    return p;
}
static void *xcalloc(size_t n, size_t s) {
    void *p = calloc(n ? n : 1, s ? s : 1);
    if (!p) abort();
    return p;
}
static void *xrealloc(void *p, size_t n) {
    void *q = realloc(p, n ? n : 1);
    if (!q) abort();
    return q;
}

uint64_t orc_splitmix64(uint64_t *state) {
    uint64_t z = (*state += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

/* One interval as the reference stores it: Interval<u32,u32>{start,end,val}
 * (gtars-core/src/models/interval.rs:8-16). */
typedef struct {
    uint32_t start, end, val;
} iv_t;

/* Interval::overlap (interval.rs:47-50) */
static int iv_overlap(const iv_t *iv, uint32_t start, uint32_t end) {
    return iv->start < end && iv->end > start;
}

/* Stable merge sort of iv_t (Rust's slice::sort / sort_by_key are stable).
 * by_end != 0: key (start,end) = Interval::cmp (interval.rs:18-31);
 * by_end == 0: key start only (ailist.rs:111). */
static int iv_less(const iv_t *a, const iv_t *b, int by_end) {
    if (a->start != b->start) return a->start < b->start;
    if (by_end) return a->end < b->end;
    return 0;
}
static void iv_msort(iv_t *a, iv_t *tmp, size_t n, int by_end) {
    if (n < 2) return;
    size_t h = n / 2;
    iv_msort(a, tmp, h, by_end);
    iv_msort(a + h, tmp, n - h, by_end);
    size_t i = 0, j = h, k = 0;
    while (i < h && j < n) {
        /* take right only if strictly less -> stable */
        if (iv_less(&a[j], &a[i], by_end))
            tmp[k++] = a[j++];
        else
            tmp[k++] = a[i++];
    }
    while (i < h) tmp[k++] = a[i++];
    while (j < n) tmp[k++] = a[j++];
    memcpy(a, tmp, n * sizeof(iv_t));
}
static void iv_stable_sort(iv_t *a, size_t n, int by_end) {
    iv_t *tmp = (iv_t *)xmalloc(n * sizeof(iv_t));
    iv_msort(a, tmp, n, by_end);
    free(tmp);
}

static int cmp_u32(const void *a, const void *b) {
    uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
    return (x > y) - (x < y);
}
static int cmp_u64(const void *a, const void *b) {
    uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return (x > y) - (x < y);
}

/* ------------------------------------------------------------------- Bits */

typedef struct {
    iv_t *intervals; /* sorted (start,end), stable */
    uint32_t *starts; /* sorted */
    uint32_t *ends;   /* sorted */
    uint32_t max_len;
    size_t n;
} bits_t;

/* Bits::build (bits.rs:101-128) */
static void bits_build(bits_t *b, iv_t *ivs, size_t n) {
    b->n = n;
    b->intervals = ivs; /* takes ownership */
    iv_stable_sort(b->intervals, n, 1);
    b->starts = (uint32_t *)xmalloc(n * sizeof(uint32_t));
    b->ends = (uint32_t *)xmalloc(n * sizeof(uint32_t));
    for (size_t i = 0; i < n; i++) {
        b->starts[i] = ivs[i].start;
        b->ends[i] = ivs[i].end;
    }
    qsort(b->starts, n, sizeof(uint32_t), cmp_u32);
    qsort(b->ends, n, sizeof(uint32_t), cmp_u32);
    uint32_t max_len = 0;
    for (size_t i = 0; i < n; i++) {
        /* checked_sub(...).unwrap_or(0) */
        uint32_t len = ivs[i].end >= ivs[i].start ? ivs[i].end - ivs[i].start : 0;
        if (len > max_len) max_len = len;
    }
    b->max_len = max_len;
}

/* Bits::lower_bound (bits.rs:250-264) -- same probe sequence */
static size_t bits_lower_bound(uint32_t start, const iv_t *intervals, size_t n) {
    size_t size = n, low = 0;
    while (size > 0) {
        size_t half = size / 2;
        size_t other_half = size - half;
        size_t probe = low + half;
        size_t other_low = low + other_half;
        const iv_t *v = &intervals[probe];
        size = half;
        low = v->start < start ? other_low : low;
    }
    return low;
}

/* Bits::bsearch_seq_ref (bits.rs:304-322) */
static size_t bits_bsearch_seq(uint32_t key, const uint32_t *elems, size_t n) {
    if (n == 0 || elems[0] >= key) return 0;
    if (elems[n - 1] < key) return n;
    size_t cursor = 0, length = n;
    while (length > 1) {
        size_t half = length >> 1;
        length -= half;
        cursor += (size_t)(elems[cursor + half - 1] < key) * half;
    }
    return cursor;
}

/* Bits::find (bits.rs:141-156) + IterFind::next (bits.rs:433-446) */
static uint64_t bits_find(const bits_t *b, uint32_t start, uint32_t stop,
                          uint32_t *os, uint32_t *oe, uint32_t *ov, uint64_t cap) {
    uint32_t key = start >= b->max_len ? start - b->max_len : 0; /* checked_sub */
    size_t off = bits_lower_bound(key, b->intervals, b->n);
    uint64_t k = 0;
    while (off < b->n) {
        const iv_t *iv = &b->intervals[off];
        off += 1;
        if (iv_overlap(iv, start, stop)) {
            if (k < cap) {
                if (os) os[k] = iv->start;
                if (oe) oe[k] = iv->end;
                if (ov) ov[k] = iv->val;
            }
            k++;
        } else if (iv->start >= stop) {
            break;
        }
    }
    return k;
}

/* Bits::count (bits.rs:337-344).  `start + 1` wraps like release-mode Rust. */
static uint64_t bits_count(const bits_t *b, uint32_t start, uint32_t stop) {
    size_t len = b->n;
    size_t first = bits_bsearch_seq(start + 1u, b->ends, len);
    size_t last = bits_bsearch_seq(stop, b->starts, len);
    size_t num_cant_after = len - last;
    return (uint64_t)(len - first - num_cant_after);
}

static void bits_free(bits_t *b) {
    free(b->intervals);
    free(b->starts);
    free(b->ends);
}

/* ----------------------------------------------------------------- AIList */

typedef struct {
    uint32_t *starts, *ends, *max_ends;
    iv_t *stored;
    size_t *header; /* header_list */
    size_t n, n_header;
} ailist_t;

/* AIList::build (ailist.rs:105-151) + decompose (ailist.rs:198-236) */
static void ailist_build(ailist_t *a, iv_t *ivs, size_t n) {
    const size_t minimum_coverage_length = 10;
    iv_stable_sort(ivs, n, 0);
    a->n = n;
    a->starts = (uint32_t *)xmalloc(n * sizeof(uint32_t));
    a->ends = (uint32_t *)xmalloc(n * sizeof(uint32_t));
    a->max_ends = (uint32_t *)xmalloc(n * sizeof(uint32_t));
    a->stored = (iv_t *)xmalloc(n * sizeof(iv_t));
    size_t hcap = 8;
    a->header = (size_t *)xmalloc(hcap * sizeof(size_t));
    a->header[0] = 0;
    a->n_header = 1;

    iv_t *cur = ivs; /* owned */
    size_t ncur = n;
    iv_t *l2 = (iv_t *)xmalloc(n * sizeof(iv_t));
    size_t filled = 0;
    for (;;) {
        size_t nl2 = 0, first = filled;
        for (size_t index = 0; index < ncur; index++) {
            const iv_t *interval = &cur[index];
            size_t count = 0;
            for (size_t i = 1; i < minimum_coverage_length * 2; i++) {
                if (index + i >= ncur) break;
                if (interval->end > cur[index + i].end) count++;
            }
            if (count >= minimum_coverage_length) {
                l2[nl2++] = *interval;
            } else {
                a->starts[filled] = interval->start;
                a->ends[filled] = interval->end;
                a->stored[filled] = *interval;
                filled++;
            }
        }
        uint32_t mx = 0;
        for (size_t i = first; i < filled; i++) {
            mx = mx > a->ends[i] ? mx : a->ends[i];
            a->max_ends[i] = mx;
        }
        /* swap(intervals, l2) */
        iv_t *t = cur;
        cur = l2;
        l2 = t;
        ncur = nl2;
        if (ncur == 0) break;
        if (a->n_header == hcap) {
            hcap *= 2;
            a->header = (size_t *)xrealloc(a->header, hcap * sizeof(size_t));
        }
        a->header[a->n_header++] = filled;
    }
    free(cur);
    free(l2);
}

/* slice::partition_point(|x| x < end) */
static size_t partition_point_lt(const uint32_t *a, size_t n, uint32_t key) {
    size_t lo = 0, hi = n;
    while (lo < hi) {
        size_t mid = lo + (hi - lo) / 2;
        if (a[mid] < key)
            lo = mid + 1;
        else
            hi = mid;
    }
    return lo;
}

/* AIList::find (ailist.rs:153-178) + query_slice (ailist.rs:238-263) */
static uint64_t ailist_find(const ailist_t *a, uint32_t start, uint32_t end,
                            uint32_t *os, uint32_t *oe, uint32_t *ov, uint64_t cap) {
    uint64_t k = 0;
    for (size_t h = 0; h < a->n_header; h++) {
        size_t lo = a->header[h];
        size_t hi = (h + 1 < a->n_header) ? a->header[h + 1] : a->n;
        const uint32_t *starts = a->starts + lo, *ends = a->ends + lo,
                       *max_ends = a->max_ends + lo;
        const iv_t *stored = a->stored + lo;
        size_t i = partition_point_lt(starts, hi - lo, end);
        while (i > 0) {
            i -= 1;
            if (start >= ends[i]) {
                if (start > max_ends[i]) break; /* return results_list */
            } else {
                if (k < cap) {
                    if (os) os[k] = stored[i].start;
                    if (oe) oe[k] = stored[i].end;
                    if (ov) ov[k] = stored[i].val;
                }
                k++;
            }
        }
    }
    return k;
}

static void ailist_free(ailist_t *a) {
    free(a->starts);
    free(a->ends);
    free(a->max_ends);
    free(a->stored);
    free(a->header);
}

/* ------------------------------------------------------ multi-chrom index */

struct orc_index {
    int kind;
    uint32_t n_chrom;
    uint8_t *present; /* chromosome has an entry in the map */
    bits_t *bits;
    ailist_t *ail;
};

orc_index *orc_index_build(const uint32_t *chrom, const uint32_t *start,
                           const uint32_t *end, const uint32_t *val, uint64_t n,
                           uint32_t n_chrom, int kind) {
    orc_index *ix = (orc_index *)xcalloc(1, sizeof(*ix));
    ix->kind = kind;
    ix->n_chrom = n_chrom;
    ix->present = (uint8_t *)xcalloc(n_chrom, 1);
    ix->bits = (bits_t *)xcalloc(n_chrom, sizeof(bits_t));
    ix->ail = (ailist_t *)xcalloc(n_chrom, sizeof(ailist_t));
    size_t *cnt = (size_t *)xcalloc(n_chrom, sizeof(size_t));
    for (uint64_t i = 0; i < n; i++)
        if (chrom[i] < n_chrom) cnt[chrom[i]]++;
    iv_t **buf = (iv_t **)xcalloc(n_chrom, sizeof(iv_t *));
    size_t *fill = (size_t *)xcalloc(n_chrom, sizeof(size_t));
    for (uint32_t c = 0; c < n_chrom; c++)
        if (cnt[c]) buf[c] = (iv_t *)xmalloc(cnt[c] * sizeof(iv_t));
    /* per-chrom vectors receive intervals in input order
     * (utils/mod.rs:57-87: chr_intervals.push(interval)) */
    for (uint64_t i = 0; i < n; i++) {
        uint32_t c = chrom[i];
        if (c >= n_chrom) continue;
        iv_t *p = &buf[c][fill[c]++];
        p->start = start[i];
        p->end = end[i];
        p->val = val ? val[i] : (uint32_t)i;
    }
    for (uint32_t c = 0; c < n_chrom; c++) {
        if (!cnt[c]) continue;
        ix->present[c] = 1;
        if (kind == ORC_KIND_BITS)
            bits_build(&ix->bits[c], buf[c], cnt[c]);
        else
            ailist_build(&ix->ail[c], buf[c], cnt[c]);
    }
    free(cnt);
    free(buf);
    free(fill);
    return ix;
}

void orc_index_free(orc_index *ix) {
    if (!ix) return;
    for (uint32_t c = 0; c < ix->n_chrom; c++) {
        if (!ix->present[c]) continue;
        if (ix->kind == ORC_KIND_BITS)
            bits_free(&ix->bits[c]);
        else
            ailist_free(&ix->ail[c]);
    }
    free(ix->present);
    free(ix->bits);
    free(ix->ail);
    free(ix);
}

static int ix_has(const orc_index *ix, uint32_t c) {
    return c < ix->n_chrom && ix->present[c];
}

uint64_t orc_index_chrom_len(const orc_index *ix, uint32_t c) {
    if (!ix_has(ix, c)) return 0;
    return ix->kind == ORC_KIND_BITS ? ix->bits[c].n : ix->ail[c].n;
}
uint32_t orc_index_max_len(const orc_index *ix, uint32_t c) {
    if (!ix_has(ix, c) || ix->kind != ORC_KIND_BITS) return 0;
    return ix->bits[c].max_len;
}
uint64_t orc_index_n_headers(const orc_index *ix, uint32_t c) {
    if (!ix_has(ix, c) || ix->kind != ORC_KIND_AILIST) return 0;
    return ix->ail[c].n_header;
}
void orc_index_headers(const orc_index *ix, uint32_t c, uint64_t *out) {
    if (!ix_has(ix, c) || ix->kind != ORC_KIND_AILIST) return;
    for (size_t i = 0; i < ix->ail[c].n_header; i++) out[i] = ix->ail[c].header[i];
}
void orc_index_stored(const orc_index *ix, uint32_t c, uint32_t *start,
                      uint32_t *end, uint32_t *val) {
    if (!ix_has(ix, c)) return;
    const iv_t *s = ix->kind == ORC_KIND_BITS ? ix->bits[c].intervals : ix->ail[c].stored;
    size_t n = ix->kind == ORC_KIND_BITS ? ix->bits[c].n : ix->ail[c].n;
    for (size_t i = 0; i < n; i++) {
        if (start) start[i] = s[i].start;
        if (end) end[i] = s[i].end;
        if (val) val[i] = s[i].val;
    }
}

uint64_t orc_find(const orc_index *ix, uint32_t c, uint32_t qs, uint32_t qe,
                  uint32_t *os, uint32_t *oe, uint32_t *ov, uint64_t cap) {
    if (!ix_has(ix, c)) return 0; /* core.get(&chr) == None */
    if (ix->kind == ORC_KIND_BITS) return bits_find(&ix->bits[c], qs, qe, os, oe, ov, cap);
    return ailist_find(&ix->ail[c], qs, qe, os, oe, ov, cap);
}

uint64_t orc_bits_count(const orc_index *ix, uint32_t c, uint32_t qs, uint32_t qe) {
    if (!ix_has(ix, c) || ix->kind != ORC_KIND_BITS) return 0;
    return bits_count(&ix->bits[c], qs, qe);
}

/* Tokenizer::tokenize inner loop (tokenizer.rs:141-156) */
uint64_t orc_tokenize(const orc_index *ix, const uint32_t *qc, const uint32_t *qs,
                      const uint32_t *qe, uint64_t nq, uint64_t *offsets,
                      uint32_t *ids, uint64_t cap) {
    uint64_t h = 0;
    for (uint64_t q = 0; q < nq; q++) {
        if (offsets) offsets[q] = h;
        uint64_t room = h < cap ? cap - h : 0;
        h += orc_find(ix, qc[q], qs[q], qe[q], NULL, NULL,
                      (ids && room) ? ids + h : NULL, room);
    }
    if (offsets) offsets[nq] = h;
    return h;
}

/* overlap_bp (multi_chrom_overlapper.rs:561-563) */
static int64_t overlap_bp(uint32_t as, uint32_t ae, uint32_t bs, uint32_t be) {
    uint32_t mn = ae < be ? ae : be;
    uint32_t mx = as > bs ? as : bs;
    return (int64_t)mn - (int64_t)mx;
}

/* Shared walker: find hits of one query, apply the min_overlap filter
 * exactly as multi_chrom_overlapper.rs:491 / :511 / :537 do
 * (min_bp <= 1 || overlap_bp >= min_bp). */
typedef struct {
    uint32_t *s, *e, *v;
    uint64_t cap;
} scratch_t;

static void scratch_reserve(scratch_t *sc, uint64_t n) {
    if (n <= sc->cap) return;
    uint64_t c = sc->cap ? sc->cap : 64;
    while (c < n) c *= 2;
    sc->s = (uint32_t *)xrealloc(sc->s, c * sizeof(uint32_t));
    sc->e = (uint32_t *)xrealloc(sc->e, c * sizeof(uint32_t));
    sc->v = (uint32_t *)xrealloc(sc->v, c * sizeof(uint32_t));
    sc->cap = c;
}
static void scratch_free(scratch_t *sc) {
    free(sc->s);
    free(sc->e);
    free(sc->v);
}

static uint64_t query_filtered(const orc_index *ix, scratch_t *sc, uint32_t c,
                               uint32_t qs, uint32_t qe, int has_min,
                               int32_t min_overlap) {
    uint64_t n = orc_find(ix, c, qs, qe, sc->s, sc->e, sc->v, sc->cap);
    if (n > sc->cap) {
        scratch_reserve(sc, n);
        n = orc_find(ix, c, qs, qe, sc->s, sc->e, sc->v, sc->cap);
    }
    int32_t min_bp = has_min ? min_overlap : 0; /* min_overlap.unwrap_or(0) */
    if (min_bp <= 1) return n;
    uint64_t k = 0;
    for (uint64_t i = 0; i < n; i++) {
        if (overlap_bp(qs, qe, sc->s[i], sc->e[i]) >= (int64_t)min_bp) {
            sc->s[k] = sc->s[i];
            sc->e[k] = sc->e[i];
            sc->v[k] = sc->v[i];
            k++;
        }
    }
    return k;
}

void orc_count_overlaps(const orc_index *ix, const uint32_t *qc, const uint32_t *qs,
                        const uint32_t *qe, uint64_t nq, int has_min,
                        int32_t min_overlap, uint64_t *counts) {
    scratch_t sc = {0};
    scratch_reserve(&sc, 64);
    for (uint64_t q = 0; q < nq; q++)
        counts[q] = query_filtered(ix, &sc, qc[q], qs[q], qe[q], has_min, min_overlap);
    scratch_free(&sc);
}

void orc_any_overlaps(const orc_index *ix, const uint32_t *qc, const uint32_t *qs,
                      const uint32_t *qe, uint64_t nq, int has_min,
                      int32_t min_overlap, uint8_t *out) {
    scratch_t sc = {0};
    scratch_reserve(&sc, 64);
    for (uint64_t q = 0; q < nq; q++)
        out[q] = query_filtered(ix, &sc, qc[q], qs[q], qe[q], has_min, min_overlap) > 0;
    scratch_free(&sc);
}

uint64_t orc_find_overlaps_regions(const orc_index *ix, const uint32_t *qc,
                                   const uint32_t *qs, const uint32_t *qe,
                                   uint64_t nq, int has_min, int32_t min_overlap,
                                   uint64_t *offsets, uint32_t *out_start,
                                   uint32_t *out_end, uint32_t *out_val,
                                   uint64_t cap) {
    scratch_t sc = {0};
    scratch_reserve(&sc, 64);
    uint64_t h = 0;
    for (uint64_t q = 0; q < nq; q++) {
        if (offsets) offsets[q] = h;
        uint64_t n = query_filtered(ix, &sc, qc[q], qs[q], qe[q], has_min, min_overlap);
        for (uint64_t i = 0; i < n; i++, h++) {
            if (h < cap) {
                if (out_start) out_start[h] = sc.s[i];
                if (out_end) out_end[h] = sc.e[i];
                if (out_val) out_val[h] = sc.v[i];
            }
        }
    }
    if (offsets) offsets[nq] = h;
    scratch_free(&sc);
    return h;
}

/* IndexedRegionSet::find_overlaps (indexed_region_set.rs:246-263) with
 * coord_lookup (indexed_region_set.rs:145-152) realised as a sorted table. */
typedef struct {
    uint32_t c, s, e;
    uint64_t idx;
} coord_t;
static int cmp_coord(const void *a, const void *b) {
    const coord_t *x = (const coord_t *)a, *y = (const coord_t *)b;
    if (x->c != y->c) return (x->c > y->c) - (x->c < y->c);
    if (x->s != y->s) return (x->s > y->s) - (x->s < y->s);
    if (x->e != y->e) return (x->e > y->e) - (x->e < y->e);
    return (x->idx > y->idx) - (x->idx < y->idx);
}
static int cmp_coord_key(const coord_t *x, uint32_t c, uint32_t s, uint32_t e) {
    if (x->c != c) return x->c < c ? -1 : 1;
    if (x->s != s) return x->s < s ? -1 : 1;
    if (x->e != e) return x->e < e ? -1 : 1;
    return 0;
}

uint64_t orc_irs_find_overlaps(const orc_index *ix, const uint32_t *src_chrom,
                               const uint32_t *src_start, const uint32_t *src_end,
                               uint64_t n_src, const uint32_t *qc, const uint32_t *qs,
                               const uint32_t *qe, uint64_t nq, int has_min,
                               int32_t min_overlap, uint64_t *offsets,
                               uint64_t *out_idx, uint64_t cap) {
    coord_t *tab = (coord_t *)xmalloc(n_src * sizeof(coord_t));
    for (uint64_t i = 0; i < n_src; i++) {
        tab[i].c = src_chrom[i];
        tab[i].s = src_start[i];
        tab[i].e = src_end[i];
        tab[i].idx = i;
    }
    qsort(tab, n_src, sizeof(coord_t), cmp_coord);
    scratch_t sc = {0};
    scratch_reserve(&sc, 64);
    uint64_t *idxs = NULL, idx_cap = 0;
    uint64_t h = 0;
    for (uint64_t q = 0; q < nq; q++) {
        if (offsets) offsets[q] = h;
        uint64_t n = query_filtered(ix, &sc, qc[q], qs[q], qe[q], has_min, min_overlap);
        uint64_t m = 0;
        for (uint64_t i = 0; i < n; i++) {
            /* lookup.get(&(chr, start, end)) -> all source rows with these coords */
            uint64_t lo = 0, hi = n_src;
            while (lo < hi) {
                uint64_t mid = lo + (hi - lo) / 2;
                if (cmp_coord_key(&tab[mid], qc[q], sc.s[i], sc.e[i]) < 0)
                    lo = mid + 1;
                else
                    hi = mid;
            }
            while (lo < n_src && cmp_coord_key(&tab[lo], qc[q], sc.s[i], sc.e[i]) == 0) {
                if (m == idx_cap) {
                    idx_cap = idx_cap ? idx_cap * 2 : 64;
                    idxs = (uint64_t *)xrealloc(idxs, idx_cap * sizeof(uint64_t));
                }
                idxs[m++] = tab[lo].idx;
                lo++;
            }
        }
        /* idxs.sort_unstable(); idxs.dedup(); */
        qsort(idxs, m, sizeof(uint64_t), cmp_u64);
        uint64_t prev = 0;
        for (uint64_t i = 0; i < m; i++) {
            if (i && idxs[i] == prev) continue;
            prev = idxs[i];
            if (h < cap && out_idx) out_idx[h] = idxs[i];
            h++;
        }
    }
    if (offsets) offsets[nq] = h;
    free(idxs);
    free(tab);
    scratch_free(&sc);
    return h;
}

/* -------------------------------------------------------------------- IGD */

typedef struct {
    uint32_t file_idx;
    int32_t start, end, value;
} rec_t; /* igd.rs:19-30 */

typedef struct {
    rec_t *records;
    size_t n, cap;
} tile_t;

typedef struct {
    uint32_t chrom;
    tile_t *tiles;
    size_t n_tiles;
} contig_t;

struct orc_igd {
    int32_t nbp;
    contig_t *contigs;
    size_t n_contigs, cap_contigs;
    int finalized;
};

orc_igd *orc_igd_new(int32_t nbp) {
    orc_igd *g = (orc_igd *)xcalloc(1, sizeof(*g));
    g->nbp = nbp > 0 ? nbp : 16384;
    return g;
}

void orc_igd_free(orc_igd *g) {
    if (!g) return;
    for (size_t c = 0; c < g->n_contigs; c++) {
        for (size_t t = 0; t < g->contigs[c].n_tiles; t++) free(g->contigs[c].tiles[t].records);
        free(g->contigs[c].tiles);
    }
    free(g->contigs);
    free(g);
}

static contig_t *igd_contig(const orc_igd *g, uint32_t chrom) {
    for (size_t c = 0; c < g->n_contigs; c++)
        if (g->contigs[c].chrom == chrom) return &g->contigs[c];
    return NULL;
}

/* Igd::add (igd.rs:109-153) */
void orc_igd_add(orc_igd *g, uint32_t chrom, int32_t start, int32_t end,
                 int32_t value, uint32_t file_idx) {
    if (g->finalized) abort(); /* assert!(!self.finalized) */
    if (start < 0 || end < 0 || start >= end) return;
    int32_t n1 = start / g->nbp;
    int32_t n2 = (end - 1) / g->nbp;
    size_t needed = (size_t)(n2 + 1);
    contig_t *ctg = igd_contig(g, chrom);
    if (!ctg) {
        if (g->n_contigs == g->cap_contigs) {
            g->cap_contigs = g->cap_contigs ? g->cap_contigs * 2 : 32;
            g->contigs = (contig_t *)xrealloc(g->contigs, g->cap_contigs * sizeof(contig_t));
        }
        ctg = &g->contigs[g->n_contigs++];
        ctg->chrom = chrom;
        ctg->tiles = NULL;
        ctg->n_tiles = 0;
    }
    if (ctg->n_tiles < needed) {
        ctg->tiles = (tile_t *)xrealloc(ctg->tiles, needed * sizeof(tile_t));
        memset(ctg->tiles + ctg->n_tiles, 0, (needed - ctg->n_tiles) * sizeof(tile_t));
        ctg->n_tiles = needed;
    }
    rec_t r = {file_idx, start, end, value};
    for (int32_t i = n1; i <= n2; i++) {
        tile_t *t = &ctg->tiles[i];
        if (t->n == t->cap) {
            t->cap = t->cap ? t->cap * 2 : 4;
            t->records = (rec_t *)xrealloc(t->records, t->cap * sizeof(rec_t));
        }
        t->records[t->n++] = r;
    }
}

/* n calls of orc_igd_add in array order (batch form for the Python binding: same insertion order, same rules) */
void orc_igd_add_arrays(orc_igd *g, const uint32_t *chrom, const int32_t *start, const int32_t *end,
                        const int32_t *value, const uint32_t *file_idx, uint64_t n) {
    for (uint64_t i = 0; i < n; i++) orc_igd_add(g, chrom[i], start[i], end[i], value[i], file_idx[i]);
}

static void rec_msort(rec_t *a, rec_t *tmp, size_t n) {
    if (n < 2) return;
    size_t h = n / 2;
    rec_msort(a, tmp, h);
    rec_msort(a + h, tmp, n - h);
    size_t i = 0, j = h, k = 0;
    while (i < h && j < n) {
        if (a[j].start < a[i].start)
            tmp[k++] = a[j++];
        else
            tmp[k++] = a[i++];
    }
    while (i < h) tmp[k++] = a[i++];
    while (j < n) tmp[k++] = a[j++];
    memcpy(a, tmp, n * sizeof(rec_t));
}

/* Igd::finalize (igd.rs:157-167): stable sort_by_key(start) per tile */
void orc_igd_finalize(orc_igd *g) {
    if (g->finalized) return;
    for (size_t c = 0; c < g->n_contigs; c++) {
        for (size_t t = 0; t < g->contigs[c].n_tiles; t++) {
            tile_t *tl = &g->contigs[c].tiles[t];
            if (tl->n < 2) continue;
            rec_t *tmp = (rec_t *)xmalloc(tl->n * sizeof(rec_t));
            rec_msort(tl->records, tmp, tl->n);
            free(tmp);
        }
    }
    g->finalized = 1;
}

uint64_t orc_igd_total_records(const orc_igd *g) {
    uint64_t s = 0;
    for (size_t c = 0; c < g->n_contigs; c++)
        for (size_t t = 0; t < g->contigs[c].n_tiles; t++) s += g->contigs[c].tiles[t].n;
    return s;
}
uint64_t orc_igd_num_contigs(const orc_igd *g) { return g->n_contigs; }

typedef void (*hit_fn)(const rec_t *rec, void *ctx);

/* Igd::walk_tile_overlaps (igd.rs:753-847) -- literal */
static void walk_tile_overlaps(const contig_t *contig, int32_t start, int32_t end,
                               int32_t min_overlap, int32_t nbp, hit_fn on_hit,
                               void *ctx) {
    int32_t n_tiles = (int32_t)contig->n_tiles;
    int32_t n1 = start / nbp;
    int32_t n2 = (end - 1) / nbp;
    if (n1 >= n_tiles) return;
    if (n1 < 0) return; /* the reference would panic (index out of bounds) */
    if (n2 > n_tiles - 1) n2 = n_tiles - 1;

    const tile_t *tile = &contig->tiles[n1];
    if (tile->n != 0 && end > tile->records[0].start) {
        int32_t tl = 0, tr = (int32_t)tile->n - 1;
        while (tl < tr - 1) {
            int32_t tm = (tl + tr) / 2;
            if (tile->records[tm].start < end)
                tl = tm;
            else
                tr = tm;
        }
        if (tile->records[tr].start < end) tl = tr;
        for (int32_t i = tl; i >= 0; i--) {
            const rec_t *rec = &tile->records[i];
            int32_t mn = rec->end < end ? rec->end : end;
            int32_t mx = rec->start > start ? rec->start : start;
            int32_t ov = mn - mx;
            if (ov >= min_overlap) on_hit(rec, ctx);
        }
    }
    if (n2 > n1) {
        int32_t bd = nbp * (n1 + 1);
        for (int32_t j = n1 + 1; j <= n2; j++) {
            tile = &contig->tiles[j];
            if (tile->n == 0) {
                bd += nbp;
                continue;
            }
            if (end > tile->records[0].start) {
                int32_t ts = 0;
                while (ts < (int32_t)tile->n && tile->records[ts].start < bd) ts++;
                int32_t tl = 0, tr = (int32_t)tile->n - 1;
                while (tl < tr - 1) {
                    int32_t tm = (tl + tr) / 2;
                    if (tile->records[tm].start < end)
                        tl = tm;
                    else
                        tr = tm;
                }
                if (tile->records[tr].start < end) tl = tr;
                for (int32_t i = tl; i >= ts; i--) {
                    const rec_t *rec = &tile->records[i];
                    int32_t mn = rec->end < end ? rec->end : end;
                    int32_t mx = rec->start > start ? rec->start : start;
                    int32_t ov = mn - mx;
                    if (ov >= min_overlap) on_hit(rec, ctx);
                }
            }
            bd += nbp;
        }
    }
}

typedef struct {
    uint64_t *hits;
    uint32_t total;
} count_ctx;
static void on_count(const rec_t *rec, void *p) {
    count_ctx *c = (count_ctx *)p;
    c->hits[rec->file_idx] += 1;
    c->total += 1;
}

/* Igd::count_overlaps (igd.rs:504-540) */
uint32_t orc_igd_count_overlaps(const orc_igd *g, uint32_t chrom, int32_t start,
                                int32_t end, int32_t min_overlap, uint64_t *hits) {
    if (!g->finalized) abort();
    if (start >= end || end <= 0) return 0;
    if (start < 0) start = 0;
    const contig_t *ctg = igd_contig(g, chrom);
    if (!ctg) return 0;
    count_ctx c = {hits, 0};
    walk_tile_overlaps(ctg, start, end, min_overlap, g->nbp, on_count, &c);
    return c.total;
}

/* Igd::count_set_overlaps (igd.rs:544-556); `as i32` casts wrap */
void orc_igd_count_set_overlaps(const orc_igd *g, const uint32_t *qc,
                                const uint32_t *qs, const uint32_t *qe, uint64_t nq,
                                int32_t min_overlap, uint64_t *hits, uint64_t n_files) {
    memset(hits, 0, n_files * sizeof(uint64_t));
    for (uint64_t q = 0; q < nq; q++)
        orc_igd_count_overlaps(g, qc[q], (int32_t)qs[q], (int32_t)qe[q], min_overlap, hits);
}

/* Igd::count_region_hits (igd.rs:563-590) */
void orc_igd_count_region_hits(const orc_igd *g, const uint32_t *qc,
                               const uint32_t *qs, const uint32_t *qe, uint64_t nq,
                               int32_t min_overlap, uint64_t *totals, uint64_t n_files) {
    memset(totals, 0, n_files * sizeof(uint64_t));
    uint64_t *per_region = (uint64_t *)xcalloc(n_files, sizeof(uint64_t));
    for (uint64_t q = 0; q < nq; q++) {
        memset(per_region, 0, n_files * sizeof(uint64_t));
        orc_igd_count_overlaps(g, qc[q], (int32_t)qs[q], (int32_t)qe[q], min_overlap,
                               per_region);
        for (uint64_t i = 0; i < n_files; i++)
            if (per_region[i] > 0) totals[i] += 1;
    }
    free(per_region);
}

/* insertion-ordered "HashSet<u32>" for one query: tiny open-addressing set */
typedef struct {
    uint32_t *keys;
    uint8_t *used;
    size_t cap, n;
} u32set;
static void set_init(u32set *s) {
    s->cap = 64;
    s->n = 0;
    s->keys = (uint32_t *)xmalloc(s->cap * sizeof(uint32_t));
    s->used = (uint8_t *)xcalloc(s->cap, 1);
}
static void set_clear(u32set *s) {
    memset(s->used, 0, s->cap);
    s->n = 0;
}
static int set_insert(u32set *s, uint32_t k);
static void set_grow(u32set *s) {
    u32set o = *s;
    s->cap = o.cap * 2;
    s->n = 0;
    s->keys = (uint32_t *)xmalloc(s->cap * sizeof(uint32_t));
    s->used = (uint8_t *)xcalloc(s->cap, 1);
    for (size_t i = 0; i < o.cap; i++)
        if (o.used[i]) set_insert(s, o.keys[i]);
    free(o.keys);
    free(o.used);
}
static int set_insert(u32set *s, uint32_t k) {
    if ((s->n + 1) * 2 > s->cap) set_grow(s);
    size_t h = ((size_t)k * 2654435761u) & (s->cap - 1);
    while (s->used[h]) {
        if (s->keys[h] == k) return 0;
        h = (h + 1) & (s->cap - 1);
    }
    s->used[h] = 1;
    s->keys[h] = k;
    s->n++;
    return 1;
}
static void set_free(u32set *s) {
    free(s->keys);
    free(s->used);
}

typedef struct {
    u32set *seen;
    uint32_t q;
    uint32_t *out_q, *out_s;
    uint64_t cap, n;
} pair_ctx;
static void on_pair(const rec_t *rec, void *p) {
    pair_ctx *c = (pair_ctx *)p;
    if (set_insert(c->seen, (uint32_t)rec->value)) {
        if (c->n < c->cap) {
            if (c->out_q) c->out_q[c->n] = c->q;
            if (c->out_s) c->out_s[c->n] = (uint32_t)rec->value;
        }
        c->n++;
    }
}

/* Igd::find_overlaps_regionset (igd.rs:645-678).  Note: no query validation
 * and no clamping here, unlike count_overlaps. */
uint64_t orc_igd_find_overlaps_regionset(const orc_igd *g, const uint32_t *qc,
                                         const uint32_t *qs, const uint32_t *qe,
                                         uint64_t nq, int32_t min_overlap,
                                         uint32_t *out_q, uint32_t *out_s,
                                         uint64_t cap) {
    if (!g->finalized) abort();
    u32set seen;
    set_init(&seen);
    pair_ctx c = {&seen, 0, out_q, out_s, cap, 0};
    for (uint64_t q = 0; q < nq; q++) {
        const contig_t *ctg = igd_contig(g, qc[q]);
        if (!ctg) continue;
        set_clear(&seen);
        c.q = (uint32_t)q;
        walk_tile_overlaps(ctg, (int32_t)qs[q], (int32_t)qe[q], min_overlap, g->nbp,
                           on_pair, &c);
    }
    set_free(&seen);
    return c.n;
}

typedef struct {
    u32set *seen;
    uint32_t *count;
} pq_ctx;
static void on_pq(const rec_t *rec, void *p) {
    pq_ctx *c = (pq_ctx *)p;
    if (set_insert(c->seen, (uint32_t)rec->value)) *c->count += 1;
}

/* Igd::count_overlaps_per_query (igd.rs:690-722) */
void orc_igd_count_overlaps_per_query(const orc_igd *g, const uint32_t *qc,
                                      const uint32_t *qs, const uint32_t *qe,
                                      uint64_t nq, int32_t min_overlap,
                                      uint32_t *counts) {
    if (!g->finalized) abort();
    u32set seen;
    set_init(&seen);
    for (uint64_t q = 0; q < nq; q++) {
        counts[q] = 0;
        const contig_t *ctg = igd_contig(g, qc[q]);
        if (!ctg) continue;
        set_clear(&seen);
        pq_ctx c = {&seen, &counts[q]};
        walk_tile_overlaps(ctg, (int32_t)qs[q], (int32_t)qe[q], min_overlap, g->nbp,
                           on_pq, &c);
    }
    set_free(&seen);
}

/* run_lola table step (gtars-lola/src/enrichment.rs:214-220) */
void orc_lola_contingency(const uint64_t *user_hits, const uint64_t *universe_hits,
                          uint64_t n_files, int64_t user_size, int64_t universe_size,
                          int64_t *a, int64_t *b, int64_t *c, int64_t *d) {
    for (uint64_t f = 0; f < n_files; f++) {
        int64_t av = (int64_t)user_hits[f];
        int64_t bv = (int64_t)universe_hits[f] - av;
        int64_t cv = user_size - av;
        int64_t dv = universe_size - av - bv - cv;
        a[f] = av;
        b[f] = bv;
        c[f] = cv;
        d[f] = dv;
    }
}
