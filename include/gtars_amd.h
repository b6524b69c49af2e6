/*
 * gtars_amd.h -- C ABI of libgtars_amd.so, the MI355X (gfx950) engine for the
 * gtars interval-overlap / region-set tokenization hot path.
 *
 * The reference (databio/gtars) has no C ABI: its seams are Rust traits and
 * structs called from pyo3.  Each entry point below names the reference
 * interface it replaces (file:line, relative to the reference checkout); a
 * Rust `-sys` crate, ctypes or cgo can bind these directly
 * (see INTEGRATION.md).
 *
 * Conventions
 *  - every function returns a gtars_status (0 = OK); gtars_last_error() gives
 *    a thread-local message for the last failure on the calling thread;
 *  - chromosomes are dense u32 ids assigned by the caller's string dictionary
 *    (the tokenizer/regionset layer in this library does that for BED input);
 *    ids >= n_chrom, or chromosomes with no indexed interval, are "unknown
 *    chromosome" and yield no hits (tokenizer.rs:143-150,
 *    multi_chrom_overlapper.rs:231-234);
 *  - `*_device` entry points take DEVICE pointers and enqueue on `stream`
 *    (a hipStream_t passed as void*; NULL = the null stream) and only
 *    synchronise where the signature returns a host value; the plain entry
 *    points take HOST pointers and do the transfers themselves;
 *  - handles are immutable after build and may be queried concurrently from
 *    several host threads and on several streams: scratch memory is kept per
 *    (host thread, stream) and grows on demand -- the first call on a stream
 *    may allocate, so issue it once before capturing a HIP graph;
 *  - memory returned through `T** out` is freed with gtars_free().
 *
 * There is no CPU fallback: without a HIP device every compute entry point
 * fails with GTARS_ERR_NO_DEVICE.
 */
#ifndef GTARS_AMD_H
#define GTARS_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum gtars_status {
    GTARS_OK = 0,
    GTARS_ERR_INVALID_ARG = 1,
    GTARS_ERR_NO_DEVICE = 2,
    GTARS_ERR_HIP = 3,        /* a HIP runtime call failed */
    GTARS_ERR_CAPACITY = 4,   /* caller-provided output buffer too small */
    GTARS_ERR_IO = 5,         /* file missing / unreadable */
    GTARS_ERR_PARSE = 6,      /* malformed BED / TOML / .gtok / .igd */
    GTARS_ERR_EMPTY = 7,      /* empty region set (RegionSetError::EmptyRegionSet) */
    GTARS_ERR_CONFIG = 8,     /* bad extension / bad tokenizer_type */
    GTARS_ERR_INTERNAL = 9
} gtars_status;

/* OverlapperType (gtars-overlaprs/src/lib.rs:139-144) */
#define GTARS_KIND_BITS 0
#define GTARS_KIND_AILIST 1

#define GTARS_UNKNOWN_CHROM 0xFFFFFFFFu

const char *gtars_last_error(void);
const char *gtars_version(void);
/* number of HIP devices visible (0 when there is none; never fails) */
int gtars_device_count(void);
void gtars_free(void *p);

/* ------------------------------------------------------------------------
 * Overlap index: per-chromosome Bits or AIList, SoA in HBM.
 * Replaces Overlapper::build for a whole genome:
 *   Bits::build        gtars-overlaprs/src/bits.rs:101-128
 *   AIList::build      gtars-overlaprs/src/ailist.rs:105-151, 198-236
 *   per-chrom buckets  gtars-tokenizers/src/utils/mod.rs:49-99,
 *                      gtars-overlaprs/src/multi_chrom_overlapper.rs:325-351
 * ---------------------------------------------------------------------- */
typedef struct gtars_index gtars_index_t;

/* Host arrays of n intervals. val may be NULL (val[i] = i).  Intervals with
 * chrom[i] >= n_chrom are rejected (GTARS_ERR_INVALID_ARG). */
gtars_status gtars_index_build(const uint32_t *chrom, const uint32_t *start,
                               const uint32_t *end, const uint32_t *val,
                               uint64_t n, uint32_t n_chrom, int kind,
                               gtars_index_t **out);
void gtars_index_free(gtars_index_t *ix);

uint64_t gtars_index_len(const gtars_index_t *ix);
uint32_t gtars_index_n_chrom(const gtars_index_t *ix);
int gtars_index_kind(const gtars_index_t *ix);
/* The HIP device the handle's memory lives on: the device that was current when it was built (-1 for NULL).  Host-buffer
 * entry points run there whatever the calling thread's current device is and put the caller's device back; `*_device` entry
 * points take the caller's device pointers and stream, so they require the calling thread's current device to BE the
 * handle's and return GTARS_ERR_INVALID_ARG ("handle lives on device k, current device is j") otherwise.  One process per
 * GPU (the multi-GPU layout of this library) never sees the difference; a process that drives several GPUs does. */
int gtars_index_device(const gtars_index_t *ix);
/* Overlapper::iter(): stored order of one chromosome copied back to the host
 * (Bits: sorted (start,end); AIList: sub-list major).  Returns its length;
 * out pointers may be NULL. */
uint64_t gtars_index_chrom_len(const gtars_index_t *ix, uint32_t chrom);
gtars_status gtars_index_stored(const gtars_index_t *ix, uint32_t chrom,
                                uint32_t *start, uint32_t *end, uint32_t *val);
/* Bits::insert (gtars-overlaprs/src/bits.rs:209-222).  The device structures are immutable: *out is a NEW index =
 * `ix` plus the interval, stored where bsearch_seq_ref (bits.rs:304-322) puts it, i.e. in front of equal (start, end)
 * keys, with max_len updated; the caller swaps handles and frees the old one.  O(n), like the reference's Vec::insert. */
gtars_status gtars_index_insert(const gtars_index_t *ix, uint32_t chrom, uint32_t start, uint32_t end, uint32_t val,
                                gtars_index_t **out);
/* Bits::seek (bits.rs:364-386) + IterFind (bits.rs:433-446) on the host copy of one chromosome: *cursor is the
 * caller's cursor into the chromosome's stored order (0 to begin), updated by the reference's rule; the vals of the
 * hits from the cursor on are written to out_vals (may be NULL: count only).  More hits than `capacity`:
 * GTARS_ERR_CAPACITY with *n_hits = the number needed.  Sequential API for completeness -- a batch of queries
 * belongs in gtars_tokenize. */
gtars_status gtars_index_seek(const gtars_index_t *ix, uint32_t chrom, uint32_t start, uint32_t stop, uint64_t *cursor,
                              uint32_t *out_vals, uint64_t capacity, uint64_t *n_hits);
/* Bits.max_len (bits.rs:110-119) / AIList.header_list (ailist.rs:127-141) */
uint32_t gtars_index_max_len(const gtars_index_t *ix, uint32_t chrom);
uint64_t gtars_index_n_sublists(const gtars_index_t *ix, uint32_t chrom);
gtars_status gtars_index_sublist_offsets(const gtars_index_t *ix, uint32_t chrom,
                                         uint64_t *out);

/* ------------------------------------------------------------------------
 * Tokenize / enumerate: for every query, the vals of all indexed intervals
 * with  iv.start < q_end && iv.end > q_start  (interval.rs:47-50), in the
 * index's result order, concatenated in query order.
 * Replaces the inner loop of Tokenizer::tokenize / encode
 *   gtars-tokenizers/src/tokenizer.rs:140-171
 * i.e. Overlapper::find  bits.rs:141-156,433-446 / ailist.rs:153-178,238-263.
 * The batch-level "[unk] when nothing overlapped" rule (tokenizer.rs:158-160)
 * is applied by gtars_tokenizer_* below, not here.
 * ---------------------------------------------------------------------- */

/* Single pass, device pointers.  d_offsets: nq+1 u64 (CSR); d_ids: room for
 * ids_capacity u32.  *total_hits (host) receives H after the stream has been
 * synchronised.  If H > ids_capacity the offsets are complete and valid, ids
 * beyond the capacity are not written and GTARS_ERR_CAPACITY is returned
 * (re-run with a larger buffer, or call gtars_fill_device).  total_hits may
 * be NULL: then nothing is synchronised and overflow is not reported. */
gtars_status gtars_tokenize_device(const gtars_index_t *ix, const uint32_t *d_qchrom,
                                   const uint32_t *d_qstart, const uint32_t *d_qend,
                                   uint64_t nq, uint64_t *d_offsets, uint32_t *d_ids,
                                   uint64_t ids_capacity, uint64_t *total_hits,
                                   void *stream);

/* The same call with a HINT about the batch.  The fused tokenizer exists in two builds that give identical offsets and ids: one
 * with the run form of hit-heavy queries (tens to hundreds of ids per query leave by wave-wide stores: 4x faster on such batches)
 * and one without it (2-3.5 % faster on batches of ~1 id per query, BASELINE config 2).  GTARS_TOK_AUTO picks by the caller's
 * id capacity -- a launch whose buffer cannot hold 4 ids per query cannot be a hit-heavy batch that completes; an offsets-only
 * launch (the sizing pass of a two-pass caller) keeps the run form -- which makes a caller that over-allocates its id buffer pay
 * the 2-3.5 %.  A caller that knows its batch says so: GTARS_TOK_NARROW (about one id per query), GTARS_TOK_WIDE (many). */
#define GTARS_TOK_AUTO 0
#define GTARS_TOK_NARROW 1
#define GTARS_TOK_WIDE 2
/* ... OR-ed with GTARS_TOK_SORTED: the batch is in (chromosome id, start) order -- what Tokenizer::tokenize receives from a
 * file-loaded RegionSet (gtars-core/src/models/region_set.rs:182, 502-505) -- and the launch runs the SWEEP form of the tokenizer
 * (k_tok_sweep: every wave stages the contiguous slice of the blocked records its 256 consecutive queries need in LDS; no search
 * image, any universe size).  Same offsets and ids for ANY batch.  EXPERIMENTAL, opt-in, and measured SLOWER than the default
 * kernel, which on a batch in order already runs at its best rate because neighbouring lanes' record requests coalesce: 18.9 vs
 * 14.4 us per 1M queries, 568 vs 411 us per 64M (profiles/r06/sweep_*.txt).  Nothing selects it by itself. */
#define GTARS_TOK_SORTED 4
gtars_status gtars_tokenize_device_ex(const gtars_index_t *ix, const uint32_t *d_qchrom,
                                      const uint32_t *d_qstart, const uint32_t *d_qend,
                                      uint64_t nq, uint64_t *d_offsets, uint32_t *d_ids,
                                      uint64_t ids_capacity, uint64_t *total_hits,
                                      void *stream, int hint);

/* Second pass of a two-pass caller: ids for existing offsets.  Contract: d_offsets are the offsets a gtars_tokenize_device call
 * on the SAME index and batch left (offsets[0] == 0), d_ids holds offsets[nq] ids.  gtars_fill_device_n is the form for a caller
 * that read offsets[nq] back to size d_ids (every two-pass caller has): total_hits bounds what is written (a buffer that turns
 * out short ends in GTARS_ERR_CAPACITY at the next synchronising call instead of an out-of-bounds write) and picks the build
 * (total_hits < 4 nq: the narrow one). */
gtars_status gtars_fill_device_n(const gtars_index_t *ix, const uint32_t *d_qchrom,
                                 const uint32_t *d_qstart, const uint32_t *d_qend, uint64_t nq,
                                 const uint64_t *d_offsets, uint32_t *d_ids, uint64_t total_hits, void *stream);
gtars_status gtars_fill_device(const gtars_index_t *ix, const uint32_t *d_qchrom,
                               const uint32_t *d_qstart, const uint32_t *d_qend,
                               uint64_t nq, const uint64_t *d_offsets, uint32_t *d_ids,
                               void *stream);

/* Host pointers; *out_ids is library-allocated (gtars_free). */
gtars_status gtars_tokenize(const gtars_index_t *ix, const uint32_t *qchrom,
                            const uint32_t *qstart, const uint32_t *qend, uint64_t nq,
                            uint64_t *offsets, uint32_t **out_ids, uint64_t *out_n);

/* Host pointers, caller-provided outputs: offsets[nq + 1], ids[ids_capacity].  The streaming form of
 * gtars_tokenize: device buffers and streams are cached per calling thread, the batch is fed in chunks
 * (host-to-device copy of chunk k+1 || kernel of chunk k || device-to-host copy of chunk k-1), nothing is
 * allocated on the way.  *out_n = number of ids; GTARS_ERR_CAPACITY (with *out_n set, offsets complete)
 * when ids_capacity is too small.  Reuse the output buffers across calls: freshly mapped pages cost more
 * than the transfer. */
gtars_status gtars_tokenize_into(const gtars_index_t *ix, const uint32_t *qchrom,
                                 const uint32_t *qstart, const uint32_t *qend, uint64_t nq,
                                 uint64_t *offsets, uint32_t *ids, uint64_t ids_capacity,
                                 uint64_t *out_n);

/* bins[id] += 1 for every id < n_bins (device pointers): the scatter-add of gtars-scoring's count matrices
 * (CountMatrix::increment, gtars-scoring/src/fragment_scoring.rs:88-105) -- one matrix row per call, the ids being
 * the token ids of one fragment file's probes (gtars_tokenize_device). */
gtars_status gtars_histogram_u32_device(const uint32_t *d_ids, uint64_t n, uint32_t n_bins, uint32_t *d_bins,
                                        void *stream);

/* mat[(row[q] - row0) * n_cols + id] += 1 for every token id of query q (the CSR of a gtars_tokenize_device call) whose row lies
 * in [row0, row0 + n_rows): the scatter-add of barcode_scoring_from_fragments (gtars-scoring/src/fragment_scoring.rs:125-155: one
 * count per (barcode, overlapped peak)) into a device-resident band of the barcode x peak matrix; d_row[q] = barcode id of
 * fragment q.  Device pointers; d_mat is not cleared. */
gtars_status gtars_histogram_rows_device(const uint64_t *d_offsets, const uint32_t *d_ids, const uint32_t *d_row, uint64_t nq,
                                         uint32_t row0, uint32_t n_rows, uint32_t n_cols, uint32_t *d_mat, void *stream);

/* ------------------------------------------------------------------------
 * Counts / any / find with the optional min-overlap filter.
 * Replaces MultiChromOverlapper::count_overlaps / any_overlaps /
 * find_overlaps_regions (multi_chrom_overlapper.rs:483-550) and
 * IndexedRegionSet::count/any/find_overlaps (indexed_region_set.rs:234-263).
 * has_min = 0 is `None`; the filter  overlap_bp >= min_overlap  is applied
 * only when min_overlap > 1 (multi_chrom_overlapper.rs:491).
 * ---------------------------------------------------------------------- */
gtars_status gtars_count_overlaps_device(const gtars_index_t *ix, const uint32_t *d_qchrom,
                                         const uint32_t *d_qstart, const uint32_t *d_qend,
                                         uint64_t nq, int has_min, int32_t min_overlap,
                                         uint32_t *d_counts, void *stream);
gtars_status gtars_count_overlaps(const gtars_index_t *ix, const uint32_t *qchrom,
                                  const uint32_t *qstart, const uint32_t *qend,
                                  uint64_t nq, int has_min, int32_t min_overlap,
                                  uint32_t *counts);
/* Bits::count (gtars-overlaprs/src/bits.rs:337-344, bsearch_seq :304-322): per query
 * len - #{ends < start+1} - #{starts >= stop} on the chromosome's separately sorted starts / ends,
 * with the reference's wrapping arithmetic (equals find().len() except for zero-length / inverted
 * queries).  Bits-kind indexes only; unknown chromosome -> 0.  Counts are u64 (Rust usize). */
gtars_status gtars_bits_count_device(const gtars_index_t *ix, const uint32_t *d_qchrom,
                                     const uint32_t *d_qstart, const uint32_t *d_qend, uint64_t nq,
                                     uint64_t *d_counts, void *stream);
gtars_status gtars_bits_count(const gtars_index_t *ix, const uint32_t *qchrom, const uint32_t *qstart,
                              const uint32_t *qend, uint64_t nq, uint64_t *counts);
gtars_status gtars_any_overlaps(const gtars_index_t *ix, const uint32_t *qchrom,
                                const uint32_t *qstart, const uint32_t *qend, uint64_t nq,
                                int has_min, int32_t min_overlap, uint8_t *out);
/* CSR of hits; any of out_start/out_end/out_val may be NULL. */
gtars_status gtars_find_overlaps(const gtars_index_t *ix, const uint32_t *qchrom,
                                 const uint32_t *qstart, const uint32_t *qend, uint64_t nq,
                                 int has_min, int32_t min_overlap, uint64_t *offsets,
                                 uint32_t **out_start, uint32_t **out_end,
                                 uint32_t **out_val, uint64_t *out_n);
/* IndexedRegionSet::find_overlaps (indexed_region_set.rs:246-263): per query
 * the sorted, de-duplicated source indices.  The index must have been built
 * with val == NULL (val[i] = i). */
gtars_status gtars_find_overlap_indices(const gtars_index_t *ix, const uint32_t *qchrom,
                                        const uint32_t *qstart, const uint32_t *qend,
                                        uint64_t nq, int has_min, int32_t min_overlap,
                                        uint64_t *offsets, uint32_t **out_idx,
                                        uint64_t *out_n);

/* Index-side subset (the index's own intervals that are hit by ANY query):
 *   MultiChromOverlapper::subset_by / subset_by_overlaps / intersect_all
 *       gtars-overlaprs/src/multi_chrom_overlapper.rs:449-478, 554-556
 *   IndexedRegionSet::intersect_all / subset_by_overlaps
 *       gtars-overlaprs/src/indexed_region_set.rs:201-230
 * One pass over the batch marks the hit stored positions in a bitmap (no hit list is materialised), the bitmap is
 * compacted.  min_overlap filters only when has_min && min_overlap > 1, as in the reference.
 *
 * gtars_subset_by_overlaps: the de-duplicated (chrom id, start, end) triples, sorted by (chrom id, start, end) -- the
 * reference's BTreeSet<(String, u32, u32)> orders chromosomes by NAME: a caller whose ids are not in name order
 * permutes the chromosome runs (the Python layer does).
 * gtars_subset_source_indices: the vals of the hit intervals, ascending and unique -- for an index built with
 * val == NULL these are the source rows, i.e. IndexedRegionSet's result in source order. */
gtars_status gtars_subset_by_overlaps(const gtars_index_t *ix, const uint32_t *qchrom, const uint32_t *qstart,
                                      const uint32_t *qend, uint64_t nq, int has_min, int32_t min_overlap,
                                      uint32_t **out_chrom, uint32_t **out_start, uint32_t **out_end,
                                      uint64_t *out_n);
gtars_status gtars_subset_source_indices(const gtars_index_t *ix, const uint32_t *qchrom, const uint32_t *qstart,
                                         const uint32_t *qend, uint64_t nq, int has_min, int32_t min_overlap,
                                         uint32_t **out_idx, uint64_t *out_n);
/* Device form of the marking pass: bit p of d_mark (ceil(gtars_index_len / 32) u32 words, zeroed by this call on
 * `stream`) is set iff the interval at STORED position p (gtars_index_stored order, chromosomes concatenated) is hit
 * by some query.  Asynchronous.  GTARS_ERR_INVALID_ARG for an index without the blocked structure (nested AIList
 * sub-lists, > 4M blocks): use the host-pointer forms above, which fall back to the generic kernels. */
gtars_status gtars_mark_overlapped_device(const gtars_index_t *ix, const uint32_t *d_qchrom, const uint32_t *d_qstart,
                                          const uint32_t *d_qend, uint64_t nq, int has_min, int32_t min_overlap,
                                          uint32_t *d_mark, void *stream);

/* ------------------------------------------------------------------------
 * IGD: multi-file interval database, per-file hit counting.
 * Replaces gtars-igd/src/igd.rs: Igd::add (:109-153), finalize (:157-167),
 * count_set_overlaps (:544-556), count_region_hits (:563-590),
 * from_single_region_set (:609-634), find_overlaps_regionset (:645-678),
 * count_overlaps_per_query (:690-722); walk_tile_overlaps (:753-847).
 * Defined for min_overlap >= 1 (GTARS_ERR_INVALID_ARG otherwise).
 * ---------------------------------------------------------------------- */
typedef struct gtars_igd gtars_igd_t;

/* Host arrays of n records; the same drop rules as Igd::add apply
 * (start<0 || end<0 || start>=end are skipped). value may be NULL (0). */
gtars_status gtars_igd_build(const uint32_t *chrom, const int32_t *start,
                             const int32_t *end, const int32_t *value,
                             const uint32_t *file_idx, uint64_t n, uint32_t n_chrom,
                             uint32_t n_files, gtars_igd_t **out);
void gtars_igd_free(gtars_igd_t *g);
uint64_t gtars_igd_len(const gtars_igd_t *g);          /* stored intervals */
uint32_t gtars_igd_n_files(const gtars_igd_t *g);
int gtars_igd_device(const gtars_igd_t *g); /* see gtars_index_device */
/* what Igd::total_records() would report for nbp (tile replicas counted) */
uint64_t gtars_igd_total_records(const gtars_igd_t *g, int32_t nbp);
/* copy the stored records back to the host, in device order = (chrom, start, insertion order), which
 * restricted to one nbp-tile is exactly the order of that tile in the reference (igd.rs:157-167);
 * used to write .igd files (igd.rs:425-486).  Each array has gtars_igd_len() entries; any may be NULL. */
gtars_status gtars_igd_export(const gtars_igd_t *g, uint32_t *chrom, int32_t *start, int32_t *end,
                              int32_t *value, uint32_t *file_idx);

/* binary = 0: Igd::count_set_overlaps (pairwise);
 * binary = 1: Igd::count_region_hits (at most 1 per query per file).
 * Query coordinates are u32 cast to i32 as the reference does (igd.rs:549-550).
 * d_hits: n_files u64, overwritten.  Asynchronous on `stream`: the call only enqueues work (whether the batch has
 * to be partitioned first is decided on the device); the first binary count of a database builds its per-record
 * "largest earlier end of the same file" column once, synchronously. */
gtars_status gtars_igd_count_device(const gtars_igd_t *g, const uint32_t *d_qchrom,
                                    const uint32_t *d_qstart, const uint32_t *d_qend,
                                    uint64_t nq, int32_t min_overlap, int binary,
                                    uint64_t *d_hits, void *stream);
gtars_status gtars_igd_count(const gtars_igd_t *g, const uint32_t *qchrom,
                             const uint32_t *qstart, const uint32_t *qend, uint64_t nq,
                             int32_t min_overlap, int binary, uint64_t *hits);
/* n_sets query sets against one database in one call: the concatenated batch, set k = rows
 * [set_off[k], set_off[k + 1]) (set_off: HOST array of n_sets + 1 offsets, set_off[0] = 0); d_hits / hits: u64[n_sets][n_files],
 * row k = what gtars_igd_count(_device) returns for set k alone.  This is the count step of run_lola
 * (gtars-lola/src/enrichment.rs:198-221: universe_hits = igd.count_region_hits(universe), then user_hits per user set):
 * the reference walks the database once per set, here up to 4 sets share ONE pass over it (the partition tags each query
 * with its set, the sweep keeps a row of counters per set); what cannot share a pass is counted set by set. */
gtars_status gtars_igd_count_sets_device(const gtars_igd_t *g, const uint32_t *d_qchrom, const uint32_t *d_qstart,
                                         const uint32_t *d_qend, const uint64_t *set_off, uint32_t n_sets,
                                         int32_t min_overlap, int binary, uint64_t *d_hits, void *stream);
gtars_status gtars_igd_count_sets(const gtars_igd_t *g, const uint32_t *qchrom, const uint32_t *qstart,
                                  const uint32_t *qend, const uint64_t *set_off, uint32_t n_sets,
                                  int32_t min_overlap, int binary, uint64_t *hits);
/* Igd::count_overlaps_per_query (distinct `value`s per query) */
gtars_status gtars_igd_count_per_query(const gtars_igd_t *g, const uint32_t *qchrom,
                                       const uint32_t *qstart, const uint32_t *qend,
                                       uint64_t nq, int32_t min_overlap, uint32_t *counts);
/* Igd::find_overlaps_regionset: (query, subject) pairs in reference walk order */
gtars_status gtars_igd_find_pairs(const gtars_igd_t *g, const uint32_t *qchrom,
                                  const uint32_t *qstart, const uint32_t *qend,
                                  uint64_t nq, int32_t min_overlap, uint32_t **out_q,
                                  uint32_t **out_s, uint64_t *out_n);

/* LOLA contingency cells from support vectors (gtars-lola/src/enrichment.rs:214-220);
 * device pointers, one thread per file. */
gtars_status gtars_lola_contingency_device(const uint64_t *d_user_hits,
                                           const uint64_t *d_universe_hits,
                                           uint64_t n_files, int64_t user_size,
                                           int64_t universe_size, int64_t *d_a,
                                           int64_t *d_b, int64_t *d_c, int64_t *d_d,
                                           void *stream);

/* ------------------------------------------------------------------------
 * Instrumentation used by bench.py: when enabled every kernel launch made by
 * this library on the calling thread is bracketed by HIP events on its own
 * stream; gtars_prof_read() synchronises and returns accumulated times.
 * ---------------------------------------------------------------------- */
void gtars_prof_enable(int on);
void gtars_prof_reset(void);
/* fills up to cap entries; returns number of distinct kernels seen */
int gtars_prof_read(const char **names, double *total_ms, uint64_t *launches, int cap);

#ifdef __cplusplus
}
#endif
#endif
