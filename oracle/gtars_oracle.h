/*
 * gtars_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C, single-threaded CPU restatement of the reference (databio/gtars,
 * Rust) algorithms on the interval-overlap / tokenization / IGD hot path.
 * It is the checker the GPU path is compared against; it is never the thing
 * shipped or measured as the product.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.
 *
 * Parity pinning: the reference cannot be compiled here (Rust-only, no
 * cargo), so this oracle is pinned by the reference's own known-answer
 * tests and fixture files (tests/test_oracle_golden.py, tests/golden/).
 *
 * Every function cites the reference file:line it restates
 * (paths relative to the reference checkout).
 */
#ifndef GTARS_ORACLE_H
#define GTARS_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* kind of single-chromosome overlap index (gtars-overlaprs/src/lib.rs:139-144) */
#define ORC_KIND_BITS 0
#define ORC_KIND_AILIST 1

typedef struct orc_index orc_index;

/* Per-chromosome bucketing + Bits::build / AIList::build
 * (gtars-tokenizers/src/utils/mod.rs:49-99,
 *  gtars-overlaprs/src/multi_chrom_overlapper.rs:325-351,
 *  bits.rs:101-128, ailist.rs:105-151, 198-236).
 * chrom[i] is a dense chromosome id < n_chrom (the string->id dictionary is
 * the caller's business); chromosomes without any interval do not exist in
 * the index (HashMap semantics). */
orc_index *orc_index_build(const uint32_t *chrom, const uint32_t *start,
                           const uint32_t *end, const uint32_t *val,
                           uint64_t n, uint32_t n_chrom, int kind);
void orc_index_free(orc_index *ix);

uint64_t orc_index_chrom_len(const orc_index *ix, uint32_t chrom);
uint32_t orc_index_max_len(const orc_index *ix, uint32_t chrom);   /* Bits.max_len */
uint64_t orc_index_n_headers(const orc_index *ix, uint32_t chrom); /* AIList.header_list.len() */
void orc_index_headers(const orc_index *ix, uint32_t chrom, uint64_t *out);
/* stored (iteration) order of one chromosome: Overlapper::iter() */
void orc_index_stored(const orc_index *ix, uint32_t chrom, uint32_t *start,
                      uint32_t *end, uint32_t *val);

/* Overlapper::find (bits.rs:141-156,433-446 / ailist.rs:153-178,238-263).
 * Returns the number of hits; writes at most cap of them (in reference
 * result order). Any out pointer may be NULL. */
uint64_t orc_find(const orc_index *ix, uint32_t chrom, uint32_t qs, uint32_t qe,
                  uint32_t *out_start, uint32_t *out_end, uint32_t *out_val,
                  uint64_t cap);
/* Bits::count (bits.rs:337-344) -- the two-binary-search identity */
uint64_t orc_bits_count(const orc_index *ix, uint32_t chrom, uint32_t qs,
                        uint32_t qe);

/* Tokenizer::tokenize / encode inner loop minus strings
 * (gtars-tokenizers/src/tokenizer.rs:141-156): concatenation over queries (in
 * input order) of hit vals (index order); unknown chromosomes skipped.
 * offsets has nq+1 entries.  Returns total hits H; ids written up to cap.
 * The batch-level "[unk] if empty" rule (tokenizer.rs:158-160) is applied by
 * the caller. */
uint64_t orc_tokenize(const orc_index *ix, const uint32_t *qc,
                      const uint32_t *qs, const uint32_t *qe, uint64_t nq,
                      uint64_t *offsets, uint32_t *ids, uint64_t cap);

/* MultiChromOverlapper::count_overlaps / any_overlaps / find_overlaps_regions
 * (multi_chrom_overlapper.rs:483-550, overlap_bp :561-563).  has_min=0 means
 * min_overlap=None. */
void orc_count_overlaps(const orc_index *ix, const uint32_t *qc,
                        const uint32_t *qs, const uint32_t *qe, uint64_t nq,
                        int has_min, int32_t min_overlap, uint64_t *counts);
void orc_any_overlaps(const orc_index *ix, const uint32_t *qc,
                      const uint32_t *qs, const uint32_t *qe, uint64_t nq,
                      int has_min, int32_t min_overlap, uint8_t *out);
uint64_t orc_find_overlaps_regions(const orc_index *ix, const uint32_t *qc,
                                   const uint32_t *qs, const uint32_t *qe,
                                   uint64_t nq, int has_min,
                                   int32_t min_overlap, uint64_t *offsets,
                                   uint32_t *out_start, uint32_t *out_end,
                                   uint32_t *out_val, uint64_t cap);

/* IndexedRegionSet::find_overlaps (indexed_region_set.rs:145-152, 246-263):
 * per query the sorted, de-duplicated source indices of every source row that
 * shares coordinates with a hit.  The index must have been built with
 * val[i] = i over the same (chrom,start,end) arrays passed here as src_*. */
uint64_t orc_irs_find_overlaps(const orc_index *ix, const uint32_t *src_chrom,
                               const uint32_t *src_start,
                               const uint32_t *src_end, uint64_t n_src,
                               const uint32_t *qc, const uint32_t *qs,
                               const uint32_t *qe, uint64_t nq, int has_min,
                               int32_t min_overlap, uint64_t *offsets,
                               uint64_t *out_idx, uint64_t cap);

/* ------------------------------------------------------------------ IGD */
typedef struct orc_igd orc_igd;

orc_igd *orc_igd_new(int32_t nbp); /* igd.rs:80-95; nbp<=0 -> 16384 */
void orc_igd_free(orc_igd *g);
/* Igd::add (igd.rs:109-153) */
void orc_igd_add(orc_igd *g, uint32_t chrom, int32_t start, int32_t end,
                 int32_t value, uint32_t file_idx);
/* n x orc_igd_add, in array order */
void orc_igd_add_arrays(orc_igd *g, const uint32_t *chrom, const int32_t *start,
                        const int32_t *end, const int32_t *value,
                        const uint32_t *file_idx, uint64_t n);
/* Igd::finalize (igd.rs:157-167) */
void orc_igd_finalize(orc_igd *g);
uint64_t orc_igd_total_records(const orc_igd *g); /* igd.rs:736-742 */
uint64_t orc_igd_num_contigs(const orc_igd *g);
/* Igd::count_overlaps (igd.rs:504-540) + walk_tile_overlaps (igd.rs:753-847) */
uint32_t orc_igd_count_overlaps(const orc_igd *g, uint32_t chrom, int32_t start,
                                int32_t end, int32_t min_overlap,
                                uint64_t *hits);
/* Igd::count_set_overlaps (igd.rs:544-556): hits[n_files] is zeroed here */
void orc_igd_count_set_overlaps(const orc_igd *g, const uint32_t *qc,
                                const uint32_t *qs, const uint32_t *qe,
                                uint64_t nq, int32_t min_overlap,
                                uint64_t *hits, uint64_t n_files);
/* Igd::count_region_hits (igd.rs:563-590) */
void orc_igd_count_region_hits(const orc_igd *g, const uint32_t *qc,
                               const uint32_t *qs, const uint32_t *qe,
                               uint64_t nq, int32_t min_overlap,
                               uint64_t *totals, uint64_t n_files);
/* Igd::find_overlaps_regionset (igd.rs:645-678); returns #pairs */
uint64_t orc_igd_find_overlaps_regionset(const orc_igd *g, const uint32_t *qc,
                                         const uint32_t *qs,
                                         const uint32_t *qe, uint64_t nq,
                                         int32_t min_overlap, uint32_t *out_q,
                                         uint32_t *out_s, uint64_t cap);
/* Igd::count_overlaps_per_query (igd.rs:690-722) */
void orc_igd_count_overlaps_per_query(const orc_igd *g, const uint32_t *qc,
                                      const uint32_t *qs, const uint32_t *qe,
                                      uint64_t nq, int32_t min_overlap,
                                      uint32_t *counts);

/* LOLA contingency table (gtars-lola/src/enrichment.rs:214-220) */
void orc_lola_contingency(const uint64_t *user_hits,
                          const uint64_t *universe_hits, uint64_t n_files,
                          int64_t user_size, int64_t universe_size,
                          int64_t *a, int64_t *b, int64_t *c, int64_t *d);

/* synthetic-data PRNG shared by generators */
uint64_t orc_splitmix64(uint64_t *state);

/* fragsplit_oracle.c: pseudobulk_fragment_files (gtars-fragsplit/src/split.rs:36-151) feeding tokenize_fragment_file
 * (gtars-tokenizers/src/utils/fragments.rs:61-82), one thread, clusters kept in memory.  map_keys[i] ("{stem}+{barcode}") ->
 * map_cluster[i] < n_clusters; chrom_names[c] is the name of the index's chromosome id c.  Per cluster: ids, sum of ids,
 * distinct barcodes.  Returns the number of fragment lines read, (uint64_t)-1 on an unreadable file or a malformed line. */
uint64_t orc_fragsplit_tokenize(const orc_index *ix, const char *const *files, uint64_t n_files, const char *const *map_keys,
                                const uint32_t *map_cluster, uint64_t n_map, uint32_t n_clusters, const char *const *chrom_names,
                                uint32_t n_chrom, uint32_t unk_id, uint64_t *out_ids, uint64_t *out_sum, uint64_t *out_barcodes);

#ifdef __cplusplus
}
#endif
#endif
