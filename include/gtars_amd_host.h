/*
 * gtars_amd_host.h -- C ABI of the host ("string world") layer of
 * libgtars_amd.so: BED / BED.gz parsing, RegionSet, Universe + Tokenizer,
 * fragment files, .gtok, IGD databases from BED files.  It sits on top of the
 * integer engine in gtars_amd.h and mirrors the reference's Rust types
 * (cited per entry point, file:line relative to the reference checkout), so a
 * binding (pyo3-style, ctypes, cgo, extendr) can expose the same classes.
 *
 * Strings are UTF-8, NUL terminated.  `const char*` results are borrowed from
 * the handle they were asked of and stay valid until that handle is freed.
 * Same status / error conventions as gtars_amd.h.
 */
#ifndef GTARS_AMD_HOST_H
#define GTARS_AMD_HOST_H

#include "gtars_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------
 * RegionSet  (gtars-core/src/models/region_set.rs:40-45)
 * ---------------------------------------------------------------------- */
typedef struct gtars_regionset gtars_regionset_t;

/* RegionSet::try_from(&Path) (region_set.rs:52-186): BED or BED.gz (by
 * extension, utils.rs:115-126); header / comment handling (:112-135); rest =
 * columns 4+ joined by tabs; EmptyRegionSet -> GTARS_ERR_EMPTY; the result is
 * stably sorted by (chr, start) (:182, :502-505). */
gtars_status gtars_regionset_from_bed(const char *path, gtars_regionset_t **out);
/* From<Vec<Region>> (region_set.rs:212-220): in-memory, NOT sorted. rest may
 * be NULL (all None) and individual entries may be NULL. */
gtars_status gtars_regionset_from_arrays(const char *const *chrs, const uint32_t *starts,
                                         const uint32_t *ends, const char *const *rest,
                                         uint64_t n, gtars_regionset_t **out);
void gtars_regionset_free(gtars_regionset_t *rs);
uint64_t gtars_regionset_len(const gtars_regionset_t *rs);
const char *gtars_regionset_header(const gtars_regionset_t *rs); /* NULL if none */
/* dictionary-encoded columns (ids index gtars_regionset_chrom_name) */
uint32_t gtars_regionset_n_chrom(const gtars_regionset_t *rs);
const char *gtars_regionset_chrom_name(const gtars_regionset_t *rs, uint32_t id);
const uint32_t *gtars_regionset_chrom_ids(const gtars_regionset_t *rs);
const uint32_t *gtars_regionset_starts(const gtars_regionset_t *rs);
const uint32_t *gtars_regionset_ends(const gtars_regionset_t *rs);
const char *gtars_regionset_rest(const gtars_regionset_t *rs, uint64_t i); /* NULL if None */
/* generate_region_to_id_map (gtars-core/src/utils.rs:202-214): dense ids in first-seen order over the whole Region
 * (chr, start, end, rest): what gtars-scoring's ConsensusSet stores as interval payload (files.rs:60-83).
 * *out_ids: gtars_regionset_len ids (gtars_free); *out_n_ids (may be NULL): number of distinct regions */
gtars_status gtars_regionset_dense_ids(const gtars_regionset_t *rs, uint32_t **out_ids, uint32_t *out_n_ids);

/* IndexedRegionSet::new(other) then count / any / find_overlaps(self)
 * (gtars-overlaprs/src/indexed_region_set.rs:111-113, 234-263) -- the
 * semantics of python RegionSet.count_overlaps / any_overlaps / find_overlaps
 * (gtars-python/src/models/region_set.rs:445-478): `self` queries, `other` is
 * indexed on the GPU.  kind: GTARS_KIND_AILIST is the reference default. */
gtars_status gtars_regionset_count_overlaps(const gtars_regionset_t *self,
                                            const gtars_regionset_t *other, int kind,
                                            int has_min, int32_t min_overlap, uint32_t *counts);
gtars_status gtars_regionset_any_overlaps(const gtars_regionset_t *self,
                                          const gtars_regionset_t *other, int kind,
                                          int has_min, int32_t min_overlap, uint8_t *out);
gtars_status gtars_regionset_find_overlaps(const gtars_regionset_t *self,
                                           const gtars_regionset_t *other, int kind,
                                           int has_min, int32_t min_overlap, uint64_t *offsets,
                                           uint32_t **out_idx, uint64_t *out_n);

/* ------------------------------------------------------------------------
 * Tokenizer  (gtars-tokenizers/src/tokenizer.rs:36-279, universe/mod.rs,
 * config.rs, utils/mod.rs:34-99, utils/special_tokens.rs)
 * ---------------------------------------------------------------------- */
typedef struct gtars_tokenizer gtars_tokenizer_t;

/* Tokenizer::from_auto / from_config / from_bed (tokenizer.rs:61-138) */
gtars_status gtars_tokenizer_from_auto(const char *path, gtars_tokenizer_t **out);
gtars_status gtars_tokenizer_from_config(const char *path, gtars_tokenizer_t **out);
gtars_status gtars_tokenizer_from_bed(const char *path, gtars_tokenizer_t **out);
void gtars_tokenizer_free(gtars_tokenizer_t *t);

uint64_t gtars_tokenizer_vocab_size(const gtars_tokenizer_t *t);       /* get_vocab_size */
int gtars_tokenizer_kind(const gtars_tokenizer_t *t);                  /* GTARS_KIND_* */
/* convert_id_to_token / convert_token_to_id (universe/mod.rs:64-80) */
const char *gtars_tokenizer_id_to_token(const gtars_tokenizer_t *t, uint32_t id); /* NULL: none */
int64_t gtars_tokenizer_token_to_id(const gtars_tokenizer_t *t, const char *token); /* -1: none */
/* get_vocab(): the i-th (token, id) pair of region_to_id, i < vocab_size */
const char *gtars_tokenizer_vocab_token(const gtars_tokenizer_t *t, uint64_t i, uint32_t *id);
/* special tokens in the order unk,pad,mask,cls,eos,bos,sep (special_tokens.rs:59-71) */
const char *gtars_tokenizer_special_token(const gtars_tokenizer_t *t, int which);
/* universe metadata (BED5+ universes): name / score of a region string, NULL / NaN if absent */
const char *gtars_tokenizer_region_name(const gtars_tokenizer_t *t, const char *region);
double gtars_tokenizer_region_score(const gtars_tokenizer_t *t, const char *region);

/* chromosome dictionary of the core (for array fast paths): -1 = unknown */
int64_t gtars_tokenizer_chrom_id(const gtars_tokenizer_t *t, const char *chr);
uint32_t gtars_tokenizer_n_chrom(const gtars_tokenizer_t *t);
const char *gtars_tokenizer_chrom_name(const gtars_tokenizer_t *t, uint32_t id);
/* borrowed engine handle for gtars_tokenize_device & friends */
const gtars_index_t *gtars_tokenizer_index(const gtars_tokenizer_t *t);

/* Tokenizer::encode (tokenizer.rs:165-171) of a region set: ids in reference
 * order, [unk] when nothing overlapped (tokenizer.rs:158-160). */
gtars_status gtars_tokenizer_encode_regionset(const gtars_tokenizer_t *t,
                                              const gtars_regionset_t *rs, uint32_t **out_ids,
                                              uint64_t *out_n);
/* same on parallel arrays with chromosome NAMES */
gtars_status gtars_tokenizer_encode_arrays(const gtars_tokenizer_t *t, const char *const *chrs,
                                           const uint32_t *starts, const uint32_t *ends,
                                           uint64_t n, uint32_t **out_ids, uint64_t *out_n);
/* additive array fast path: chromosome IDS of this tokenizer's dictionary
 * (GTARS_UNKNOWN_CHROM for unknown); returns the CSR too (offsets n+1) and
 * does NOT apply the batch-level unk rule. */
gtars_status gtars_tokenizer_encode_ids(const gtars_tokenizer_t *t, const uint32_t *chrom_ids,
                                        const uint32_t *starts, const uint32_t *ends, uint64_t n,
                                        uint64_t *offsets, uint32_t **out_ids, uint64_t *out_n);

/* ------------------------------------------------------------------------
 * Fragment files as SoA columns (parse_fragment_line, utils/fragments.rs:12-40):
 * `chr start end barcode count` split on whitespace, lines starting with '#'
 * skipped, fewer than 5 fields / unparsable start or end -> GTARS_ERR_PARSE with
 * the reference's message and 0-based line number.  .gz by extension.  The text
 * is parsed in place by all host threads (GTARS_HOST_THREADS overrides);
 * chromosome and barcode ids are dictionary codes in first-seen order.
 * ---------------------------------------------------------------------- */
typedef struct gtars_fragments gtars_fragments_t;
gtars_status gtars_fragments_read(const char *path, gtars_fragments_t **out);
/* the same with the fifth field (read support) required to parse as u32, as Fragment::from_str does
 * (gtars-core/src/models/fragments.rs:16-41: the parser behind gtars-scoring) */
gtars_status gtars_fragments_read_strict(const char *path, gtars_fragments_t **out);
/* BED3 text mode of the `gtars overlaprs` front end (gtars-cli/src/overlaprs/handlers.rs:64-92, 123-139): EVERY line is a
 * record (no header / comment skipping), fields split on TAB only, start and end through str::parse::<u32>; errors name
 * the file and the 1-based line ("Missing start field", "Missing end field", "invalid digit found in string").  The
 * columns come back in a gtars_fragments_t without barcodes (file order, chromosome ids in first-seen order). */
gtars_status gtars_bed3_lines_read(const char *path, gtars_fragments_t **out);
/* the front end's output (handlers.rs:141-150): one line chr<TAB>start<TAB>end per hit, in the order given; *out_text is
 * malloc'ed (gtars_free), NUL-terminated, *out_len bytes long */
gtars_status gtars_format_hit_lines(const char *const *chrom_names, const uint32_t *hit_chrom, const uint32_t *hit_start,
                                    const uint32_t *hit_end, uint64_t n, char **out_text, uint64_t *out_len);
void gtars_fragments_free(gtars_fragments_t *f);
uint64_t gtars_fragments_len(const gtars_fragments_t *f);
uint32_t gtars_fragments_n_chrom(const gtars_fragments_t *f);
uint32_t gtars_fragments_n_barcodes(const gtars_fragments_t *f);
const char *gtars_fragments_chrom_name(const gtars_fragments_t *f, uint32_t id);
const char *gtars_fragments_barcode_name(const gtars_fragments_t *f, uint32_t id);
const uint32_t *gtars_fragments_chrom_ids(const gtars_fragments_t *f);
const uint32_t *gtars_fragments_starts(const gtars_fragments_t *f);
const uint32_t *gtars_fragments_ends(const gtars_fragments_t *f);
const uint32_t *gtars_fragments_barcode_ids(const gtars_fragments_t *f);

/* tokenize_fragment_file (utils/fragments.rs:61-82): one single-region
 * tokenize per fragment line (so every non-overlapping fragment yields one
 * unk id), grouped by barcode.  Result: n_barcodes names (first-seen order)
 * and a CSR of ids per barcode. */
typedef struct gtars_fragment_tokens {
    uint64_t n_barcodes;
    char **barcodes;      /* n_barcodes strings */
    uint64_t *offsets;    /* n_barcodes + 1 */
    uint32_t *ids;        /* offsets[n_barcodes] */
} gtars_fragment_tokens_t;
gtars_status gtars_tokenizer_tokenize_fragment_file(const gtars_tokenizer_t *t, const char *path,
                                                    gtars_fragment_tokens_t **out);
void gtars_fragment_tokens_free(gtars_fragment_tokens_t *ft);
/* the barcodes of `ft` in one buffer, '\n'-separated (no newline behind the last; barcodes are whitespace-free fields): one call
 * for a binding that would otherwise convert n_barcodes C strings one by one (19,200 of them were a fifth of the fused
 * pipeline's 48-file call from Python).  *out: gtars_free. */
gtars_status gtars_fragment_tokens_barcodes_joined(const gtars_fragment_tokens_t *ft, char **out, uint64_t *out_len);

/* ------------------------------------------------------------------------
 * gtars-fragsplit: pseudobulking of fragment files by a barcode -> cluster map.
 *   gtars_barcode_map_from_file   BarcodeToClusterMap::from_file (gtars-fragsplit/src/map.rs:34-81): one
 *                                 `<file stem>+<barcode> <cluster>` pair per line, split on whitespace, later lines win;
 *                                 a line with fewer than two fields -> GTARS_ERR_PARSE ("Invalid line format ...")
 *   gtars_fragsplit               pseudobulk_fragment_files (split.rs:36-151): every regular file of `files_dir`
 *                                 (.gz by extension), lines split on whitespace into chr start end barcode
 *                                 read_support (fewer than five fields -> GTARS_ERR_PARSE "Failed to parse fragments
 *                                 file at line {0-based index}: {line}"), looked up as "{stem}+{barcode}" with the stem
 *                                 stripped of ALL extensions (utils.rs remove_all_extensions), and written as
 *                                 "chr\tstart\tend\tbarcode\tread_support\n" to <out_dir>/cluster_<id>.bed.gz (one
 *                                 file per cluster label, also when empty; gzip level 6).  The reference visits the
 *                                 files in read_dir order (unspecified); here they are visited in byte order of their
 *                                 names, so the output is deterministic.  Files are parsed by all host threads.
 *   gtars_fragsplit_tokenize      the "gtars-fragsplit -> tokenizer" pipeline without the intermediate files: for every
 *                                 cluster (labels in byte order) exactly what tokenize_fragment_file returns for
 *                                 cluster_<id>.bed.gz -- the routed lines are parsed like fragment-file lines ('#'
 *                                 lines skipped, start / end must parse as u32), one batched GPU tokenization per cluster.
 * ---------------------------------------------------------------------- */
typedef struct gtars_barcode_map gtars_barcode_map_t;
gtars_status gtars_barcode_map_from_file(const char *path, gtars_barcode_map_t **out);
void gtars_barcode_map_free(gtars_barcode_map_t *m);
uint64_t gtars_barcode_map_len(const gtars_barcode_map_t *m);
uint32_t gtars_barcode_map_n_clusters(const gtars_barcode_map_t *m);
/* i-th cluster label in byte order */
const char *gtars_barcode_map_cluster_label(const gtars_barcode_map_t *m, uint32_t i);
/* cluster label of a "stem+barcode" key, NULL when it is not in the map */
const char *gtars_barcode_map_lookup(const gtars_barcode_map_t *m, const char *key);
gtars_status gtars_fragsplit(const char *files_dir, const gtars_barcode_map_t *m, const char *out_dir,
                             uint64_t *n_reads, uint64_t *n_written);
/* *out: array of n_clusters results (gtars_fragment_tokens_free each, gtars_free the array) */
gtars_status gtars_fragsplit_tokenize(const gtars_tokenizer_t *t, const char *files_dir, const gtars_barcode_map_t *m,
                                      gtars_fragment_tokens_t ***out, uint64_t *n_reads);
/* the same pipeline over an explicit list of fragment files, visited in the order given: what one rank of the multi-GPU
 * driver runs on its run of the directory's sorted file list (files are independent; the per-cluster results
 * of consecutive runs merge by concatenation per barcode, gtars_amd/sharding.py fragsplit_tokenize_sharded) */
gtars_status gtars_fragsplit_tokenize_files(const gtars_tokenizer_t *t, const char *const *paths, uint64_t n_paths,
                                            const gtars_barcode_map_t *m, gtars_fragment_tokens_t ***out, uint64_t *n_reads);

/* Diagnostic: seconds per stage of the calling thread's last gtars_fragsplit_tokenize(_files) call.  out12 = { 1 if the text was
 * parsed on the device (else on the host threads), waves, read + inflate on the host threads (host parser: + parse + route),
 * per-cluster append (host parser only), device waves / tokenizer calls in total, of which behind the last wave's files,
 * regroup by barcode, then for the device waves: text to the device, line split + parse + sort by cluster, gather, tokenize,
 * results to the host }. */
void gtars_fragsplit_last_stages(double *out12);

/* Host threads one call of the file pipelines above starts at most: hardware threads, capped by the container's CPU quota
 * (cgroup v2 cpu.max) and by `cap`, divided by LOCAL_WORLD_SIZE when the process is one of several ranks of a launcher on this
 * node (torch.distributed.run sets it; each rank of the sharded fragment pipeline inflates and parses on its share of the cores);
 * GTARS_HOST_THREADS overrides.  Diagnostic: the reference has no threads on this path (gtars-fragsplit/src/split.rs:36-151). */
uint32_t gtars_host_threads(uint32_t cap);

/* get_dynamic_reader (gtars-core/src/utils.rs:115-126) as one call: the file's bytes, gunzipped iff its extension is "gz"
 * (concatenated members decoded one after the other, every member's CRC-32 and length checked; a ".gz" without the gzip magic is
 * returned as it is).  What every file front end above reads through.  *out: malloc'ed (gtars_free), *out_n bytes.
 * GTARS_ERR_IO with the reader's message otherwise. */
gtars_status gtars_read_file(const char *path, char **out, uint64_t *out_n);

/* ------------------------------------------------------------------------
 * .gtok  (gtars-io/src/gtok.rs:125-210, consts.rs:1-3)
 * ---------------------------------------------------------------------- */
gtars_status gtars_gtok_write(const char *path, const uint32_t *tokens, uint64_t n);
gtars_status gtars_gtok_read(const char *path, uint32_t **out_tokens, uint64_t *out_n);

/* ------------------------------------------------------------------------
 * IGD database built from BED files (gtars-igd/src/igd.rs:170-242, 850-867)
 * ---------------------------------------------------------------------- */
typedef struct gtars_igddb gtars_igddb_t;

/* Igd::from_bed_files: unreadable files and files without a parseable line
 * are skipped; lines with start < 0 are parsed but not added. */
gtars_status gtars_igddb_from_bed_files(const char *const *paths, uint64_t n_paths,
                                        gtars_igddb_t **out);
/* Igd::from_bed_dir: *.bed / *.gz regular files of the directory, sorted */
gtars_status gtars_igddb_from_bed_dir(const char *dir, gtars_igddb_t **out);
void gtars_igddb_free(gtars_igddb_t *db);
uint32_t gtars_igddb_n_files(const gtars_igddb_t *db);
uint32_t gtars_igddb_n_contigs(const gtars_igddb_t *db);
/* FileInfo (igd.rs:52-59) */
const char *gtars_igddb_file_name(const gtars_igddb_t *db, uint32_t i);
uint32_t gtars_igddb_file_num_regions(const gtars_igddb_t *db, uint32_t i);
double gtars_igddb_file_avg_width(const gtars_igddb_t *db, uint32_t i);
int64_t gtars_igddb_chrom_id(const gtars_igddb_t *db, const char *chr); /* -1 unknown */
const char *gtars_igddb_chrom_name(const gtars_igddb_t *db, uint32_t id); /* contigs in creation order */
const gtars_igd_t *gtars_igddb_engine(const gtars_igddb_t *db);        /* borrowed */
/* Igd::count_set_overlaps (binary=0) / count_region_hits (binary=1) of a region set */
gtars_status gtars_igddb_count_regionset(const gtars_igddb_t *db, const gtars_regionset_t *rs,
                                         int32_t min_overlap, int binary, uint64_t *hits);

/* Igd::save / Igd::from_igd_file (gtars-igd/src/igd.rs:320-486): the .igd v1 file -- LE i32 header {nbp, gType = 1,
 * nCtg}, tiles per contig, record counts per tile, 40-byte NUL-padded contig names, 16-byte records {file idx, start,
 * end, value} tile by tile (a record is written once for every nbp-tile it touches) -- and its companion
 * <stem>.tsv ("Index\tFile\tNumber of Regions\tAvg size").  On load a record is kept from the tile it starts in, so
 * the device holds every stored interval once; gType 0 files (12-byte records) load with value 0.
 * gtars_igddb_from_arrays builds a database handle from columns the caller has already parsed (chromosome ids index
 * chrom_names; file i is described by file_names[i], num_regions[i], avg_width[i]). */
gtars_status gtars_igddb_from_arrays(const char *const *chrom_names, uint32_t n_chrom, const uint32_t *chrom,
                                     const int32_t *start, const int32_t *end, const int32_t *value,
                                     const uint32_t *file_idx, uint64_t n, const char *const *file_names,
                                     const uint32_t *num_regions, const double *avg_width, uint32_t n_files,
                                     gtars_igddb_t **out);
gtars_status gtars_igddb_save(const gtars_igddb_t *db, const char *path, int32_t nbp);
gtars_status gtars_igddb_load(const char *path, gtars_igddb_t **out, int32_t *nbp);

/* ------------------------------------------------------------------------
 * LOLA statistics tail  (gtars-lola/src/enrichment.rs:19-169, 243-394; output.rs:35-113)
 * ---------------------------------------------------------------------- */
/* Everything run_lola computes from the contingency cells, for all tables of a run in one threaded call.
 * a, b, c, d: [n_user_sets x n_db] row-major i64 cells (what gtars_lola_contingency_device leaves; b, c, d may be
 * negative: such a table gets pValueLog 0 and oddsRatio NaN, enrichment.rs:226-247).  direction: 0 enrichment
 * (p = sf(a - 1)), 1 depletion (p = cdf(a)).  Outputs, same layout, row = user set * n_db + db set:
 *   p_value_log  -log10(p + 1e-322)                                  (enrichment.rs:166-169)
 *   odds_ratio   conditional MLE as R's fisher.test: NaN for a one-point support, 0 / inf at its ends (:62-160)
 *   rnk_pv / rnk_or / rnk_sup / max_rnk / mean_rnk: min-ranks inside a user set, descending, NaN odds ratios last,
 *                ties by bit pattern (NaN == NaN, 0.0 != -0.0)       (:296-394)  -- all five NULL: values only
 *   order        the rows in the reference's output order: pValueLog descending, then meanRnk ascending, stable (:285-294)
 *   q_value      Benjamini-Hochberg per user set over the rows in that order (output.rs:35-113)
 * order / q_value may be NULL.  Hypergeometric sums and the odds-ratio equation are evaluated over the window of terms that
 * matter (see csrc/lola_stats.cpp); parity with the reference's statrs 0.18 values is to floating-point tolerance. */
gtars_status gtars_lola_stats(const int64_t *a, const int64_t *b, const int64_t *c, const int64_t *d, uint64_t n_db,
                              uint64_t n_user_sets, int direction, double *p_value_log, double *odds_ratio,
                              uint32_t *rnk_pv, uint32_t *rnk_or, uint32_t *rnk_sup, uint32_t *max_rnk, double *mean_rnk,
                              uint64_t *order, double *q_value);
/* rank_results (enrichment.rs:353-394) on the n rows of ONE user set, values given: min-ranks by pValueLog, oddsRatio (NaN
 * last) and support, all descending; maxRnk and meanRnk. */
gtars_status gtars_lola_rank(const double *p_value_log, const double *odds_ratio, const uint64_t *support, uint64_t n,
                             uint32_t *rnk_pv, uint32_t *rnk_or, uint32_t *rnk_sup, uint32_t *max_rnk, double *mean_rnk);
/* apply_fdr_correction (output.rs:35-113) on n_rows result rows in the order they stand: Benjamini-Hochberg q-values per
 * user set (q_value[r] for row r). */
gtars_status gtars_lola_fdr(const double *p_value_log, const uint64_t *user_set, uint64_t n_rows, double *q_value);
/* ContingencyTable::fisher_pvalue / odds_ratio of one table (enrichment.rs:19-53, 62-160) */
double gtars_lola_fisher_pvalue(uint64_t a, uint64_t b, uint64_t c, uint64_t d, int direction);
double gtars_lola_odds_ratio(uint64_t a, uint64_t b, uint64_t c, uint64_t d);

#ifdef __cplusplus
}
#endif
#endif
