// Internal declarations shared by the HIP translation units of libgtars_amd.so.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/gtars_amd.h"

namespace gtars {

using u8 = uint8_t;
using u32 = uint32_t;
using u64 = uint64_t;
using i32 = int32_t;
using i64 = int64_t;

// ---- error plumbing --------------------------------------------------------
void set_error(const std::string &msg);
gtars_status fail(gtars_status st, const std::string &msg);
gtars_status hip_fail(hipError_t e, const char *what, const char *file, int line);

#define GT_HIP(expr)                                                            \
    do {                                                                        \
        hipError_t _e = (expr);                                                 \
        if (_e != hipSuccess) return ::gtars::hip_fail(_e, #expr, __FILE__, __LINE__); \
    } while (0)

gtars_status require_device();

// ---- environment switches -----------------------------------------------------
// Every GTARS_* variable is read ONCE, into an immutable snapshot taken at first use (a getenv per call raced with a host
// program's setenv, and a switch flipped mid-process reached some kernels and not others).  cfg_get returns the snapshot's
// value or null.  gtars_debug_reload_env() (test hook, not in the public headers' contract) takes a new snapshot: tests and
// the bench harness call it after they change a switch.
const char *cfg_get(const char *name);
inline bool cfg_flag(const char *name) { return cfg_get(name) != nullptr; }
long cfg_int(const char *name, long dflt);

// ---- device views ----------------------------------------------------------
// One genome-wide overlap index: every chromosome's intervals concatenated in
// chromosome-id order, SoA, u32.
//  Bits kind  : within a chromosome sorted by (start, end, input order)
//               (bits.rs:105); chrom_aux[c] = max_len (bits.rs:110-119).
//  AIList kind: within a chromosome sub-list major (ailist.rs:127-141), each
//               sub-list sorted by start; max_ends = running max of ends per
//               sub-list (ailist.rs:229-235); sub_off[chrom_sub[c]..chrom_sub[c+1]]
//               are the sub-list start offsets (global positions), with one
//               terminating entry per chromosome.
struct IndexView {
    const u32 *starts;
    const u32 *ends;
    const u32 *vals;
    const u32 *max_ends;   // AIList only
    const u32 *chrom_off;  // [n_chrom + 1]
    const u32 *chrom_aux;  // Bits: max_len [n_chrom]
    const u32 *chrom_sub;  // AIList: [n_chrom + 1] into sub_off
    const u32 *sub_off;    // AIList: sub-list boundaries (global positions)
    u32 n_chrom;
    u32 n;
};

// Blocked acceleration structure of a Bits-kind index (built once at index
// build).  Every chromosome's sorted intervals are cut into blocks of
// ACC_OWN = 2; a block's record carries its two intervals and copies of the
// NEXT block's two (the "look-ahead": the intervals the forward scan looks at
// next), so that one burst of 16-byte loads from one random line answers a
// query without a dependent load unless it reaches past BOTH look-ahead intervals
// (with one look-ahead interval 0.5 % of the C2 queries walked on -- some lane of
// two wave-tiles out of three, i.e. a dependent round trip for nearly every wave):
//     rec2 (32 B per block):   quad 0: s0 s1 ns0 ns1    quad 1: e0 e1 ne0 ne1
//     rec4 (64-B slots):       the same two quads + quad 2: v0 v1 nv0 nv1 (token ids)
// Why this shape: a CU's vector-memory path prices a divergent 16-byte request at 2.3 clk
// per lane and every further 16 bytes of the same record at ~1 clk (tools/ubench/ta.hip:
// 1 / 2 / 3 / 4 quads = 2.3 / 2.4 / 3.2 / 4.0 clk per query), and that path -- not HBM --
// is what the tokenizer saturates.  rec2 serves counting, and tokenizing whenever the ids
// follow from the position: ids_affine = every chromosome's stored values ascend by one
// in stored order (a universe file sorted by position), id = ACC_OWN * block + slot + idc[chrom]
// (mod 2^32).  rec4 exists only for indexes whose ids do not.
// Unused slots are sentinels (start = 0xFFFFFFFF, end = 0: never overlap and
// stop the forward scan).  blk_first[b] = the block's search key: the largest END
// among all intervals of the chromosome up to and including the block's own (a
// prefix maximum, so it ascends; 0xFFFFFFFF for padding blocks).  Only intervals
// with end > q_start can overlap a query, so the first block whose key is
// > q_start is where the scan starts: same hits, same order as Bits::find's
// lower_bound(q_start - max_len), but never earlier than necessary.
//
// Search structure kept in LDS by the workgroups, over "units" of 2^top_shift blocks
// (each chromosome's block range is padded to a multiple of 2^top_shift):
//  * all unit keys (key of the unit's last block) live in ONE ascending key space:
//    chromosome c's keys are offset by gbase[c] (sum of span + 2^q_shift of the earlier
//    chromosomes, span = max end + 1) and sentinel keys become gbase[c] + span[c];
//  * `lut[b]` (u16) = number of units whose key is < b << lut_shift: a direct-mapped table
//    over the key space -- genomic positions spread evenly, so a bucket holds a handful of
//    units and the search inside it takes `search_top` = 2^(steps-1) halving steps;
//  * `qkeys[u]` (u16) = (key(u) mod 2^lut_shift) >> q_shift, floor-quantised.  The search
//    may stop a unit early (never late, never in an earlier chromosome: those are 2^q_shift
//    away); the record scan then simply walks on, so results stay exact.
// chrom_tab[c] = {gbase low word, span, gbase high word, end of the chromosome's block range}.
constexpr int ACC_OWN = 2;   // own intervals of a block
constexpr int ACC_SLOTS = 4; // intervals a record carries (own + look-ahead)
struct AccelView {
    const uint4 *rec2;        // [n_blocks * 2] starts | ends (32 B per block)
    const uint4 *rec4;        // [n_blocks * 4] starts | ends | ids | unused; null when ids_affine
    // [n_units * 4] when top_shift == 1 (two blocks per LDS key: universes of ~130k-260k regions) and ids_affine, null otherwise: the
    // UNIT's record -- the four own intervals of its two blocks and, as look-ahead, the four of the next two blocks: 8 starts | 8 ends,
    // 64 bytes.  One request per query instead of the block key (blk_first) and then a 32-byte record (k_tok_lds<.., U64>).
    const uint4 *rec8;
    const u32 *blk_first;     // [n_blocks] prefix-max end up to each block (local coordinates)
    const u32 *lut;           // [lut_words] packed u16, n_buckets + 1 entries (16-byte padded)
    const u32 *qkeys;         // [q_words] packed u16, n_units entries (16-byte padded)
    const uint4 *chrom_tab;   // [n_chrom] {gbase lo, span, gbase hi, blk_end}
    const u32 *idc;           // [n_chrom] id of (block b, slot k) = ACC_OWN * b + k + idc[c] when ids_affine
    u32 n_blocks;
    u32 n_units;
    u32 n_buckets;
    u32 lut_words, q_words;   // multiples of 4
    u32 lut_shift;
    u32 q_shift;
    u32 search_top;           // first step of the in-bucket search (power of two, 0: buckets hold <= 0 units)
    u32 top_shift;
    u32 n_chrom;
    u32 ids_affine;
    u32 max_chrom_n;  // most intervals on one chromosome: bounds a query's hits (tile totals are 32-bit)
    // Run form of wide queries (tail_run, tokenize_lds.hip): the intervals behind a query's first record that start before q_end
    // are ALL hits -- one run of stored positions, measured by a second search instead of walked -- whenever they all end after
    // q_start, which holds
    //   runs_ok   (no interval of the index is inverted, start <= end) and the record's fourth interval starts after q_start
    //             (everything behind it then starts, hence ends, after q_start): any universe, most wide queries; or
    //   ends_mono (on every chromosome the ends also ascend with the starts: disjoint universes, what consensus peak sets are)
    //             and the first record has a hit (everything behind a hit then ends after q_start as well).
    // chrom_iv_end[c] = ACC_OWN * first block of c + intervals of c that start below 0xFFFFFFFF.
    u32 runs_ok, ends_mono;
    const u32 *chrom_iv_end;
};

// IGD database: all stored intervals (tile replicas are NOT materialised),
// chromosome-major, sorted by (start, insertion order) within a chromosome.
struct IgdView {
    const i32 *starts;
    const i32 *ends;
    const u32 *files;
    const i32 *values;
    const u32 *chrom_off;  // [n_chrom + 1]
    const i32 *chrom_maxlen;  // [n_chrom] max(end-start)
    // [n_chrom] tiles of the reference's contig at nbp = 16384, (largest end - 1) / 16384 + 1 -- what the walk of
    // igd.rs:772-846 needs to reproduce min_overlap <= 0; null until such a query is made (gtars_igd::ensure_ntiles)
    const i32 *chrom_ntiles;
    // IgdTiles::pm when built (null otherwise): prefix maximum of the ends inside a chromosome -- the per-query kernels start a
    // min_overlap >= 1 scan at the first record whose prefix-max end is > q_start
    const i32 *pm;
    u32 n_chrom;
    u32 n;
    u32 n_files;
    // 1: a PIECES view (api.hip, build_pieces_view): records longer than the piece length were cut into pieces, bit 31 of files[] (bit 15
    // of the u16 copy) marks a continuation piece, which counts for a query only if it starts at or before the query's start
    u32 pieces;
};
constexpr u32 IGD_FILE_MASK = 0x7FFFFFFFu;

// ---- per-thread grow-only device workspace ----------------------------------
// host-side bookkeeping of a chained-scan workspace that is reused across launches without
// being cleared (see scan.h): current epoch, tickets drawn so far, bytes known to be zeroed
struct ScanEpoch {
    u32 epoch = 0;
    u32 ticket_base = 0;
    size_t cleared_bytes = 0;
};

struct Workspace {
    void *ptr = nullptr;
    size_t bytes = 0;
    int device = -1;
    ScanEpoch ep;
    u32 igd_calls = 0;  // IGD sweeps served by this buffer (launch_igd_sweep: the parity picks one of two flag words at its head)
    gtars_status reserve(size_t need);
    ~Workspace();
};
Workspace &tls_workspace(int slot, hipStream_t stream);

// ---- launchers (kernels.hip) ------------------------------------------------
struct EnumOut {
    u64 *offsets;    // [nq+1] device
    u32 *vals;       // may be null
    u32 *starts;     // may be null
    u32 *ends;       // may be null
    u64 capacity;    // elements available in each non-null output
    int hint = 0;    // GTARS_TOK_AUTO / _NARROW / _WIDE (gtars_amd.h): which build of the LDS tokenizer a launch runs
    bool sorted = false;  // GTARS_TOK_SORTED: the batch is in (chromosome, start) order -> the sweep form (k_tok_sweep)
};

// Head of every fused-scan workspace; after the launch has completed the host
// reads {err, total} from here (total = H, err != 0: look-back spin limit hit).
struct ScanHead {
    u32 ticket;
    u32 err;
    u64 total;
};

// fused single pass (chained scan); scan_ws is zeroed by the launcher.
gtars_status launch_enumerate_fused(const IndexView &v, int kind, const u32 *qc, const u32 *qs,
                                    const u32 *qe, u64 nq, int has_min, i32 min_overlap,
                                    const EnumOut &out, void *scan_ws, size_t scan_ws_bytes,
                                    hipStream_t st);
u32 enumerate_fused_tile_queries();
size_t enumerate_fused_ws_bytes(u64 nq);

// LDS-tiled fused tokenizer (tokenize_lds.hip), Bits order only
// d_base (may be null): device word holding what precedes this launch's first offset (chained launches of one batch);
// d_total_out (may be null): receives base + this launch's hits; reverse: every query's hits in descending stored
// order (AIList::find order of a single-sub-list index)
gtars_status launch_tokenize_lds(const AccelView &a, const u32 *qc, const u32 *qs, const u32 *qe, u64 nq,
                                 int has_min, i32 min_overlap, const EnumOut &out, void *scan_ws,
                                 size_t scan_ws_bytes, ScanEpoch &ep, hipStream_t st, const u64 *d_base = nullptr,
                                 u64 *d_total_out = nullptr, bool reverse = false);
size_t tokenize_lds_ws_bytes(u64 nq);
bool tokenize_lds_supported(const AccelView &a);
// the same call for a batch in (chromosome, start) order (k_tok_sweep: a tile of consecutive queries stages its contiguous slice of the
// blocked records in LDS; any universe size)
gtars_status launch_tokenize_sweep(const AccelView &a, const u32 *qc, const u32 *qs, const u32 *qe, u64 nq, int has_min, i32 min_overlap,
                                   const EnumOut &out, void *scan_ws, size_t scan_ws_bytes, ScanEpoch &ep, hipStream_t st,
                                   const u64 *d_base, u64 *d_total_out, bool reverse);
bool tokenize_sweep_supported(const AccelView &a);

gtars_status launch_count_lds(const AccelView &a, const u32 *qc, const u32 *qs, const u32 *qe, u64 nq, int has_min,
                              i32 min_overlap, u32 *counts, u8 *any, hipStream_t st);
gtars_status launch_mark_lds(const AccelView &a_pos, const u32 *qc, const u32 *qs, const u32 *qe, u64 nq, int has_min, i32 min_overlap,
                             u32 *mark, hipStream_t st);
gtars_status launch_bits_count(const IndexView &v, const u32 *ends_sorted, const u32 *qc, const u32 *qs, const u32 *qe,
                               u64 nq, u64 *out, hipStream_t st);
gtars_status launch_count(const IndexView &v, int kind, const u32 *qc, const u32 *qs, const u32 *qe,
                          u64 nq, int has_min, i32 min_overlap, u32 *counts, u8 *any,
                          hipStream_t st);
// ids for existing offsets
gtars_status launch_fill(const IndexView &v, int kind, const u32 *qc, const u32 *qs, const u32 *qe,
                         u64 nq, int has_min, i32 min_overlap, const u64 *offsets, u32 *vals,
                         u32 *starts, u32 *ends, hipStream_t st);
// payload columns of hits given by stored position (any of vals / starts / ends may be null)
gtars_status launch_gather_hits(const IndexView &v, const u32 *pos, u64 n, u32 *vals, u32 *starts, u32 *ends, hipStream_t st);
// exclusive scan u32 counts -> u64 offsets[n+1]
gtars_status launch_scan_u32_to_u64(const u32 *counts, u64 n, u64 *offsets, void *ws, size_t ws_bytes,
                                    hipStream_t st);
size_t scan_ws_bytes(u64 n);

// pme_file (may be null): IgdTiles::pme_file, used for binary counts with min_overlap == 1
// (min_overlap < 1 needs v.chrom_ntiles)
// pm (IgdTiles::pm, may be null): the scan of a query starts at the first record whose prefix-max end is > q_start
gtars_status launch_igd_count(const IgdView &v, const i32 *pme_file, const u32 *qc, const u32 *qs, const u32 *qe, u64 nq,
                              i32 min_overlap, int binary, u64 *hits, hipStream_t st, const i32 *pm = nullptr);
gtars_status launch_occupy(u32 workgroups, u32 lds_bytes, u32 microseconds, hipStream_t st);
gtars_status launch_hist_u32(const u32 *ids, u64 n, u32 n_bins, u32 *bins, hipStream_t st);
gtars_status launch_hist_rows(const u64 *offsets, const u32 *ids, const u32 *row, u64 nq, u32 row0, u32 n_rows, u32 n_cols, u32 *mat,
                              hipStream_t st);
gtars_status launch_has_adjacent_equal(const u32 *a, u64 n, u32 *dup, hipStream_t st);
gtars_status launch_permute_marks(const u32 *in, const u32 *map, u64 n, u32 *out, hipStream_t st);  // out bit map[p] |= in bit p
gtars_status launch_igd_count_per_query(const IgdView &v, const u32 *qc, const u32 *qs, const u32 *qe,
                                        u64 nq, i32 min_overlap, u32 *counts, bool unique_values, hipStream_t st);
gtars_status launch_igd_fill_pairs(const IgdView &v, const u32 *qc, const u32 *qs, const u32 *qe, u64 nq,
                                   i32 min_overlap, const u64 *offsets, u32 *out_q, u32 *out_s,
                                   bool unique_values, hipStream_t st);
gtars_status launch_lola_contingency(const u64 *user_hits, const u64 *universe_hits, u64 n_files,
                                     i64 user_size, i64 universe_size, i64 *a, i64 *b, i64 *c, i64 *d,
                                     hipStream_t st);
// a nested AIList index's enumeration order from its flat companion's hits (kernels.hip: k_ailist_reorder)
gtars_status launch_ailist_reorder(u32 *ids, const u64 *offsets, u64 nq, u64 capacity, const u32 *key_by_pos, const u32 *val_by_key,
                                   hipStream_t st);
gtars_status launch_sort_unique_segments(u32 *vals, const u64 *offsets, u64 nq, u32 *new_counts,
                                         hipStream_t st);

// K1 (sort.hip): one-pass partition of (a, b) pairs by a small key (< n_bins <= MS_MAX_BINS), not stable: out_ab receives
// the pairs interleaved, bin_off[n_bins + 1] the bin boundaries; elements whose key is `drop_bin` are left out (their bin
// must be the last one)
constexpr u32 MS_MAX_BINS = 36864;       // one-level split: 144 KB of LDS counters
constexpr u32 MS_MAX_BINS_2L = 65535;    // two-level split (no per-bin LDS state): what 16-bit keys can name
size_t multisplit_ws_bytes(u32 n_bins, u32 n);
// key: 16-bit keys (n_bins <= 65535); `a` is clamped on the way, a = max((i32)a, 0) (raw IGD query starts, igd.rs:517).  The
// caller's own kernel has COUNTED the keys before the call (see multisplit_pairs in sort.hip for what it leaves where).
// Bins (tiles of the IGD sweep) far heavier than the average, listed in PARTS of `part` elements so that their consumer can hand
// them to several workgroups: part 0 of every bin is implied, parts 1 .. ceil(total / part) - 1 of a bin with more than `part`
// elements are appended to list[] as (bin, part) -- at most cap entries (sum over bins of total / part never exceeds n / part);
// *count is zeroed by the caller.  part == 0: nothing is listed.
struct HeavyBins {
    uint2 *list = nullptr;
    u32 *count = nullptr;
    u32 part = 0, n_real_bins = 0, cap = 0;
    __device__ __forceinline__ void note(u32 bin, u32 total) const {
        if (!part || bin >= n_real_bins || total <= part) return;
        const u32 extra = (total - 1) / part;  // parts 1 .. extra
        const u32 at = atomicAdd(count, extra);
        for (u32 p = 0; p < extra; ++p)
            if (at + p < cap) list[at + p] = make_uint2(bin, p + 1);
    }
};
gtars_status multisplit_pairs(const unsigned short *key, const u32 *a, const u32 *b, u32 n, u32 n_bins, u32 drop_bin, uint2 *out_ab,
                              u32 *bin_off, void *ws, size_t ws_bytes, hipStream_t st, const u32 *run_if, const u32 *set_bounds,
                              const HeavyBins *heavy, u32 n_count_rows);
// set_bounds (host, 3 values; null: one set): the input holds up to 4 row ranges ("sets") -- first row of set 1, 2, 3, 0xFFFFFFFF
// for a set that does not exist; the set of a row leaves in bit 31 of its pair (b: set & 1, a: set >> 1; see SetTags, sort.hip)
u32 multisplit_workgroups(u32 n);
u32 multisplit_chunk(u32 n);  // elements per workgroup of that grid (a multiple of 4)
u32 *multisplit_table(void *ws);
// two-level split (large n, many bins): start of the words the caller zeroes (multisplit_zeroed_words); null: one-level
u32 *multisplit_totals(void *ws, u32 n_bins, u32 n);
// ... and the totals of the split's coarse bins (bin >> multisplit_coarse_shift(n_bins), <= 256 of them) here, inside the zeroed words
u32 *multisplit_coarse_totals(void *ws, u32 n_bins, u32 n);
// ... and its fine counts as one ROW per counting workgroup at multisplit_table(ws): 16-bit counts packed two per word (a
// workgroup counts at most 65535 elements), multisplit_row_words(n_bins) words per row (zero-padded to whole 16-byte vectors)
__host__ __device__ inline u32 multisplit_row_words(u32 n_bins) { return (((n_bins + 1u) >> 1) + 3u) & ~3u; }
u32 multisplit_coarse_shift(u32 n_bins);
size_t multisplit_zeroed_words(u32 n_bins);  // what such a caller zeroes from multisplit_totals() on (totals + relative cursors)

// K1 (sort.hip): permutation that orders rows by (chrom, k1, [k2], input order); device columns in/out
gtars_status device_sort_perm(const u32 *d_chrom, const u32 *d_k1, const u32 *d_k2, u32 n, u32 n_chrom, u32 *d_perm,
                              hipStream_t st);

size_t device_sort_perm_ws_bytes(u32 n);
gtars_status device_sort_perm_ws(const u32 *d_chrom, const u32 *d_k1, const u32 *d_k2, u32 n, u32 n_chrom, u32 *d_perm,
                                 void *scratch, size_t scratch_bytes, hipStream_t st);
// IGD batch sweep (igd_sweep.hip): the database cut into tiles of IGD_TILE_RECORDS consecutive records of one chromosome
struct IgdTiles {
    const u32 *first, *cnt, *chrom;  // [n_tiles] first record, record count, chromosome
    const i32 *carry;                // [n_tiles] largest end among the chromosome's records before the tile (0: none)
    const u32 *bnd;                  // [n_tiles] ownership bound: last start + max_len + 1 (saturating); a query
                                     //   (c, s) is owned by the first tile of chromosome c with bnd > s
    const u32 *chrom_tile_off;       // [n_chrom + 1] tiles of each chromosome
    const i32 *pme_file;             // [n] largest end among the EARLIER records of the same file and chromosome (0: none),
                                     //   or null: not built yet
    // static per-tile tables of the sweep (igd_sweep.hip, launch_igd_tile_tables), built with the index:
    const i32 *pm;                   // [n] prefix maximum of the ends over the chromosome's records up to each record
    const unsigned short *files16;   // [n] file ids as u16 (null when n_files > 65535: such a database is never swept)
    const u32 *tab;                  // [n_tiles * IGD_TILE_TAB_WORDS] descriptor + two search tables per tile
    // rank-histogram form of the sweep (k_igd_sweep_rank, round 5), per tile a block of IGD_TILE_BLOCK slots (null: not built):
    const i32 *ends_sorted;          // the ends of the tile's staged records (tile + halo) in ascending order, padded with INT_MAX
    const unsigned short *erank;     // slot (p0 & 3) + i: position of staged record i's end in that order
    const u32 *tab_r;                // [n_tiles * IGD_TILE_TABR_WORDS] descriptor + the search tables over the starts and the sorted ends
    // static routing table (owner tile of a query): entry route_base[c] + j (u16, packed two per word) = first tile of
    // chromosome c whose ownership bound is > (j << route_shift); route_len[c] = the chromosome's last bound (0: no tiles);
    // null when the database has more than 65535 tiles
    const u32 *route_lut, *route_base, *route_len;
    u32 route_n, route_shift;
    // FINE routing tables (null when they would not fit the routing kernel's LDS next to its counters): buckets of 2^route_fshift
    // (>= 2^16) positions, so fine that a bucket holds about one tile boundary, and the boundaries themselves as 16-bit offsets
    // inside their bucket.  key[t] = bnd[t] - 1 (the last position tile t owns); route_flut[route_fbase[c] + j] (u16, two per
    // word) = first tile of chromosome c with key >= j << route_fshift; route_kq[t] (u16, two per word) = (key[t] mod
    // 2^route_fshift) >> (route_fshift - 16): the owner of (c, s) is the first tile t in [flut[j], flut[j + 1]), j = s >>
    // route_fshift, with kq[t] > sq -- or == sq and, when route_fshift > 16, bnd[t] > s (a global read, one query in 2^16) -- else
    // flut[j + 1].  One or two LDS probes instead of a binary search of the chromosome's bounds.
    const u32 *route_flut, *route_kq, *route_fbase;
    u32 route_fn, route_fshift;
    u32 n_tiles;
};
// words per tile: a 12-word descriptor + the two direct-mapped search tables (u16 entries, buckets + 2 of them): over the
// staged starts (exact searches start from it) and over the prefix-max ends (its lower bracket IS the first candidate: the
// sweep never reads the prefix maxima themselves, so this table is the finer one)
#ifndef IGD_LUT_P_BUCKETS
#define IGD_LUT_P_BUCKETS 1024
#endif
#ifndef IGD_LUT_S_BUCKETS
#define IGD_LUT_S_BUCKETS 512
#endif
constexpr u32 IGD_LUT_S_NB = IGD_LUT_S_BUCKETS, IGD_LUT_P_NB = IGD_LUT_P_BUCKETS;
static_assert((IGD_LUT_S_NB & (IGD_LUT_S_NB - 1)) == 0 && IGD_LUT_S_NB >= 256 && IGD_LUT_S_NB <= 4096, "a power of two");
static_assert((IGD_LUT_P_NB & (IGD_LUT_P_NB - 1)) == 0 && IGD_LUT_P_NB >= 256 && IGD_LUT_P_NB <= 4096, "a power of two");
constexpr u32 IGD_TILE_TAB_WORDS = (12 + (IGD_LUT_S_NB + 2) / 2 + (IGD_LUT_P_NB + 2) / 2 + 3) / 4 * 4;  // 784 words: 3 KB per 2048 records
// the rank-histogram sweep's per-tile data: IGD_TILE_BLOCK = staged records (tile + halo) rounded up to whole 16-byte vectors
// behind a <= 3-slot alignment shift; its tables: a 12-word descriptor + two search tables of IGD_LUT_R_NB buckets
#ifndef GTARS_IGD_HALO
#define GTARS_IGD_HALO 256  // records staged behind a tile's own (a query's scan runs on into them; beyond: global memory, by the query's lane)
#endif
constexpr u32 IGD_TILE_BLOCK = 2048 + GTARS_IGD_HALO + 4;
constexpr u32 IGD_LUT_R_NB = 1024;
constexpr u32 IGD_TILE_TABR_WORDS = (12 + 2 * ((IGD_LUT_R_NB + 2) / 2) + 3) / 4 * 4;
gtars_status launch_igd_tile_tables_rank(const IgdView &v, const u32 *tile_first, const u32 *tile_cnt, const u32 *tile_chrom, u32 n_tiles,
                                         i32 *ends_sorted, unsigned short *erank, u32 *tab_r, hipStream_t st);
gtars_status launch_igd_tile_tables(const IgdView &v, const u32 *tile_first, const u32 *tile_cnt, const u32 *tile_chrom,
                                    const i32 *tile_carry, u32 n_tiles, i32 *pm, unsigned short *files16, u32 *tile_tab, hipStream_t st);
size_t igd_route_fine_lds_bytes(u32 n_tiles, u32 n_chrom, u64 n_fine);  // LDS of the routing kernel with the fine tables
bool igd_sweep_supported(const IgdView &v, u64 nq);
size_t igd_sweep_ws_bytes(u64 nq, u32 n_tiles, u32 n_chrom);
size_t igd_pme_ws_bytes(u32 n);
gtars_status igd_build_pme_file(const IgdView &v, i32 *pme, void *ws, size_t ws_bytes, hipStream_t st);
size_t radix_sort_ws_bytes(u32 n);
gtars_status radix_sort_pairs(u32 *k0, u32 *v0, u32 *k1, u32 *v1, u32 n, int begin_bit, int end_bit, void *ws, size_t ws_bytes,
                              int *result_in, hipStream_t st);
gtars_status launch_igd_tile_bounds(const IgdView &v, const u32 *tile_first, const u32 *tile_cnt, const u32 *tile_chrom, u32 n_tiles,
                                    u32 *bnd, hipStream_t st);
gtars_status launch_igd_tile_max_end(const i32 *ends, const u32 *tile_first, const u32 *tile_cnt, u32 n_tiles, i32 *tile_max,
                                     hipStream_t st);
// n_sets > 1: the batch is the concatenation of n_sets <= 4 query sets (set_bounds as for multisplit_pairs), hits is
// u64[n_sets][n_files]; needs igd_sweep_sets_supported
// ws: a workspace whose first 64 bytes were zero when it was allocated and are never touched by anybody else (Workspace::reserve
// zeroes them); call_no: how many sweeps this workspace has served before (Workspace::igd_calls++): its parity picks the order flag
gtars_status launch_igd_sweep(const IgdView &v, const IgdTiles &tl, const u32 *qc, const u32 *qs, const u32 *qe, u64 nq, i32 min_overlap,
                              int binary, u64 *hits, void *ws, size_t ws_bytes, hipStream_t st, u32 call_no, u32 n_sets = 1,
                              const u32 *set_bounds = nullptr);
bool igd_sweep_sets_supported(const IgdView &v, const IgdTiles &tl, u64 nq, u32 n_sets);
constexpr u32 IGD_TILE_RECORDS = 2048;
gtars_status device_gather_u32(const u32 *src, const u32 *idx, u32 n, u32 *dst, hipStream_t st);

// ---- profiling hooks --------------------------------------------------------
struct ProfScope {
    ProfScope(const char *name, hipStream_t st);
    ~ProfScope();
    const char *name;
    hipStream_t st;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    bool on = false;
};
// profiling mode only: counts one launch of entry `if_set` / `if_clear` by the state of a device word (synchronises `st`)
void prof_note_fact(const char *name);  // profiling mode only: counts one launch of the entry
void prof_note_device_flag(const char *if_set, const char *if_clear, const u32 *d_flag, hipStream_t st);

}  // namespace gtars
