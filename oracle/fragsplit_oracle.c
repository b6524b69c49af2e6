/*
 * fragsplit_oracle.c -- TEST INFRASTRUCTURE ONLY (see gtars_oracle.h).
 *
 * The "fragsplit -> tokenizer" pipeline of BASELINE config 5 on the CPU, one thread, compiled code: what bench.py times as
 * `cpu_baseline` of `fragsplit_config5` (kind "port"), so that the GPU figure stands next to a C figure and not next to a
 * pure-Python one.  It restates
 *   - pseudobulk_fragment_files (gtars-fragsplit/src/split.rs:36-151): every line split on whitespace into chr start end barcode
 *     read_support (fewer than five fields: an error), looked up as "{stem}+{barcode}" with the stem stripped of ALL extensions
 *     (gtars-core/src/utils.rs:372-387), routed to its cluster (unmapped barcodes are dropped);
 *   - tokenize_fragment_file (gtars-tokenizers/src/utils/fragments.rs:61-82) on every routed line: '#' lines skipped, start / end
 *     must parse as u32, one single-region tokenize per line (Tokenizer::tokenize, tokenizer.rs:140-163: unknown chromosome ->
 *     no hits; no hits at all -> the unk id), ids appended to the barcode's vector.
 * The per-cluster results are summarised (ids, sum of ids, distinct barcodes): enough for bench.py to check them against the
 * GPU pipeline's, which tests/test_gpu_host.py compares id by id with the Python oracle.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#include "gtars_oracle.h"

/* ---- a string -> u32 map (open addressing, FNV-1a), keys owned ---- */
typedef struct {
    char **keys;
    uint32_t *vals;
    uint64_t cap, n;
} smap;

static uint64_t fnv1a(const char *s, size_t n) {
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; ++i) {
        h ^= (unsigned char)s[i];
        h *= 1099511628211ull;
    }
    return h;
}
static void smap_init(smap *m, uint64_t cap) {
    m->cap = 16;
    while (m->cap < cap * 2) m->cap <<= 1;
    m->n = 0;
    m->keys = (char **)calloc(m->cap, sizeof(char *));
    m->vals = (uint32_t *)calloc(m->cap, sizeof(uint32_t));
}
static void smap_free(smap *m) {
    for (uint64_t i = 0; i < m->cap; ++i) free(m->keys[i]);
    free(m->keys);
    free(m->vals);
}
static void smap_grow(smap *m);
/* slot of key (len n): existing, or the empty slot where it would go */
static uint64_t smap_slot(const smap *m, const char *k, size_t n) {
    uint64_t i = fnv1a(k, n) & (m->cap - 1);
    while (m->keys[i] && !(strlen(m->keys[i]) == n && memcmp(m->keys[i], k, n) == 0)) i = (i + 1) & (m->cap - 1);
    return i;
}
static void smap_put(smap *m, const char *k, size_t n, uint32_t v) {
    if ((m->n + 1) * 2 > m->cap) smap_grow(m);
    const uint64_t i = smap_slot(m, k, n);
    if (!m->keys[i]) {
        m->keys[i] = (char *)malloc(n + 1);
        memcpy(m->keys[i], k, n);
        m->keys[i][n] = 0;
        m->n++;
    }
    m->vals[i] = v; /* later lines win (map.rs:34-81: HashMap::insert) */
}
static void smap_grow(smap *m) {
    smap o = *m;
    m->cap = o.cap * 2;
    m->n = 0;
    m->keys = (char **)calloc(m->cap, sizeof(char *));
    m->vals = (uint32_t *)calloc(m->cap, sizeof(uint32_t));
    for (uint64_t i = 0; i < o.cap; ++i)
        if (o.keys[i]) {
            const uint64_t j = smap_slot(m, o.keys[i], strlen(o.keys[i]));
            m->keys[j] = o.keys[i];
            m->vals[j] = o.vals[i];
            m->n++;
        }
    free(o.keys);
    free(o.vals);
}
static int smap_get(const smap *m, const char *k, size_t n, uint32_t *v) {
    const uint64_t i = smap_slot(m, k, n);
    if (!m->keys[i]) return 0;
    *v = m->vals[i];
    return 1;
}

/* per cluster: barcode -> growing vector of ids (HashMap<String, Vec<u32>>) */
typedef struct {
    uint32_t *p;
    uint64_t n, cap;
} idvec;
typedef struct {
    smap barcodes; /* barcode -> index into vecs */
    idvec *vecs;
    uint64_t n_vecs, cap_vecs;
} cluster;

static void idvec_push(idvec *v, uint32_t x) {
    if (v->n == v->cap) {
        v->cap = v->cap ? v->cap * 2 : 8;
        v->p = (uint32_t *)realloc(v->p, v->cap * sizeof(uint32_t));
    }
    v->p[v->n++] = x;
}

static int is_ws(char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\f' || c == '\v'; }

/* str::parse::<u32>: optional '+', digits only, no overflow */
static int parse_u32(const char *s, size_t n, uint32_t *out) {
    size_t i = 0;
    if (n && s[0] == '+') i = 1;
    if (i >= n) return 0;
    uint64_t v = 0;
    for (; i < n; ++i) {
        if (s[i] < '0' || s[i] > '9') return 0;
        v = v * 10 + (uint64_t)(s[i] - '0');
        if (v > 0xFFFFFFFFull) return 0;
    }
    *out = (uint32_t)v;
    return 1;
}

/* remove_all_extensions (utils.rs:372-387) of the file name */
static size_t stem_of(const char *path, const char **begin) {
    const char *b = strrchr(path, '/');
    b = b ? b + 1 : path;
    size_t n = strlen(b);
    for (;;) { /* Path::extension: text after the last '.', unless the name starts with it */
        size_t dot = n;
        for (size_t i = n; i > 1; --i)
            if (b[i - 1] == '.') {
                dot = i - 1;
                break;
            }
        if (dot == n || dot == 0) break;
        n = dot;
    }
    *begin = b;
    return n;
}

uint64_t orc_fragsplit_tokenize(const orc_index *ix, const char *const *files, uint64_t n_files, const char *const *map_keys,
                                const uint32_t *map_cluster, uint64_t n_map, uint32_t n_clusters, const char *const *chrom_names,
                                uint32_t n_chrom, uint32_t unk_id, uint64_t *out_ids, uint64_t *out_sum, uint64_t *out_barcodes) {
    smap map, chroms;
    smap_init(&map, n_map + 1);
    for (uint64_t i = 0; i < n_map; ++i) smap_put(&map, map_keys[i], strlen(map_keys[i]), map_cluster[i]);
    smap_init(&chroms, n_chrom + 1);
    for (uint32_t c = 0; c < n_chrom; ++c) smap_put(&chroms, chrom_names[c], strlen(chrom_names[c]), c);
    cluster *cl = (cluster *)calloc(n_clusters ? n_clusters : 1, sizeof(cluster));
    for (uint32_t c = 0; c < n_clusters; ++c) smap_init(&cl[c].barcodes, 64);
    uint64_t reads = 0;
    int bad = 0;
    size_t line_cap = 1 << 16, key_cap = 256;
    char *line = (char *)malloc(line_cap), *key = (char *)malloc(key_cap);
    uint32_t hs[64], he[64], hv[64];
    for (uint64_t f = 0; f < n_files && !bad; ++f) {
        gzFile gz = gzopen(files[f], "rb"); /* (reads plain files too) */
        if (!gz) {
            bad = 1;
            break;
        }
        gzbuffer(gz, 1 << 18);
        const char *stem;
        const size_t stem_n = stem_of(files[f], &stem);
        while (gzgets(gz, line, (int)line_cap)) {
            size_t n = strlen(line);
            if (n && line[n - 1] == '\n') --n;
            const char *fld[5];
            size_t fl[5];
            int nf = 0;
            size_t i = 0;
            while (i < n && nf < 5) {
                while (i < n && is_ws(line[i])) ++i;
                const size_t st = i;
                while (i < n && !is_ws(line[i])) ++i;
                if (i > st) {
                    fld[nf] = line + st;
                    fl[nf] = i - st;
                    ++nf;
                }
            }
            if (nf < 5) {
                bad = 1;
                break;
            }
            ++reads;
            if (stem_n + 1 + fl[3] + 1 > key_cap) {
                key_cap = (stem_n + 1 + fl[3] + 1) * 2;
                key = (char *)realloc(key, key_cap);
            }
            memcpy(key, stem, stem_n);
            key[stem_n] = '+';
            memcpy(key + stem_n + 1, fld[3], fl[3]);
            uint32_t c = 0;
            if (!smap_get(&map, key, stem_n + 1 + fl[3], &c) || c >= n_clusters) continue; /* a cell dropped in QC */
            if (fld[0][0] == '#') continue;                                                /* tokenize_fragment_file skips '#' lines */
            uint32_t s = 0, e = 0;
            if (!parse_u32(fld[1], fl[1], &s) || !parse_u32(fld[2], fl[2], &e)) {
                bad = 1;
                break;
            }
            /* the barcode's vector */
            cluster *k = &cl[c];
            uint32_t bi = 0;
            if (!smap_get(&k->barcodes, fld[3], fl[3], &bi)) {
                bi = (uint32_t)k->n_vecs;
                smap_put(&k->barcodes, fld[3], fl[3], bi);
                if (k->n_vecs == k->cap_vecs) {
                    k->cap_vecs = k->cap_vecs ? k->cap_vecs * 2 : 64;
                    k->vecs = (idvec *)realloc(k->vecs, k->cap_vecs * sizeof(idvec));
                }
                memset(&k->vecs[k->n_vecs++], 0, sizeof(idvec));
            }
            idvec *v = &k->vecs[bi];
            /* one single-region tokenize (tokenizer.rs:140-163) */
            uint32_t cid = 0;
            uint64_t h = 0;
            if (smap_get(&chroms, fld[0], fl[0], &cid)) {
                h = orc_find(ix, cid, s, e, hs, he, hv, 64);
                if (h <= 64) {
                    for (uint64_t x = 0; x < h; ++x) idvec_push(v, hv[x]);
                } else { /* (more hits than the stack buffer: once more with room) */
                    uint32_t *bs = (uint32_t *)malloc(h * 4), *be = (uint32_t *)malloc(h * 4), *bv = (uint32_t *)malloc(h * 4);
                    orc_find(ix, cid, s, e, bs, be, bv, h);
                    for (uint64_t x = 0; x < h; ++x) idvec_push(v, bv[x]);
                    free(bs);
                    free(be);
                    free(bv);
                }
            }
            if (h == 0) idvec_push(v, unk_id);
        }
        gzclose(gz);
    }
    for (uint32_t c = 0; c < n_clusters; ++c) {
        uint64_t ids = 0, sum = 0;
        for (uint64_t b = 0; b < cl[c].n_vecs; ++b) {
            ids += cl[c].vecs[b].n;
            for (uint64_t x = 0; x < cl[c].vecs[b].n; ++x) sum += cl[c].vecs[b].p[x];
            free(cl[c].vecs[b].p);
        }
        if (out_ids) out_ids[c] = ids;
        if (out_sum) out_sum[c] = sum;
        if (out_barcodes) out_barcodes[c] = cl[c].n_vecs;
        free(cl[c].vecs);
        smap_free(&cl[c].barcodes);
    }
    free(cl);
    free(line);
    free(key);
    smap_free(&map);
    smap_free(&chroms);
    return bad ? (uint64_t)-1 : reads;
}
