// frag_device.h -- the interface between the host layer (host.cpp, plain C++) and the device side of the fused fragment pipeline
// (fragparse.hip): the inflated TEXT of a wave of fragment files goes to the GPU, which splits it into lines, parses the five
// fields, resolves chromosome and barcode, groups the routed fragments by cluster and tokenizes them -- what
// gtars-fragsplit/src/split.rs:84-131 and gtars-tokenizers/src/utils/fragments.rs:12-56 do line by line on one thread.
// No HIP types here.
#pragma once

#include <cstdint>
#include <cstdlib>
#include <memory>
#include <string>
#include <vector>

#include "../../include/gtars_amd_host.h"

namespace gtars {

// Host memory the GPU's copy engines read and write directly (pinned), from a process-wide pool (round 5).  The fused pipeline
// moves ~45 bytes of text per fragment to the device and ~12 back; from ordinary (pageable) memory the runtime stages every copy
// through its own bounce buffers -- 11 to 37 ms for config 5's 216 MB of text, the longest stage of the call once the inflate was
// fast, and 2 to 3 API calls per file on top (1000 small files: 30 to 40 ms).  The host threads therefore inflate straight INTO
// pinned blocks, and the results come back into pinned blocks.  Pinning costs (the driver maps and locks the pages), so blocks are
// cached when released -- up to GTARS_PINNED_POOL_MB (default 4096) -- and a warm call allocates nothing (at most
// GTARS_PINNED_MAX_MB, default 16384, are handed out at once: beyond that callers get ordinary memory); a pooled block is also
// already faulted in, which an ordinary 4.5-MB allocation is not (first touch: a page fault per 4 KiB).
// acquire: a block of at least `bytes` (*capacity = its size), or nullptr when pinned memory is not to be had (the caller then
// uses ordinary memory: slower copies, same results).
void *frag_pinned_acquire(size_t bytes, size_t *capacity);
void frag_pinned_release(void *p, size_t capacity);

// a block of host memory for device transfers: pinned when the pool has it, malloc'ed otherwise; move-only
struct HostBlock {
    void *p = nullptr;
    size_t cap = 0;
    bool pinned = false;
    HostBlock() = default;
    HostBlock(const HostBlock &) = delete;
    HostBlock &operator=(const HostBlock &) = delete;
    HostBlock(HostBlock &&o) noexcept : p(o.p), cap(o.cap), pinned(o.pinned) { o.p = nullptr, o.cap = 0; }
    HostBlock &operator=(HostBlock &&o) noexcept {
        if (this != &o) {
            reset();
            p = o.p, cap = o.cap, pinned = o.pinned;
            o.p = nullptr, o.cap = 0;
        }
        return *this;
    }
    ~HostBlock() { reset(); }
    void reset() {
        if (p) {
            if (pinned)
                frag_pinned_release(p, cap);
            else
                free(p);
        }
        p = nullptr, cap = 0;
    }
    bool alloc(size_t bytes, bool want_pinned = true) {  // (contents undefined; false: out of memory)
        reset();
        if (want_pinned && (p = frag_pinned_acquire(bytes, &cap))) {
            pinned = true;
            return true;
        }
        pinned = false;
        cap = bytes ? bytes : 1;
        p = malloc(cap);
        if (!p) cap = 0;
        return p != nullptr;
    }
};
template <class T>
struct HostArray {
    HostBlock b;
    bool alloc(size_t n, bool want_pinned = true) { return b.alloc((n ? n : 1) * sizeof(T), want_pinned); }
    void reset() { b.reset(); }
    T *get() const { return (T *)b.p; }
    T &operator[](size_t i) const { return ((T *)b.p)[i]; }
    explicit operator bool() const { return b.p != nullptr; }
};

// 32-bit FNV-1a with a final mix: the ONE hash of the open-addressing tables both sides use
inline uint32_t frag_hash(const char *p, uint32_t n) {
    uint32_t h = 2166136261u;
    for (uint32_t i = 0; i < n; ++i) h = (h ^ (unsigned char)p[i]) * 16777619u;
    return h ^ (h >> 15);
}

// a slot of an open-addressing string table (power-of-two slot count, linear probing): len == 0 is an empty slot
struct FragSlot {
    uint32_t off, len;  // the key's bytes in the table's blob
    uint32_t value;     // cluster index (barcode tables) / chromosome id (chromosome table)
    uint32_t pad;
};

// One gzip member of a file whose CRC-32 the DEVICE checks (round 5): the host thread inflated the member's deflate stream raw --
// zlib's crc32 is a quarter of its inflate time, and the inflated bytes go to the GPU anyway -- and hands over where the member's
// bytes lie in the file's text and what its trailer says.
struct FragGzMember {
    uint64_t off, len;  // bytes [off, off + len) of the file's text
    uint32_t crc;       // CRC-32 of the trailer (RFC 1952)
};

// one fragment file of a wave, inflated, as the host hands it over
struct FragFileIn {
    const char *text = nullptr;  // the whole file; ends with '\n' unless empty (the caller appends one when the file lacks it)
    uint64_t n = 0;
    // the file's barcode table: the map keys that start with "{stem}+", keyed by the barcode alone (split.rs:100-106)
    const FragSlot *slots = nullptr;
    uint32_t n_slots = 0;  // a power of two (>= 1)
    const char *keys = nullptr;
    uint32_t n_key_bytes = 0;
    // gzip members whose CRC-32 is still to be checked (empty: the host has verified the file, or it is not gzip).  NOTE: the
    // newline the caller may append behind the last line is not part of any member.
    const FragGzMember *members = nullptr;
    uint32_t n_members = 0;
};

// What comes back for a wave: the token ids of its routed, tokenized fragments REGROUPED BY (file, barcode) on the device (round 5,
// last cut; fragparse.hip k_frag_emit).  The fragments are sorted by (file, slot of the barcode in the file's table), stable, so
// every (file, barcode) is one run of fragments in line order, and `ids` holds the runs' ids one after the other -- a fragment
// without hits contributes the unk id (fragments.rs:35-56).  A (file, barcode) belongs to one cluster (the slot's value), so the
// host is left with: per cluster, its runs in the order their first lines appear (= the first-seen barcode order of a pass over
// the cluster's file), a dictionary lookup per run, one memcpy per run.
struct FragWaveOut {
    uint64_t n = 0;      // routed fragments that are tokenized (lines whose chromosome field starts with '#' are not)
    uint64_t n_ids = 0;  // their ids, the unk fills included
    HostArray<uint32_t> ids;        // [n_ids]
    std::vector<uint32_t> slot_off;  // [n_files + 1]: file f's slots are [slot_off[f], slot_off[f + 1]) of the two arrays below
    HostArray<uint32_t> run_start;  // [slots]: first id of the slot's run in `ids`, 0xFFFFFFFF: no tokenized fragment has this barcode;
                                    // the runs lie in slot order, so a run ends where the next present one starts (the last: n_ids)
    HostArray<uint32_t> run_line;   // [slots]: the wave-wide number of the line that opens the run (lines count through the files in order)
    std::vector<uint64_t> n_reads, n_written;  // per file: lines, routed lines ('#' lines included)
    int64_t first_error_file = -1;     // first file (wave order) with a line the reference fails on, or with a gzip member whose
                                       // CRC-32 is not its trailer's; -1: none.  The caller reads that file again on the host (zlib's
                                       // own check, the host parser) for the reference's message.
    double t_h2d = 0, t_parse = 0, t_group = 0, t_tok = 0, t_d2h = 0;  // seconds (GTARS_HOST_TIMING)
};

// the tokenizer's chromosome dictionary as a device table; created once per pipeline call
struct FragChroms;
gtars_status frag_chroms_create(const std::vector<std::string> &names, FragChroms **out);
void frag_chroms_free(FragChroms *c);

// One wave on the calling thread's current device.  The text of all files together must stay below 4 GiB.
// unk_id: the id a fragment without hits contributes.  GTARS_ERR_CAPACITY: the wave yields more than 4e9 ids (the caller parses on the host).
gtars_status frag_wave_device(const gtars_index_t *ix, const FragChroms *chroms, const std::vector<FragFileIn> &files, uint32_t n_clusters,
                              uint32_t unk_id, FragWaveOut &out);
int frag_current_device();  // the calling thread's current HIP device (the wave thread selects the caller's)
gtars_status frag_select_device(int device);

}  // namespace gtars
