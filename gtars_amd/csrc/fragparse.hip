// fragparse.hip -- the fragment TEXT of the fused fragsplit -> tokenizer pipeline on the device (round 5; BASELINE config 5).
//
// Round 4 left config 5 a host benchmark: 48 files x 1e5 fragments took 48 ms on 16 host threads, 42 of them gunzip + parse +
// route, and zlib's inflate is only ~45 % of that.  Here the host threads inflate and nothing else; a wave's inflated text goes
// to the GPU, where
//   k_frag_lines      counts / locates the line ends (one pass each over the text, 64 bytes per lane);
//   k_frag_parse      one lane per line: the five whitespace-separated fields (split.rs:84-98 / fragments.rs:12-33), the
//                     barcode -> cluster lookup in the FILE's own key table (the map keys that start with "{stem}+",
//                     split.rs:100-106), the '#' rule of the cluster file's reader (fragments.rs:61-82), the two integers as
//                     str::parse::<u32> takes them, the chromosome id of the tokenizer's dictionary;
//   a stable radix sort of the lines by (file, barcode) -- the key is the slot of the line's barcode in the wave's concatenated
//   tables (sort.hip; unrouted and '#' lines sort last) -- and a gather of the three query columns in that order;
//   the fused tokenizer (tokenize_lds.hip) runs on those columns where they lie;
//   k_frag_emit       the per-barcode regrouping (HashMap<String, Vec<u32>> of fragments.rs:35-56): sorted as they are, the
//                     fragments of one (file, barcode) are one run in line order; the runs' ids are written one after the other
//                     (the unk id where a fragment has none), with where every run starts and which line opens it;
//   k_crc_*           the gzip members' CRC-32 (the host threads decode the deflate streams raw).
// Back to the host go the regrouped ids and two words per (file, barcode) slot; the host is left with one dictionary lookup and
// one memcpy per run (host.cpp).
// A line the reference would fail on is only DETECTED here (first file in wave order); the host re-parses that file for the
// reference's message.  Bound: PCIe for the text in (45 bytes per fragment), then HBM; no MFMA.
#include "common.h"
#include "frag_device.h"
#include "scan.h"

#include <algorithm>
#include <chrono>
#include <cstring>
#include <mutex>

namespace gtars {

namespace {

constexpr int FP_TPB = 256;
constexpr u32 FP_BYTES = 64;                  // text bytes per lane
constexpr u32 FP_CHUNK = FP_TPB * FP_BYTES;   // ... per workgroup
#ifndef GTARS_FRAG_PARSE_GLOBAL
#define GTARS_FRAG_PARSE_GLOBAL 0  // 1: k_frag_parse reads its lines in global memory (A/B build)
#endif
constexpr u32 NO_CLUSTER = 0xFFFFu;           // sort key of a line that is not tokenized (clusters are < 65535)

// 0x80 in every byte of v that equals '\n'
__device__ __forceinline__ u32 newline_bytes(u32 v) {
    v ^= 0x0A0A0A0Au;
    return ~(((v & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | v | 0x7F7F7F7Fu);
}

// line ends: COUNT per workgroup chunk (FILL = false), or their positions at chunk_base[chunk] + rank (FILL = true)
template <bool FILL>
__global__ void __launch_bounds__(FP_TPB)
k_frag_lines(const u32 *__restrict__ text32, u32 n_bytes, u32 *__restrict__ chunk_cnt, const u32 *__restrict__ chunk_base,
             u32 *__restrict__ line_end) {
    __shared__ u32 s_scan[FP_TPB / 64];
    const u32 b0 = blockIdx.x * FP_CHUNK + threadIdx.x * FP_BYTES;  // (the buffer is padded with zero bytes to whole chunks)
    const uint4 *p = reinterpret_cast<const uint4 *>(text32 + b0 / 4);
    uint4 w[FP_BYTES / 16];
#pragma unroll
    for (u32 k = 0; k < FP_BYTES / 16; ++k) w[k] = b0 < n_bytes ? p[k] : make_uint4(0, 0, 0, 0);
    u32 cnt = 0;
#pragma unroll
    for (u32 k = 0; k < FP_BYTES / 16; ++k)
        cnt += __popc(newline_bytes(w[k].x)) + __popc(newline_bytes(w[k].y)) + __popc(newline_bytes(w[k].z)) + __popc(newline_bytes(w[k].w));
    u32 total;
    const u32 ex = block_exclusive_scan<FP_TPB>(cnt, s_scan, total);
    if (!FILL) {
        if (threadIdx.x == 0) chunk_cnt[blockIdx.x] = total;
        return;
    }
    u32 at = chunk_base[blockIdx.x] + ex;
#pragma unroll
    for (u32 k = 0; k < FP_BYTES / 16; ++k) {
        const u32 ww[4] = {w[k].x, w[k].y, w[k].z, w[k].w};
#pragma unroll
        for (u32 j = 0; j < 4; ++j) {
            u32 m = newline_bytes(ww[j]);
            while (m) {
                const u32 byte = (u32)(__ffs((int)m) - 1) >> 3;
                m &= m - 1;
                line_end[at++] = b0 + k * 16u + j * 4u + byte;
            }
        }
    }
}

// exclusive scan of the chunk counts (one workgroup), total behind the last entry
__global__ void __launch_bounds__(1024)
k_frag_scan_chunks(const u32 *__restrict__ cnt, u32 n, u32 *__restrict__ base) {
    __shared__ u32 s_scan[16];
    __shared__ u32 s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (u32 b0 = 0; b0 < n; b0 += 1024) {
        const u32 i = b0 + threadIdx.x;
        const u32 v = i < n ? cnt[i] : 0u;
        u32 total;
        const u32 ex = block_exclusive_scan<1024>(v, s_scan, total);
        const u32 carry = s_carry;
        if (i < n) base[i] = carry + ex;
        __syncthreads();
        if (threadIdx.x == 0) s_carry = carry + total;
        __syncthreads();
    }
    if (threadIdx.x == 0) base[n] = s_carry;
}

// first line of every file: the number of line ends in front of the file's first byte (files end with '\n')
__global__ void k_frag_file_lines(const u32 *__restrict__ line_end, u32 n_lines, const u32 *__restrict__ file_off, u32 n_files,
                                  u32 *__restrict__ file_line) {
    const u32 f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f > n_files) return;
    const u32 x = file_off[f];
    u32 lo = 0, hi = n_lines;
    while (lo < hi) {
        const u32 mid = lo + ((hi - lo) >> 1);
        if (line_end[mid] < x)
            lo = mid + 1;
        else
            hi = mid;
    }
    file_line[f] = lo;
}

__device__ __forceinline__ bool dev_is_ws(unsigned char ch) { return ch == ' ' || (ch >= '\t' && ch <= '\r'); }  // isspace(), C locale

// The text is read through a READER rd(p) = byte p: the workgroup's piece of the text staged in LDS (ds_read_u8), or global memory.
// (Not one generic pointer for both: flat loads that land in LDS cost several times what the LDS instructions do.)
typedef u32 u32_a1 __attribute__((aligned(1)));  // (a 4-byte load at any address: gfx950 takes unaligned global and LDS accesses)
struct GlobalBytes {
    const unsigned char *t;
    __device__ __forceinline__ unsigned char operator()(u32 p) const { return t[p]; }
    __device__ __forceinline__ u32 word(u32 p) const { return *reinterpret_cast<const u32_a1 *>(t + p); }
};
struct LdsBytes {
    const unsigned char *t;  // (a __shared__ array: the compiler keeps the address space through the inlined calls)
    u32 rel;
    __device__ __forceinline__ unsigned char operator()(u32 p) const { return t[p - rel]; }
    __device__ __forceinline__ u32 word(u32 p) const { return *reinterpret_cast<const u32_a1 *>(t + (p - rel)); }
};

// str::parse::<u32>(): optional '+', ASCII digits, must fit
template <class R>
__device__ __forceinline__ bool dev_parse_u32(const R &rd, u32 p, u32 n, u32 &out) {
    u32 i = (n && rd(p) == '+') ? 1u : 0u;
    if (i >= n) return false;
    u64 v = 0;
    for (; i < n; ++i) {
        const u32 d = (u32)rd(p + i) - '0';
        if (d > 9) return false;
        v = v * 10 + d;
        if (v > 0xFFFFFFFFull) return false;
    }
    out = (u32)v;
    return true;
}

template <class R>
__device__ __forceinline__ u32 dev_hash(const R &rd, u32 p, u32 n) {  // frag_hash (frag_device.h)
    u32 h = 2166136261u;
    for (u32 i = 0; i < n; ++i) h = (h ^ rd(p + i)) * 16777619u;
    return h ^ (h >> 15);
}

// slot of key [p, p + n) of the text in an open-addressing table, or 0xFFFFFFFF
template <class R>
__device__ __forceinline__ u32 table_find(const FragSlot *__restrict__ slots, u32 n_slots, const unsigned char *__restrict__ blob, const R &rd,
                                          u32 p, u32 n) {
    if (!n) return 0xFFFFFFFFu;
    const u32 mask = n_slots - 1u;
    u32 k = dev_hash(rd, p, n) & mask;
    for (u32 probes = 0; probes < n_slots; ++probes) {
        const FragSlot s = slots[k];
        if (!s.len) return 0xFFFFFFFFu;
        if (s.len == n) {
            // four bytes at a time (both sides are readable a word past the key: the blob and the text are padded); a byte-wise
            // compare is a chain of dependent L2 loads, one per byte of the barcode
            const unsigned char *q = blob + s.off;
            bool same = true;
            for (u32 i = 0; same && i < n; i += 4) {
                const u32 left = n - i, m = left >= 4 ? 0xFFFFFFFFu : (1u << (8u * left)) - 1u;
                same = ((*reinterpret_cast<const u32_a1 *>(q + i) ^ rd.word(p + i)) & m) == 0u;
            }
            if (same) return k;
        }
        k = (k + 1u) & mask;
    }
    return 0xFFFFFFFFu;
}

struct FragTables {
    const FragSlot *slots;        // all files' tables, one after the other
    const unsigned char *keys;    // ... and their key blobs
    const u32 *slot_off;          // [n_files + 1]
    const u32 *key_off;           // [n_files]
    const FragSlot *chrom_slots;  // the tokenizer's chromosome dictionary
    const unsigned char *chrom_keys;
    u32 n_chrom_slots;
};

// One lane per line.  key[i] = the line's (file, barcode) -- the slot of its barcode in the wave's concatenated tables -- when it
// is routed AND tokenized, `no_key` (= the number of slots: sorts behind every real key) otherwise; counts per file.
__global__ void __launch_bounds__(FP_TPB)
k_frag_parse(const unsigned char *__restrict__ text, const u32 *__restrict__ line_end, u32 n_lines, const u32 *__restrict__ file_line,
             u32 n_files, FragTables tb, u32 no_key, u32 *__restrict__ key, u32 *__restrict__ q_chrom, u32 *__restrict__ q_start,
             u32 *__restrict__ q_end, u32 *__restrict__ n_written, u32 *__restrict__ wg_written, u32 *__restrict__ err_file) {
    // The workgroup's 256 lines are one contiguous piece of the text (~11 KB): it is brought into LDS by 16-byte loads, every lane
    // 16 consecutive bytes, and the lanes parse their lines THERE -- a lane that walks its line byte by byte in global memory
    // makes every byte a wave instruction over 64 different cache lines (310 us per 0.7M lines before: 100 GB/s of text).  A piece
    // larger than the buffer (very long lines) is parsed where it lies.  The pointer `tp` is a generic one.
    constexpr u32 PARSE_LDS = 32u << 10;
    __shared__ uint4 s_text4[PARSE_LDS / 16];
    const u32 first = blockIdx.x * FP_TPB, last = min(first + (u32)FP_TPB, n_lines) - 1u;  // (the grid covers n_lines: first <= last)
    const u32 span_lo = first ? line_end[first - 1] + 1u : 0u, span_hi = line_end[last] + 1u, base = span_lo & ~15u;
    const bool staged = span_hi - base <= PARSE_LDS && !(GTARS_FRAG_PARSE_GLOBAL);  // (uniform)
    if (staged) {
        for (u32 o = threadIdx.x * 16u; o < span_hi - base; o += FP_TPB * 16u)  // (the text buffer is padded: a whole last vector)
            s_text4[o >> 4] = *reinterpret_cast<const uint4 *>(text + base + o);
        __syncthreads();
    }
    __shared__ u32 s_written;
    if (threadIdx.x == 0) s_written = 0;
    const u32 i = blockIdx.x * FP_TPB + threadIdx.x;
    const bool valid = i < n_lines;
    const u32 il = valid ? i : n_lines - 1u;  // (lanes behind the last line read its bounds and do nothing)
    const u32 lo = il ? line_end[il - 1] + 1u : 0u, hi = valid ? line_end[il] : lo;
    // the line's file: last f with file_line[f] <= i -- for the workgroup's first line (uniform: scalar loads), which is every
    // line's file unless a file ends inside the workgroup's lines
    auto file_of = [&](u32 line) {
        u32 a = 0, b = n_files;
        while (a + 1 < b) {
            const u32 mid = (a + b) >> 1;
            if (file_line[mid] <= line)
                a = mid;
            else
                b = mid;
        }
        return a;
    };
    u32 f = file_of(first);
    const bool one_file = f + 1 >= n_files || file_line[f + 1] > last;  // (uniform)
    if (!one_file) f = file_of(il);
    u32 k = no_key;
    bool bad = false, written = false;
    auto parse = [&](const auto &rd) {
        u32 fs[5], fl[5], nf = 0, p = lo;
        while (p < hi && nf < 5) {
            while (p < hi && dev_is_ws(rd(p))) ++p;
            const u32 st = p;
            while (p < hi && !dev_is_ws(rd(p))) ++p;
            if (p > st) {
                fs[nf] = st;
                fl[nf] = p - st;
                ++nf;
            }
        }
        bad = nf < 5;  // fewer than five fields: "Failed to parse fragments file at line ..." (routed or not)
        if (bad) return;
        const u32 so = tb.slot_off[f], ns = tb.slot_off[f + 1] - so;
        const u32 slot = table_find(tb.slots + so, ns, tb.keys + tb.key_off[f], rd, fs[3], fl[3]);
        if (slot == 0xFFFFFFFFu) return;  // most likely a cell dropped in QC -- nothing else of the line is looked at
        written = true;
        if (rd(fs[0]) == '#') return;  // (the cluster file's reader skips '#' lines: fragments.rs:70-73)
        u32 s = 0, e = 0;
        if (!dev_parse_u32(rd, fs[1], fl[1], s) || !dev_parse_u32(rd, fs[2], fl[2], e)) {
            bad = true;
            return;
        }
        const u32 cs = table_find(tb.chrom_slots, tb.n_chrom_slots, tb.chrom_keys, rd, fs[0], fl[0]);
        k = so + slot;
        q_chrom[i] = cs == 0xFFFFFFFFu ? GTARS_UNKNOWN_CHROM : tb.chrom_slots[cs].value;
        q_start[i] = s;
        q_end[i] = e;
    };
    if (valid) {
        if (staged)
            parse(LdsBytes{reinterpret_cast<const unsigned char *>(s_text4), base});
        else
            parse(GlobalBytes{text});
    }
    if (valid) key[i] = k;
    if (bad) atomicMin(err_file, f);
    // Routed lines per file.  A workgroup whose lines share a file (nearly all do) leaves ONE number, in its own word -- summed per
    // file by k_frag_sum_written; one atomic per wave on the file's counter was 12 500 atomics per batch on two or three addresses,
    // which the L2 serves one after the other: 170 of the kernel's 173 us.
    if (one_file) {
        const u64 m = __ballot(written);
        __syncthreads();  // s_written is zero
        if (m && (threadIdx.x & 63) == 0) atomicAdd(&s_written, (u32)__popcll(m));
        __syncthreads();
        if (threadIdx.x == 0) wg_written[blockIdx.x] = s_written;
    } else {
        if (threadIdx.x == 0) wg_written[blockIdx.x] = 0;
        if (written) atomicAdd(&n_written[f], 1u);
    }
}

// n_written[f] += the counts of the workgroups of k_frag_parse whose first line lies in file f (a workgroup that spans files left 0
// and counted by atomics)
__global__ void __launch_bounds__(256)
k_frag_sum_written(const u32 *__restrict__ wg_written, const u32 *__restrict__ file_line, u32 n_files, u32 *__restrict__ n_written) {
    __shared__ u32 s_scan[4];
    const u32 f = blockIdx.x;
    if (f >= n_files) return;
    const u32 w0 = (file_line[f] + FP_TPB - 1) / FP_TPB, w1 = (file_line[f + 1] + FP_TPB - 1) / FP_TPB;
    u32 sum = 0;
    for (u32 w = w0 + threadIdx.x; w < w1; w += 256) sum += wg_written[w];
    u32 total;
    (void)block_exclusive_scan<256>(sum, s_scan, total);
    if (threadIdx.x == 0 && total) atomicAdd(&n_written[f], total);
}

__global__ void k_frag_iota(u32 *__restrict__ p, u32 n) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = i;
}

// *out = first position of the sorted keys with key >= bound (= the number of tokenized fragments for bound = no_key)
__global__ void k_frag_lower_bound(const u32 *__restrict__ sorted_key, u32 n, u32 bound, u32 *__restrict__ out) {
    if (blockIdx.x || threadIdx.x) return;
    u32 lo = 0, hi = n;
    while (lo < hi) {
        const u32 mid = lo + ((hi - lo) >> 1);
        if (sorted_key[mid] < bound)
            lo = mid + 1;
        else
            hi = mid;
    }
    *out = lo;
}

// columns of the tokenized fragments in (file, barcode) order: line perm[j] of the wave
__global__ void k_frag_gather(const u32 *__restrict__ perm, u32 n, const u32 *__restrict__ q_chrom, const u32 *__restrict__ q_start,
                              const u32 *__restrict__ q_end, u32 *__restrict__ oc, u32 *__restrict__ os, u32 *__restrict__ oe) {
    const u32 j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const u32 i = perm[j];
    oc[j] = q_chrom[i];
    os[j] = q_start[i];
    oe[j] = q_end[i];
}

// ---- the per-barcode regrouping on the device (round 5, last cut) --------------------------------------------------------------
// tokenize_fragment_file groups a cluster file's token ids by barcode (fragments.rs:35-56: HashMap<String, Vec<u32>>; a fragment
// without hits contributes the unk id).  The fragments arrive here SORTED by (file, barcode slot) -- stable, so a barcode's
// fragments keep their line order -- which makes every (file, barcode) one RUN of consecutive fragments, and the ids of a run,
// concatenated, are that barcode's share of the result from this file.  k_frag_emit writes the runs' ids one after the other
// (with the unk fill), and per slot where its run starts and which line opened it; the host is left with one memcpy per run and
// a dictionary lookup per (file, barcode) instead of three passes over every fragment.
constexpr u32 EM_TPB = 1024;
// ids a chunk of EM_TPB fragments emits
__global__ void __launch_bounds__(EM_TPB)
k_frag_emit_counts(const u64 *__restrict__ off, u32 n, u32 *__restrict__ chunk_tot) {
    __shared__ u32 s_scan[16];
    const u32 j = blockIdx.x * EM_TPB + threadIdx.x;
    u32 c = 0;
    if (j < n) {
        const u64 h = off[j + 1] - off[j];
        c = h ? (u32)h : 1u;
    }
    u32 total;
    (void)block_exclusive_scan<EM_TPB>(c, s_scan, total);
    if (threadIdx.x == 0) chunk_tot[blockIdx.x] = total;
}
__global__ void __launch_bounds__(EM_TPB)
k_frag_emit(const u64 *__restrict__ off, const u32 *__restrict__ ids, u32 n, const u32 *__restrict__ chunk_base, const u32 *__restrict__ sorted_key,
            const u32 *__restrict__ perm, u32 unk_id, u32 *__restrict__ out_ids, u32 *__restrict__ run_start, u32 *__restrict__ run_line) {
    __shared__ u32 s_scan[16];
    const u32 j = blockIdx.x * EM_TPB + threadIdx.x;
    u64 o = 0;
    u32 h = 0;
    if (j < n) {
        o = off[j];
        h = (u32)(off[j + 1] - o);
    }
    u32 total;
    const u32 dst = chunk_base[blockIdx.x] + block_exclusive_scan<EM_TPB>(j < n ? (h ? h : 1u) : 0u, s_scan, total);
    if (j >= n) return;
    const u32 g = sorted_key[j];
    if (j == 0 || sorted_key[j - 1] != g) {  // the fragment that opens its (file, barcode)'s run
        run_start[g] = dst;
        run_line[g] = perm[j];
    }
    if (!h) {
        out_ids[dst] = unk_id;
    } else {
        for (u32 k = 0; k < h; ++k) out_ids[dst + k] = ids[o + k];
    }
}

// ---- CRC-32 (RFC 1952: reflected polynomial 0xEDB88320) of the gzip members on the device ----------------------------------
// The register after processing bytes B from register s is  shift_|B|(s) ^ f(0, B)  (the CRC is linear over GF(2)), shift_n =
// "process n zero bytes".  So: (1) one lane per 512-byte chunk forms f(0, chunk) byte by byte (table of 256 words in LDS);
// (2) one lane per group of 64 chunks folds them left to right with shift_512 (four tables of 256 words: one per byte of s);
// (3) one lane per member folds its groups with shift_32768, starting from 0xFFFFFFFF, and compares the complement with the
// trailer's CRC.  The last chunk / group of a member is short: its shift is applied 512 bytes, then a byte at a time.
constexpr u32 CRC_CHUNK = 512, CRC_GROUP = 64;
struct CrcTables {
    u32 byte[256];         // one byte of data
    u32 s512[4][256];      // shift by 512 zero bytes, per byte of the register
    u32 s32k[4][256];      // shift by 512 * 64 zero bytes
};
__device__ __forceinline__ u32 crc_shift(const u32 (*t)[256], u32 s) {
    return t[0][s & 0xFFu] ^ t[1][(s >> 8) & 0xFFu] ^ t[2][(s >> 16) & 0xFFu] ^ t[3][s >> 24];
}
__device__ __forceinline__ u32 crc_zero_bytes(const u32 *byte_tab, u32 s, u32 n) {  // n < 512 zero bytes, one at a time
    for (u32 i = 0; i < n; ++i) s = byte_tab[s & 0xFFu] ^ (s >> 8);
    return s;
}
struct CrcMember {
    u32 off, len, crc, file;   // text bytes [off, off + len), trailer CRC, file of the wave
    u32 chunk0, group0;        // first chunk / group of the member in the flattened arrays
};
// (1) f(0, chunk) for every chunk of every member
__global__ void __launch_bounds__(256)
k_crc_chunks(const unsigned char *__restrict__ text, const CrcMember *__restrict__ mem, u32 n_mem, u32 n_chunks, const CrcTables *__restrict__ tb,
             u32 *__restrict__ part) {
    __shared__ u32 s_t[256];
    s_t[threadIdx.x] = tb->byte[threadIdx.x];
    __syncthreads();
    const u32 c = blockIdx.x * 256u + threadIdx.x;
    if (c >= n_chunks) return;
    u32 lo = 0, hi = n_mem;  // the chunk's member: last m with chunk0 <= c
    while (lo + 1 < hi) {
        const u32 mid = (lo + hi) >> 1;
        if (mem[mid].chunk0 <= c)
            lo = mid;
        else
            hi = mid;
    }
    const CrcMember m = mem[lo];
    const u32 b0 = (c - m.chunk0) * CRC_CHUNK, n = min(CRC_CHUNK, m.len - b0);
    const unsigned char *p = text + m.off + b0;
    u32 s = 0;
    u32 i = 0;
    for (; i + 4 <= n && ((uintptr_t)(p + i) & 3u) != 0; ++i) s = s_t[(s ^ p[i]) & 0xFFu] ^ (s >> 8);
    for (; i + 4 <= n; i += 4) {
        u32 w = *reinterpret_cast<const u32 *>(p + i);
        s ^= w;
        s = s_t[s & 0xFFu] ^ (s >> 8);
        s = s_t[s & 0xFFu] ^ (s >> 8);
        s = s_t[s & 0xFFu] ^ (s >> 8);
        s = s_t[s & 0xFFu] ^ (s >> 8);
    }
    for (; i < n; ++i) s = s_t[(s ^ p[i]) & 0xFFu] ^ (s >> 8);
    part[c] = s;
}
// the three tables in LDS: the folds below are chains of dependent table lookups (64 / ~140 steps a lane), an L2 round trip each
// when the tables lie in global memory
__device__ __forceinline__ const CrcTables *crc_tables_to_lds(const CrcTables *__restrict__ tb, CrcTables *s_tb) {
    const u32 *src = reinterpret_cast<const u32 *>(tb);
    u32 *dst = reinterpret_cast<u32 *>(s_tb);
    for (u32 w = threadIdx.x; w < sizeof(CrcTables) / 4; w += blockDim.x) dst[w] = src[w];
    __syncthreads();
    return s_tb;
}
// (2) f(0, group) for every group of <= 64 chunks of a member
__global__ void k_crc_groups(const CrcMember *__restrict__ mem, u32 n_mem, u32 n_groups, const CrcTables *__restrict__ tb_g, const u32 *__restrict__ part,
                             u32 *__restrict__ gpart) {
    __shared__ CrcTables s_tb;
    const CrcTables *tb = crc_tables_to_lds(tb_g, &s_tb);
    const u32 g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_groups) return;
    u32 lo = 0, hi = n_mem;
    while (lo + 1 < hi) {
        const u32 mid = (lo + hi) >> 1;
        if (mem[mid].group0 <= g)
            lo = mid;
        else
            hi = mid;
    }
    const CrcMember m = mem[lo];
    const u32 n_ch = (m.len + CRC_CHUNK - 1) / CRC_CHUNK, c0 = (g - m.group0) * CRC_GROUP, c1 = min(n_ch, c0 + CRC_GROUP);
    u32 s = 0;
    for (u32 c = c0; c < c1; ++c) {
        const u32 n = min(CRC_CHUNK, m.len - c * CRC_CHUNK);
        s = (n == CRC_CHUNK ? crc_shift(tb->s512, s) : crc_zero_bytes(tb->byte, s, n)) ^ part[m.chunk0 + c];
    }
    gpart[g] = s;
}
// (3) the member's CRC against its trailer; the first file (wave order) with a mismatch is noted
__global__ void k_crc_members(const CrcMember *__restrict__ mem, u32 n_mem, const CrcTables *__restrict__ tb_g, const u32 *__restrict__ gpart,
                              u32 *__restrict__ err_file) {
    __shared__ CrcTables s_tb;
    const CrcTables *tb = crc_tables_to_lds(tb_g, &s_tb);
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_mem) return;
    const CrcMember m = mem[i];
    const u32 n_ch = (m.len + CRC_CHUNK - 1) / CRC_CHUNK, n_g = (n_ch + CRC_GROUP - 1) / CRC_GROUP;
    u32 s = 0xFFFFFFFFu;
    for (u32 g = 0; g < n_g; ++g) {
        const u32 bytes = min(CRC_CHUNK * CRC_GROUP, m.len - g * CRC_CHUNK * CRC_GROUP);
        if (bytes == CRC_CHUNK * CRC_GROUP) {
            s = crc_shift(tb->s32k, s);
        } else {
            for (u32 k = 0; k < bytes / CRC_CHUNK; ++k) s = crc_shift(tb->s512, s);
            s = crc_zero_bytes(tb->byte, s, bytes % CRC_CHUNK);
        }
        s ^= gpart[m.group0 + g];
    }
    if ((s ^ 0xFFFFFFFFu) != m.crc) atomicMin(err_file, m.file);
}

// the three tables, built once per process on the host
const CrcTables &crc_tables_host() {
    static const CrcTables t = [] {
        CrcTables x;
        for (u32 i = 0; i < 256; ++i) {
            u32 c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            x.byte[i] = c;
        }
        auto zeros = [&](u32 s, u32 n) {
            for (u32 i = 0; i < n; ++i) s = x.byte[s & 0xFFu] ^ (s >> 8);
            return s;
        };
        for (u32 b = 0; b < 4; ++b)
            for (u32 v = 0; v < 256; ++v) x.s512[b][v] = zeros(v << (8 * b), CRC_CHUNK);
        auto sh512 = [&](u32 s) { return x.s512[0][s & 0xFFu] ^ x.s512[1][(s >> 8) & 0xFFu] ^ x.s512[2][(s >> 16) & 0xFFu] ^ x.s512[3][s >> 24]; };
        for (u32 b = 0; b < 4; ++b)
            for (u32 v = 0; v < 256; ++v) {
                u32 s = v << (8 * b);
                for (u32 k = 0; k < CRC_GROUP; ++k) s = sh512(s);
                x.s32k[b][v] = s;
            }
        return x;
    }();
    return t;
}

struct DevMem {
    void *p = nullptr;
    ~DevMem() {
        if (p) (void)hipFree(p);
    }
    gtars_status alloc(size_t bytes) {
        GT_HIP(hipMalloc(&p, std::max<size_t>(bytes, 256)));
        return GTARS_OK;
    }
    template <class T>
    T *as() const {
        return (T *)p;
    }
};

// pieces of a workspace, 256-byte aligned
struct Carve {
    char *p = nullptr;
    size_t at = 0;
    template <class T>
    T *take(size_t n) {
        at = (at + 255) & ~(size_t)255;
        T *r = reinterpret_cast<T *>(p + at);
        at += n * sizeof(T);
        return r;
    }
};
struct View {
    void *p;
    template <class T>
    T *as() const {
        return (T *)p;
    }
};

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// builds the slots of an open-addressing table over `n` keys (views), values given; cap = power of two >= 2 n
void build_table(const std::vector<std::pair<const char *, u32>> &keys, const std::vector<u32> &values, std::vector<FragSlot> &slots,
                 std::string &blob) {
    size_t cap = 1;
    while (cap < keys.size() * 2 + 1) cap <<= 1;
    slots.assign(cap, FragSlot{0, 0, 0, 0});
    blob.clear();
    for (size_t i = 0; i < keys.size(); ++i) {
        if (!keys[i].second) continue;  // (an empty key can never be a field)
        u32 k = frag_hash(keys[i].first, keys[i].second) & (u32)(cap - 1);
        while (slots[k].len) k = (k + 1) & (u32)(cap - 1);
        slots[k] = FragSlot{(u32)blob.size(), keys[i].second, values[i], 0};
        blob.append(keys[i].first, keys[i].second);
    }
}

}  // namespace

struct FragChroms {
    DevMem slots, keys, crc;  // the chromosome table; the CRC-32 tables (device copy, made with it)
    u32 n_slots = 0;
    int device = 0;
};

gtars_status frag_chroms_create(const std::vector<std::string> &names, FragChroms **out) {
    *out = nullptr;
    gtars_status st = require_device();
    if (st) return st;
    std::vector<std::pair<const char *, u32>> keys;
    std::vector<u32> values;
    for (size_t i = 0; i < names.size(); ++i) {
        keys.emplace_back(names[i].data(), (u32)names[i].size());
        values.push_back((u32)i);
    }
    std::vector<FragSlot> slots;
    std::string blob;
    build_table(keys, values, slots, blob);
    std::unique_ptr<FragChroms> c(new FragChroms());
    GT_HIP(hipGetDevice(&c->device));
    if ((st = c->slots.alloc(slots.size() * sizeof(FragSlot)))) return st;
    if ((st = c->keys.alloc(blob.size() + 16))) return st;
    GT_HIP(hipMemcpy(c->slots.p, slots.data(), slots.size() * sizeof(FragSlot), hipMemcpyHostToDevice));
    if (!blob.empty()) GT_HIP(hipMemcpy(c->keys.p, blob.data(), blob.size(), hipMemcpyHostToDevice));
    c->n_slots = (u32)slots.size();
    if ((st = c->crc.alloc(sizeof(CrcTables)))) return st;
    GT_HIP(hipMemcpy(c->crc.p, &crc_tables_host(), sizeof(CrcTables), hipMemcpyHostToDevice));
    *out = c.release();
    return GTARS_OK;
}
void frag_chroms_free(FragChroms *c) { delete c; }

// ---- the pinned host memory pool (frag_device.h) ----
namespace {
struct PinnedPool {
    std::mutex mu;
    std::vector<std::pair<void *, size_t>> free_blocks;
    size_t cached = 0;
    size_t outstanding = 0;  // bytes of the blocks in callers' hands
    // (read at every call, not once: tests switch them between calls -- gtars_debug_reload_env)
    static size_t max_outstanding() {  // beyond this much pinned memory in use at once, callers get ordinary memory
        const char *e = cfg_get("GTARS_PINNED_MAX_MB");
        return e ? (size_t)std::max(0ll, atoll(e)) << 20 : (size_t)16384 << 20;
    }
    static size_t limit() {
        const char *e = cfg_get("GTARS_PINNED_POOL_MB");
        return e ? (size_t)std::max(0ll, atoll(e)) << 20 : (size_t)4096 << 20;
    }
    static bool off() { return cfg_get("GTARS_NO_PINNED") != nullptr; }  // (A/B and tests: ordinary memory everywhere)
};
PinnedPool &pinned_pool() {
    static PinnedPool *p = new PinnedPool;  // (never destroyed: its blocks must not be freed behind the HIP runtime's own teardown)
    return *p;
}
}  // namespace

void *frag_pinned_acquire(size_t bytes, size_t *capacity) {
    PinnedPool &pp = pinned_pool();
    if (pp.off()) return nullptr;
    const size_t need = std::max<size_t>(bytes, 4096);
    {
        std::lock_guard<std::mutex> lk(pp.mu);
        size_t best = (size_t)-1;
        for (size_t i = 0; i < pp.free_blocks.size(); ++i) {
            const size_t c = pp.free_blocks[i].second;
            // (a block far larger than the request stays for the large requests: text blocks for texts, small ones for counters)
            if (c >= need && c <= 2 * need + (1u << 20) && (best == (size_t)-1 || c < pp.free_blocks[best].second)) best = i;
        }
        if (best != (size_t)-1) {
            void *p = pp.free_blocks[best].first;
            *capacity = pp.free_blocks[best].second;
            pp.cached -= *capacity;
            pp.outstanding += *capacity;
            pp.free_blocks[best] = pp.free_blocks.back();
            pp.free_blocks.pop_back();
            return p;
        }
        if (pp.outstanding + need > pp.max_outstanding()) return nullptr;  // (a folder of thousands of files holds its results until the end)
    }
    // a little more than asked: the next file of a folder is about as large as this one, not exactly as large
    const size_t cap = (need + need / 8 + (256u << 10) - 1) & ~(size_t)((256u << 10) - 1);
    void *p = nullptr;
    if (hipHostMalloc(&p, cap, hipHostMallocPortable) != hipSuccess || !p) {
        (void)hipGetLastError();
        return nullptr;
    }
    *capacity = cap;
    {
        std::lock_guard<std::mutex> lk(pp.mu);
        pp.outstanding += cap;
    }
    return p;
}

void frag_pinned_release(void *p, size_t capacity) {
    if (!p) return;
    PinnedPool &pp = pinned_pool();
    {
        std::lock_guard<std::mutex> lk(pp.mu);
        pp.outstanding -= std::min(pp.outstanding, capacity);
        if (pp.cached + capacity <= pp.limit()) {
            pp.free_blocks.emplace_back(p, capacity);
            pp.cached += capacity;
            return;
        }
    }
    (void)hipHostFree(p);
}

int frag_current_device() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return dev;
}

gtars_status frag_select_device(int device) {
    GT_HIP(hipSetDevice(device));
    return GTARS_OK;
}

// Wait for the stream WITHOUT spinning: the thread that drives the device shares the host's cores with the threads that inflate the
// next files -- hipStreamSynchronize polls, i.e. it takes one of 16 cores away from 16 loaders, and 48 files then need four rounds on
// some core instead of three (12.8 ms instead of 9.6).  An event created with hipEventBlockingSync puts the thread to sleep until the
// GPU's interrupt (tens of microseconds later than a poll would notice: nothing here is that urgent).
static gtars_status wait_stream(hipStream_t st) {
    struct Event {  // (one per thread and device it waits on; destroyed with the thread)
        hipEvent_t e[16] = {};
        ~Event() {
            for (hipEvent_t x : e)
                if (x) (void)hipEventDestroy(x);
        }
    };
    static thread_local Event tl;
    if (cfg_get("GTARS_FRAG_SPIN_WAIT")) {  // (A/B)
        GT_HIP(hipStreamSynchronize(st));
        return GTARS_OK;
    }
    int dev = 0;
    GT_HIP(hipGetDevice(&dev));
    hipEvent_t &ev = tl.e[dev & 15];
    if (!ev) GT_HIP(hipEventCreateWithFlags(&ev, hipEventBlockingSync | hipEventDisableTiming));
    GT_HIP(hipEventRecord(ev, st));
    GT_HIP(hipEventSynchronize(ev));
    return GTARS_OK;
}
#define GT_WAIT(st_)                                   \
    do {                                               \
        const gtars_status ws_ = wait_stream(st_);     \
        if (ws_) return ws_;                           \
    } while (0)

gtars_status frag_wave_device(const gtars_index_t *ix, const FragChroms *chroms, const std::vector<FragFileIn> &files, uint32_t n_clusters,
                              uint32_t unk_id, FragWaveOut &out) {
    const u32 n_files = (u32)files.size();
    out.n = 0;
    out.n_ids = 0;
    out.slot_off.assign((size_t)n_files + 1, 0);
    out.n_reads.assign(n_files, 0);
    out.n_written.assign(n_files, 0);
    out.first_error_file = -1;
    out.ids.reset();
    if (!n_files) return GTARS_OK;
    if (n_clusters >= NO_CLUSTER || n_files >= 65535) return fail(GTARS_ERR_INVALID_ARG, "fragment wave: too many clusters or files for the device path");
    u64 total = 0, total_slots = 0, total_keys = 0;
    for (const FragFileIn &f : files) {
        if (f.n && f.text[f.n - 1] != '\n') return fail(GTARS_ERR_INTERNAL, "fragment wave: a file's text must end with a newline");
        total += f.n;
        total_slots += f.n_slots;
        total_keys += f.n_key_bytes;
    }
    if (total >= 0xFFFF0000ull || total_slots >= 0xFFFFFFF0ull) return fail(GTARS_ERR_INVALID_ARG, "fragment wave: more than 4 GiB of text");
    const double t0 = now_s();
    // a stream of the calling thread's own (per device): the host pipeline drives the device from two threads, whose batches must
    // overlap -- copies of one with kernels of the other -- and the null stream would serialise them
    hipStream_t st = nullptr;
    {
        struct Streams {  // (the pipeline's device threads live for one call: their streams go with them)
            hipStream_t s[16] = {};
            ~Streams() {
                for (hipStream_t x : s)
                    if (x) (void)hipStreamDestroy(x);
            }
        };
        static thread_local Streams tl;
        int dev = 0;
        GT_HIP(hipGetDevice(&dev));
        hipStream_t &slot = tl.s[dev & 15];
        if (!slot && !cfg_get("GTARS_FRAG_NULL_STREAM")) GT_HIP(hipStreamCreateWithFlags(&slot, hipStreamNonBlocking));
        st = slot;
    }
    // Whatever way this call ends, the stream is idle when it does (round-5 advisor): copies and kernels queued on `st` read and
    // write pinned pool blocks (the files' texts, `staging`, `mailbox`) and pooled workspaces that go back to their pools when
    // this frame unwinds -- an early error return with work still in flight would let a loader thread inflate the next file
    // into a block the DMA engine is still reading.  The success paths have waited already: a second wait on an idle stream
    // returns at once.
    struct DrainStream {
        hipStream_t s;
        ~DrainStream() { (void)hipStreamSynchronize(s); }
    } drain_on_exit{st};
    const u32 n_bytes = (u32)total;
    const u32 n_chunks = (n_bytes + FP_CHUNK - 1) / FP_CHUNK;
    // ---- text + tables to the device ----
    // Device memory comes from three grow-only workspaces of the calling thread (pooled across threads, api.hip tls_workspace):
    // a wave made ~10 allocations of up to 70 MB, and device allocations / frees synchronise.  Carved by `Carve` below.
    auto pad = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t m1 = (size_t)n_files + 1;
    const size_t text_bytes = (size_t)std::max<u32>(n_chunks, 1) * FP_CHUNK + 64;
    const size_t in_bytes = pad(text_bytes) + pad(std::max<u64>(total_slots, 1) * sizeof(FragSlot)) + pad(total_keys + 16) +
                            pad((m1 * 4 + n_files + 1 + n_clusters + 1) * 4) + pad(((size_t)n_chunks + 1) * 2 * 4) + 1024;
    Workspace &ws_in = tls_workspace(5, st);
    gtars_status s = ws_in.reserve(in_bytes);
    if (s) return s;
    Carve cin{(char *)ws_in.ptr, 0};
    View d_text{cin.take<char>(text_bytes)}, d_slots{cin.take<FragSlot>(std::max<u64>(total_slots, 1))}, d_keys{cin.take<char>(total_keys + 16)};
    View d_meta{cin.take<u32>(m1 * 4 + n_files + 1 + n_clusters + 1)}, d_chunks{cin.take<u32>(((size_t)n_chunks + 1) * 2)};
    // The texts go file by file (they lie in pinned blocks: the host threads inflated into them, so every copy is one DMA); the
    // files' barcode tables and the offset arrays are gathered into ONE pinned staging block first -- three copies for the wave
    // instead of two per file and three small ones from pageable memory (1000 files: 2000 calls of 20 us each before).
    // staging: file_off | slot_off | key_off [m1 words each] | slots | keys | gzip members
    size_t n_members = 0;
    for (const FragFileIn &f : files) n_members += f.n_members;
    const size_t stage_slots = pad(3 * m1 * 4), stage_keys = stage_slots + pad(total_slots * sizeof(FragSlot)),
                 stage_mem = stage_keys + pad(total_keys + 16), stage_bytes = stage_mem + pad(n_members * sizeof(CrcMember)) + 256;
    HostBlock staging;
    if (!staging.alloc(stage_bytes)) return fail(GTARS_ERR_INTERNAL, "out of host memory");
    // the small answers (line count, error file, per-cluster and per-file counts) land in a pinned mailbox: a copy into pageable
    // memory is synchronous, and that wait polls
    HostBlock mailbox;
    const size_t mb_coff = 64, mb_written = mb_coff + pad(((size_t)n_clusters + 1) * 4), mb_fline = mb_written + pad((size_t)n_files * 4),
                 mb_cbase = mb_fline + pad(m1 * 4), mb_bytes = mb_cbase + pad(((size_t)n_clusters + 1) * 8);
    if (!mailbox.alloc(mb_bytes)) return fail(GTARS_ERR_INTERNAL, "out of host memory");
    volatile u32 *mb_words = (volatile u32 *)mailbox.p;  // [0] line count, [1] error file, [2] wide flag
    u32 *file_off = (u32 *)staging.p, *slot_off = file_off + m1, *key_off = slot_off + m1;
    file_off[0] = slot_off[0] = key_off[0] = 0;
    for (u32 f = 0; f < n_files; ++f) {
        file_off[f + 1] = file_off[f] + (u32)files[f].n;
        slot_off[f + 1] = slot_off[f] + files[f].n_slots;
        key_off[f + 1] = key_off[f] + files[f].n_key_bytes;
        if (files[f].n) GT_HIP(hipMemcpyAsync(d_text.as<char>() + file_off[f], files[f].text, files[f].n, hipMemcpyHostToDevice, st));
        if (files[f].n_slots) memcpy((char *)staging.p + stage_slots + (size_t)slot_off[f] * sizeof(FragSlot), files[f].slots, (size_t)files[f].n_slots * sizeof(FragSlot));
        if (files[f].n_key_bytes) memcpy((char *)staging.p + stage_keys + key_off[f], files[f].keys, files[f].n_key_bytes);
    }
    for (u32 f = 0; f <= n_files; ++f) out.slot_off[f] = slot_off[f];
    if (total_slots) GT_HIP(hipMemcpyAsync(d_slots.as<FragSlot>(), (char *)staging.p + stage_slots, total_slots * sizeof(FragSlot), hipMemcpyHostToDevice, st));
    if (total_keys) GT_HIP(hipMemcpyAsync(d_keys.as<char>(), (char *)staging.p + stage_keys, total_keys, hipMemcpyHostToDevice, st));
    GT_HIP(hipMemsetAsync(d_text.as<char>() + n_bytes, 0, (size_t)std::max<u32>(n_chunks, 1) * FP_CHUNK + 64 - n_bytes, st));
    // meta: file_off | slot_off | key_off | file_line [n_files + 1 each] | n_written [n_files] | err_file | coff [n_clusters + 1]
    u32 *d_file_off = d_meta.as<u32>(), *d_slot_off = d_file_off + m1, *d_key_off = d_slot_off + m1, *d_file_line = d_key_off + m1;
    u32 *d_written = d_file_line + m1, *d_err = d_written + n_files, *d_coff = d_err + 1;
    GT_HIP(hipMemcpyAsync(d_file_off, file_off, 3 * m1 * 4, hipMemcpyHostToDevice, st));
    GT_HIP(hipMemsetAsync(d_written, 0, (size_t)n_files * 4, st));
    GT_HIP(hipMemsetAsync(d_err, 0xFF, 4, st));
    // ---- the gzip members' CRC-32 (the host threads inflated them raw) ----
    CrcMember *h_mem_p = (CrcMember *)((char *)staging.p + stage_mem);
    size_t h_mem_n = 0;
    u32 crc_chunks = 0, crc_groups = 0;
    for (u32 f = 0; f < n_files; ++f)
        for (u32 k = 0; k < files[f].n_members; ++k) {
            const FragGzMember &gm = files[f].members[k];
            if (gm.off + gm.len > files[f].n) return fail(GTARS_ERR_INTERNAL, "fragment wave: a gzip member lies outside its file's text");
            CrcMember m{file_off[f] + (u32)gm.off, (u32)gm.len, gm.crc, f, crc_chunks, crc_groups};
            const u32 n_ch = ((u32)gm.len + CRC_CHUNK - 1) / CRC_CHUNK;
            crc_chunks += n_ch;
            crc_groups += (n_ch + CRC_GROUP - 1) / CRC_GROUP;
            h_mem_p[h_mem_n++] = m;
        }
    if (h_mem_n) {
        Workspace &ws_crc = tls_workspace(8, st);
        if ((s = ws_crc.reserve(pad(h_mem_n * sizeof(CrcMember)) + pad((size_t)crc_chunks * 4) + pad((size_t)crc_groups * 4) + 1024))) return s;
        Carve cc{(char *)ws_crc.ptr, 0};
        CrcMember *d_mem = cc.take<CrcMember>(h_mem_n);
        u32 *d_part = cc.take<u32>(crc_chunks), *d_gpart = cc.take<u32>(crc_groups);
        GT_HIP(hipMemcpyAsync(d_mem, h_mem_p, h_mem_n * sizeof(CrcMember), hipMemcpyHostToDevice, st));
        const CrcTables *d_tb = chroms->crc.as<CrcTables>();
        const u32 nm = (u32)h_mem_n;
        if (crc_chunks)
            hipLaunchKernelGGL(k_crc_chunks, dim3((crc_chunks + 255) / 256), dim3(256), 0, st, d_text.as<unsigned char>(), (const CrcMember *)d_mem, nm,
                               crc_chunks, d_tb, d_part);
        if (crc_groups)
            hipLaunchKernelGGL(k_crc_groups, dim3((crc_groups + 63) / 64), dim3(64), 0, st, (const CrcMember *)d_mem, nm, crc_groups, d_tb,
                               (const u32 *)d_part, d_gpart);
        hipLaunchKernelGGL(k_crc_members, dim3((nm + 63) / 64), dim3(64), 0, st, (const CrcMember *)d_mem, nm, d_tb, (const u32 *)d_gpart, d_err);
        GT_HIP(hipGetLastError());
    }
    GT_WAIT(st);
    const double t1 = now_s();
    out.t_h2d = t1 - t0;
    // ---- line ends ----
    u32 *d_cnt = d_chunks.as<u32>(), *d_base = d_cnt + n_chunks + 1;
    u32 n_lines = 0;
    if (n_chunks) {
        hipLaunchKernelGGL(k_frag_lines<false>, dim3(n_chunks), dim3(FP_TPB), 0, st, d_text.as<u32>(), n_bytes, d_cnt, (const u32 *)nullptr,
                           (u32 *)nullptr);
        hipLaunchKernelGGL(k_frag_scan_chunks, dim3(1), dim3(1024), 0, st, (const u32 *)d_cnt, n_chunks, d_base);
        GT_HIP(hipMemcpyAsync((void *)&mb_words[0], d_base + n_chunks, 4, hipMemcpyDeviceToHost, st));
        GT_HIP(hipMemcpyAsync((void *)&mb_words[1], d_err, 4, hipMemcpyDeviceToHost, st));
        GT_WAIT(st);
        n_lines = mb_words[0];
    } else {
        GT_HIP(hipMemcpyAsync((void *)&mb_words[1], d_err, 4, hipMemcpyDeviceToHost, st));
        GT_WAIT(st);
    }
    if (!n_lines) {  // (no text at all: a member's CRC may still be wrong)
        const u32 h_err0 = mb_words[1];
        if (h_err0 != 0xFFFFFFFFu) out.first_error_file = h_err0;
    }
    if (n_lines) {
        const size_t sort_ws = radix_sort_ws_bytes(n_lines);
        Workspace &ws_ln = tls_workspace(6, st);
        if ((s = ws_ln.reserve(pad((size_t)n_lines * 4) + pad((size_t)n_lines * 4 * 5) + pad((size_t)n_lines * 4 * 3 + sort_ws + 64) +
                               pad(((size_t)n_lines / FP_TPB + 2) * 4) + 1024)))
            return s;
        Carve cln{(char *)ws_ln.ptr, 0};
        View d_lines{cln.take<u32>(n_lines)}, d_cols{cln.take<u32>((size_t)n_lines * 5)}, d_sort{cln.take<char>((size_t)n_lines * 4 * 3 + sort_ws + 64)};
        u32 *d_wg_written = cln.take<u32>((size_t)n_lines / FP_TPB + 2);
        u32 *d_line_end = d_lines.as<u32>();
        hipLaunchKernelGGL(k_frag_lines<true>, dim3(n_chunks), dim3(FP_TPB), 0, st, d_text.as<u32>(), n_bytes, (u32 *)nullptr, (const u32 *)d_base,
                           d_line_end);
        hipLaunchKernelGGL(k_frag_file_lines, dim3((n_files + 1 + 63) / 64), dim3(64), 0, st, (const u32 *)d_line_end, n_lines,
                           (const u32 *)d_file_off, n_files, d_file_line);
        // ---- parse: per-line columns key | chrom | start | end ----
        u32 *d_key = d_cols.as<u32>(), *d_qc = d_key + n_lines, *d_qs = d_qc + n_lines, *d_qe = d_qs + n_lines;
        FragTables tb{d_slots.as<FragSlot>(), d_keys.as<unsigned char>(), d_slot_off, d_key_off, chroms->slots.as<FragSlot>(),
                      chroms->keys.as<unsigned char>(), chroms->n_slots};
        const u32 no_key = (u32)total_slots;  // (< 2^32: the tables of a wave's files are 16 bytes a slot, of < 4 GiB of text)
        const u32 n_parse_wg = (n_lines + FP_TPB - 1) / FP_TPB;
        hipLaunchKernelGGL(k_frag_parse, dim3(n_parse_wg), dim3(FP_TPB), 0, st, d_text.as<unsigned char>(), (const u32 *)d_line_end, n_lines,
                           (const u32 *)d_file_line, n_files, tb, no_key, d_key, d_qc, d_qs, d_qe, d_written, d_wg_written, d_err);
        hipLaunchKernelGGL(k_frag_sum_written, dim3(n_files), dim3(256), 0, st, (const u32 *)d_wg_written, (const u32 *)d_file_line, n_files, d_written);
        GT_HIP(hipGetLastError());
        // ---- the lines by (file, barcode) (stable: line order inside a barcode), the tokenized ones in front ----
        u32 *d_v0 = d_sort.as<u32>(), *d_k1 = d_v0 + n_lines, *d_v1 = d_k1 + n_lines;
        void *ws = (void *)(((uintptr_t)(d_v1 + n_lines) + 63) & ~(uintptr_t)63);
        hipLaunchKernelGGL(k_frag_iota, dim3((n_lines + 255) / 256), dim3(256), 0, st, d_v0, n_lines);
        int res = 0, key_bits = 1;
        while (key_bits < 32 && (no_key >> key_bits)) ++key_bits;
        if ((s = radix_sort_pairs(d_key, d_v0, d_k1, d_v1, n_lines, 0, key_bits, ws, sort_ws, &res, st))) return s;
        const u32 *sk = res ? d_k1 : d_key, *sp = res ? d_v1 : d_v0;
        hipLaunchKernelGGL(k_frag_lower_bound, dim3(1), dim3(64), 0, st, sk, n_lines, no_key, d_coff);
        const u32 *h_written = (const u32 *)((char *)mailbox.p + mb_written), *h_file_line = (const u32 *)((char *)mailbox.p + mb_fline);
        GT_HIP(hipMemcpyAsync((void *)&mb_words[3], d_coff, 4, hipMemcpyDeviceToHost, st));
        GT_HIP(hipMemcpyAsync((void *)h_written, d_written, (size_t)n_files * 4, hipMemcpyDeviceToHost, st));
        GT_HIP(hipMemcpyAsync((void *)h_file_line, d_file_line, m1 * 4, hipMemcpyDeviceToHost, st));
        GT_HIP(hipMemcpyAsync((void *)&mb_words[1], d_err, 4, hipMemcpyDeviceToHost, st));
        GT_WAIT(st);
        const u32 h_err = mb_words[1];
        const double t2 = now_s();
        out.t_parse = t2 - t1;
        for (u32 f = 0; f < n_files; ++f) {
            out.n_reads[f] = h_file_line[f + 1] - h_file_line[f];
            out.n_written[f] = h_written[f];
        }
        if (h_err != 0xFFFFFFFFu) {
            out.first_error_file = h_err;
            return GTARS_OK;  // (the caller reports it)
        }
        const u32 n = mb_words[3];
        out.n = n;
        if (n) {
            // columns c | s | e (u32 each) in (file, barcode) order, the token CSR, the emit pass's tables
            u64 cap = (u64)n * 2 + 1024, h = 0;  // ids: a guessed capacity, the fill pass when it was short
            const u32 n_em = (n + EM_TPB - 1) / EM_TPB;
            const size_t cols_bytes = (size_t)n * 4 * 3 + 64 + ((size_t)n + 1) * 8 + ((size_t)n_em + 1) * 2 * 4 + ((size_t)no_key + 1) * 2 * 4 + 64;
            Workspace &ws_out = tls_workspace(7, st);
            if ((s = ws_out.reserve(pad(cols_bytes) + pad(cap * 4) + 1024))) return s;
            Carve cout_{(char *)ws_out.ptr, 0};
            View d_outcols{cout_.take<char>(cols_bytes)};
            u32 *d_ids_ws = cout_.take<u32>(cap);
            u32 *oc = d_outcols.as<u32>(), *os = oc + n, *oe = os + n;
            u64 *d_off = reinterpret_cast<u64 *>(((uintptr_t)(oe + n) + 7) & ~(uintptr_t)7);
            u32 *d_em_tot = reinterpret_cast<u32 *>(d_off + n + 1), *d_em_base = d_em_tot + n_em + 1;
            u32 *d_run_start = d_em_base + n_em + 1, *d_run_line = d_run_start + no_key + 1;
            hipLaunchKernelGGL(k_frag_gather, dim3((n + 255) / 256), dim3(256), 0, st, sp, n, (const u32 *)d_qc, (const u32 *)d_qs,
                               (const u32 *)d_qe, oc, os, oe);
            GT_HIP(hipGetLastError());
            GT_WAIT(st);
            const double t3 = now_s();
            out.t_group = t3 - t2;
            // ---- tokenize where the columns lie: one fused pass with a guessed capacity, the fill pass when it was short ----
            DevMem bigger;  // (only when the guess was short: a hit-heavy universe)
            u32 *d_ids_p = d_ids_ws;
            s = gtars_tokenize_device(ix, oc, os, oe, n, (uint64_t *)d_off, d_ids_p, cap, &h, st);
            if (s == GTARS_ERR_CAPACITY) {
                if ((s = bigger.alloc(h * 4))) return s;
                d_ids_p = bigger.as<u32>();
                if ((s = gtars_fill_device_n(ix, oc, os, oe, n, (const uint64_t *)d_off, d_ids_p, h, st))) return s;
                GT_WAIT(st);
            } else if (s) {
                return s;
            }
            const double t4 = now_s();
            out.t_tok = t4 - t3;
            // ---- the ids regrouped by (file, barcode), the unk id where a fragment has none (k_frag_emit) ----
            if (h + n >= 0xFFFFFFF0ull) return fail(GTARS_ERR_CAPACITY, "fragment wave: more than 4e9 token ids");  // (the caller: host parser)
            Workspace &ws_ids = tls_workspace(9, st);
            if ((s = ws_ids.reserve(pad((h + n) * 4) + 256))) return s;
            u32 *d_ids2 = (u32 *)ws_ids.ptr;
            hipLaunchKernelGGL(k_frag_emit_counts, dim3(n_em), dim3(EM_TPB), 0, st, (const u64 *)d_off, n, d_em_tot);
            hipLaunchKernelGGL(k_frag_scan_chunks, dim3(1), dim3(1024), 0, st, (const u32 *)d_em_tot, n_em, d_em_base);
            GT_HIP(hipMemsetAsync(d_run_start, 0xFF, ((size_t)no_key + 1) * 2 * 4, st));  // (run_start | run_line)
            hipLaunchKernelGGL(k_frag_emit, dim3(n_em), dim3(EM_TPB), 0, st, (const u64 *)d_off, (const u32 *)d_ids_p, n, (const u32 *)d_em_base, sk, sp,
                               unk_id, d_ids2, d_run_start, d_run_line);
            GT_HIP(hipGetLastError());
            GT_HIP(hipMemcpyAsync((void *)&mb_words[2], d_em_base + n_em, 4, hipMemcpyDeviceToHost, st));
            if (!out.run_start.alloc((size_t)no_key + 1) || !out.run_line.alloc((size_t)no_key + 1)) return fail(GTARS_ERR_INTERNAL, "out of host memory");
            GT_HIP(hipMemcpyAsync(out.run_start.get(), d_run_start, ((size_t)no_key + 1) * 4, hipMemcpyDeviceToHost, st));
            GT_HIP(hipMemcpyAsync(out.run_line.get(), d_run_line, ((size_t)no_key + 1) * 4, hipMemcpyDeviceToHost, st));
            GT_WAIT(st);
            const u32 n_ids = mb_words[2];
            out.n_ids = n_ids;
            // (the results land in pinned blocks: one DMA each, no staging by the runtime)
            if (!out.ids.alloc(std::max<u32>(n_ids, 1))) return fail(GTARS_ERR_INTERNAL, "out of host memory");
            GT_HIP(hipMemcpyAsync(out.ids.get(), d_ids2, (size_t)n_ids * 4, hipMemcpyDeviceToHost, st));
            GT_WAIT(st);
            out.t_d2h = now_s() - t4;
        }
    }
    return GTARS_OK;
}

}  // namespace gtars
