// inflate_dev.hip -- raw DEFLATE (RFC 1951) on the GPU: ONE WAVE PER STREAM, for the fused fragment pipeline's gzip'ed input
// (gtars-fragsplit/src/split.rs:84-131 reads every fragment file through flate2's MultiGzDecoder, gtars-core/src/utils.rs:115-126).
//
// Why: with the parse on the GPU the pipeline's floor is the host's inflate AND the copy of the inflated text over PCIe -- config 5
// at 1000 files: 24 ms of inflate on 16 host threads, 26 ms of host-to-device copies for 400 MB of text, 37 ms per call.  The
// compressed files are a third of the text, and a folder of fragment files is hundreds to thousands of independent streams.
//
// How a stream is decoded by a wave: DEFLATE is serial inside a stream -- the next symbol starts where the previous one ended -- so
// the wave runs the decoder's control flow UNIFORMLY (every lane computes the same bit-buffer state; values read from LDS are pinned
// to scalar registers with readfirstlane) and uses its 64 lanes where the format is data-parallel: staging the input (16 bytes per
// lane and KiB), filling the decode tables of a block, copying a match (one byte per lane), flushing the output (16 bytes per lane).
// Everything the serial chain touches lives in LDS, whose round trip is ~100 cycles against ~1 us for the L2:
//   window   32 KiB ring of the stream's last output bytes (a match reads it, literals and matches write it); flushed to the
//            result in 4-KiB pieces, 16 bytes per lane and store
//   input    2 KiB ring (+ 8 mirrored bytes), refilled a KiB at a time, the next KiB already in registers
//   tables   literal/length: 512 entries (9 bits), distance: 256 entries (8 bits), u32 each; a longer code -- rare -- is decoded
//            bit by bit from the canonical code's per-length counts and its symbols in code order (what zlib's puff.c does)
// = 39.5 KiB per wave: four streams per CU, 1024 at once on the chip.
//
// The decoder REFUSES rather than diagnoses, like the host's inflate_fast.h: status != 0 (invalid code, distance in front of the
// stream, input or output exhausted) sends the file back to the host path, whose checks and messages are the ones a user sees.
// The caller verifies length and CRC-32 of what comes out (the pipeline computes the CRC on the device anyway).
#include <hip/hip_runtime.h>

#include "common.h"

namespace gtars {

namespace {

constexpr u32 INF_WIN = 32768, INF_WIN_MASK = INF_WIN - 1u;
constexpr u32 INF_IN = 2048, INF_CHUNK = 1024;
constexpr u32 INF_LL_BITS = 9, INF_D_BITS = 8;
constexpr u32 INF_FLUSH = 4096;

// table entry: bits 0-3 code length (0: no short code -- the slow path), 4-7 number of extra bits, bit 8 literal, bit 9 end of
// block, bit 10 invalid symbol, 16-31 literal / base value
constexpr u32 IE_LIT = 1u << 8, IE_EOB = 1u << 9, IE_BAD = 1u << 10;

__constant__ unsigned char c_clen_order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

__device__ __forceinline__ u32 uni(u32 x) { return (u32)__builtin_amdgcn_readfirstlane((int)x); }

// base values and extra-bit counts of the length and distance symbols (RFC 1951 3.2.5), computed: a table in constant memory
// costs a round trip to the L2 wherever it is indexed
__device__ __forceinline__ u32 ll_entry(u32 sym) {
    if (sym < 256u) return IE_LIT | (sym << 16);
    if (sym == 256u) return IE_EOB;
    if (sym < 265u) return (sym - 254u) << 16;  // lengths 3..10, no extra bits
    if (sym < 285u) {
        const u32 x = sym - 261u, xb = x >> 2;  // 1..5 extra bits
        return ((3u + ((4u + (x & 3u)) << xb)) << 16) | (xb << 4);
    }
    if (sym == 285u) return 258u << 16;
    return IE_BAD;
}
__device__ __forceinline__ u32 d_entry(u32 sym) {
    if (sym < 4u) return (sym + 1u) << 16;
    if (sym < 30u) {
        const u32 xb = (sym >> 1) - 1u;  // 1..13 extra bits
        return ((1u + ((2u + (sym & 1u)) << xb)) << 16) | (xb << 4);
    }
    return IE_BAD;
}
__device__ __forceinline__ u32 bitrev(u32 code, u32 len) { return __brev(code) >> (32u - len); }

// per-wave LDS
struct InfLds {
    unsigned char win[INF_WIN];
    unsigned char in[INF_IN + 16];
    u32 ll[1u << INF_LL_BITS];
    u32 dd[1u << INF_D_BITS];
    unsigned short ll_sym[288], d_sym[32];  // symbols in code order (canonical decoding of long codes)
    unsigned short ll_cnt[16], d_cnt[16];    // codes per length
    unsigned char lens[320];                 // code lengths of the block being set up
    u32 cl[128];                             // code-length code: 7-bit table (entry: length | symbol << 8; 0: invalid)
};

struct BitReader {
    const unsigned char *src;  // the stream (global)
    u32 n;                     // its length
    u64 bb;                    // bit buffer
    u32 bn;                    // valid bits
    u32 ip;                    // bytes of the stream consumed into the buffer
    u32 staged;                // stream bytes [0, staged) have been written to the input ring (a multiple of INF_CHUNK)
    uint4 pre;                 // the chunk [staged, staged + INF_CHUNK): this lane's 16 bytes, in flight or landed
    u64 w0, w1;                // the two aligned 8-byte words of the input ring that hold the bytes at ip, requested by the previous
                               // refill (an LDS round trip ahead of their use: the refill is off the symbol-to-symbol chain)
};

// this lane's 16 bytes of the chunk at `at` (zeros beyond the stream: the decoder notices exhaustion by ip > n)
__device__ __forceinline__ uint4 inf_fetch(const BitReader &r, u32 at, int lane) {
    const u32 o = at + (u32)lane * 16u;
    // (the stream's buffer is padded to a multiple of 16 bytes + 16 by the caller, and its start is 16-byte aligned)
    if (o < r.n + 16u) return *reinterpret_cast<const uint4 *>(r.src + o);
    return make_uint4(0, 0, 0, 0);
}
__device__ __forceinline__ void inf_stage(InfLds &L, BitReader &r, int lane) {
    // write the pre-fetched chunk into its half of the ring, request the next one
    const u32 slot = r.staged & (INF_IN - 1u);
    *reinterpret_cast<uint4 *>(L.in + slot + (u32)lane * 16u) = r.pre;
    if (slot == 0 && lane == 0) *reinterpret_cast<uint4 *>(L.in + INF_IN) = r.pre;  // the mirror of the ring's first 16 bytes
    r.staged += INF_CHUNK;
    r.pre = inf_fetch(r, r.staged, lane);
}
// request the two aligned words around ip (all lanes the same address: a broadcast).  Aligned reads: an 8-byte LDS read at an
// arbitrary byte offset is no fast path.
__device__ __forceinline__ void inf_request(InfLds &L, BitReader &r, int lane) {
    while (r.ip + 16u > r.staged) inf_stage(L, r, lane);  // (the bytes [ip & ~7, + 16) must be staged)
    const u32 at = r.ip & (INF_IN - 1u) & ~7u;
    r.w0 = *reinterpret_cast<const u64 *>(L.in + at);
    r.w1 = *reinterpret_cast<const u64 *>(L.in + at + 8u);
}
__device__ __forceinline__ void inf_refill(InfLds &L, BitReader &r, int lane) {
    // the eight bytes at ip, out of the two words requested by the previous refill
    const u32 sh = (r.ip & 7u) * 8u;
    const u64 a = ((u64)uni((u32)(r.w0 >> 32)) << 32) | uni((u32)r.w0), b = ((u64)uni((u32)(r.w1 >> 32)) << 32) | uni((u32)r.w1);
    const u64 w = sh ? (a >> sh) | (b << (64u - sh)) : a;
    r.bb |= w << r.bn;
    r.ip += (63u - r.bn) >> 3;
    r.bn |= 56u;
    inf_request(L, r, lane);
}
__device__ __forceinline__ u32 inf_bits(BitReader &r, u32 k) {  // k <= 32 bits, already in the buffer
    const u32 v = (u32)(r.bb & ((1ull << k) - 1ull));
    r.bb >>= k;
    r.bn -= k;
    return v;
}

// canonical decoding, bit by bit (codes longer than the table's index: rare).  The code's bits are consumed; returns the symbol's
// table entry with a length field of 0, or IE_BAD.
__device__ u32 inf_slow(BitReader &r, const unsigned short *cnt, const unsigned short *sym, int kind) {
    u32 code = 0, first = 0, index = 0;
#pragma unroll 1
    for (u32 len = 1; len <= 15; ++len) {
        code |= (u32)(r.bb & 1ull);
        r.bb >>= 1;
        r.bn -= 1;
        const u32 c = uni(cnt[len]);
        if (code < first + c) {
            const u32 s = uni(sym[index + (code - first)]);
            return kind ? d_entry(s) : ll_entry(s);
        }
        index += c;
        first += c;
        first <<= 1;
        code <<= 1;
    }
    return IE_BAD;
}

// Decode tables of a canonical Huffman code given as code lengths in L.lens[base, base + n).  Lanes work on symbols side by side.
// kind 0: literal/length, 1: distance.  -> false: over-subscribed or incomplete (a distance code of <= 1 symbol is accepted)
__device__ bool inf_build(InfLds &L, u32 base, u32 n, int kind, int lane) {
    u32 *table = kind ? L.dd : L.ll;
    unsigned short *cnt = kind ? L.d_cnt : L.ll_cnt, *syms = kind ? L.d_sym : L.ll_sym;
    const u32 tb = kind ? INF_D_BITS : INF_LL_BITS;
    // counts per length (uniform: n <= 288)
    u32 count[16];
#pragma unroll
    for (int l = 0; l < 16; ++l) count[l] = 0;
    for (u32 i0 = 0; i0 < n; i0 += 64) {
        const u32 i = i0 + (u32)lane;
        const u32 l = i < n ? L.lens[base + i] : 0u;
#pragma unroll
        for (int k = 1; k < 16; ++k) count[k] += (u32)__popcll(__ballot(l == (u32)k));
    }
    u32 used = 0;
#pragma unroll
    for (int l = 1; l < 16; ++l) used += count[l];
    int left = 1;
    bool over = false;
#pragma unroll
    for (int l = 1; l < 16; ++l) {
        left = (left << 1) - (int)count[l];
        over = over || left < 0;
    }
    if (over) return false;
    if (left > 0 && !(kind == 1 && used <= 1u)) return false;  // incomplete (zlib accepts it only for the distance code of <= 1 symbol)
    // first code and first index (in code order) of every length
    u32 next_code[16], offs[16];
    {
        u32 code = 0, o = 0;
        next_code[0] = 0, offs[0] = 0;
#pragma unroll
        for (int l = 1; l < 16; ++l) {
            code = (code + (l > 1 ? count[l - 1] : 0u)) << 1;
            next_code[l] = code;
            offs[l] = o;
            o += count[l];
        }
    }
    if (lane < 16) cnt[lane] = 0;
#pragma unroll
    for (int l = 1; l < 16; ++l)
        if (lane == l) cnt[l] = (unsigned short)count[l];
    for (u32 i = (u32)lane; i < (1u << tb); i += 64) table[i] = 0u;  // 0: no short code here (slow path, or invalid)
    // symbols in order: symbol s of length l is number (symbols of length l below s) among its length
    for (u32 i0 = 0; i0 < n; i0 += 64) {
        const u32 s = i0 + (u32)lane;
        const u32 l = s < n ? L.lens[base + s] : 0u;
        u32 rank = 0, nc = 0, of = 0;
#pragma unroll
        for (int k = 1; k < 16; ++k) {
            const u64 m = __ballot(l == (u32)k);
            if (l == (u32)k) {
                rank = (u32)__popcll(m & ((1ull << lane) - 1ull));
                nc = next_code[k];
                of = offs[k];
            }
            const u32 c = (u32)__popcll(m);
            next_code[k] += c;
            offs[k] += c;
        }
        if (l) {
            const u32 code = nc + rank;
            syms[of + rank] = (unsigned short)s;
            if (l <= tb) {
                const u32 e = (kind ? d_entry(s) : ll_entry(s)) | l;
                const u32 r = bitrev(code, l);
                for (u32 x = r; x < (1u << tb); x += 1u << l) table[x] = e;
            }
        }
    }
    return true;
}

}  // namespace

#ifndef INF_STATS
#define INF_STATS 0  // diagnostic build: symbol counts and shader-clock totals (tools/r06_inflate_bench.py prints them)
#endif
#if INF_STATS
__device__ unsigned long long g_inf_stats[8];  // literals, matches, blocks, cycles: block setup, symbol loop, total; match bytes; slow-path symbols
#define INF_COUNT(k, v) (is_acc[k] += (v))
#define INF_CLOCK() __builtin_amdgcn_s_memtime()
#else
#define INF_COUNT(k, v) \
    do {                \
    } while (0)
#define INF_CLOCK() 0ull
#endif
// status: 0 ok | 1 invalid block type / stored length | 2 invalid code lengths | 3 invalid symbol or distance | 4 input exhausted |
// 5 output capacity exceeded
__global__ void __launch_bounds__(64)
k_inflate_streams(const unsigned char *__restrict__ comp, const u64 *__restrict__ in_off, const u32 *__restrict__ in_len,
                  unsigned char *__restrict__ out, const u64 *__restrict__ out_off, const u32 *__restrict__ out_cap, u32 n_streams,
                  u32 *__restrict__ out_len, u32 *__restrict__ consumed, u32 *__restrict__ status) {
    __shared__ InfLds L;
    const int lane = threadIdx.x;
    for (u32 sid = blockIdx.x; sid < n_streams; sid += gridDim.x) {
        BitReader r;
        r.src = comp + in_off[sid];
        r.n = in_len[sid];
        r.bb = 0, r.bn = 0, r.ip = 0, r.staged = 0;
        r.pre = inf_fetch(r, 0, lane);
        inf_request(L, r, lane);
        unsigned char *dst = out + out_off[sid];
        const u32 cap = out_cap[sid];
        u32 op = 0, flushed = 0, st = 0;
#if INF_STATS
        u64 is_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const u64 is_t0 = INF_CLOCK();
#endif
        auto flush = [&](u32 upto) {  // whole 4-KiB pieces of the ring to the result
            while (flushed + INF_FLUSH <= upto) {
#pragma unroll
                for (u32 k = 0; k < INF_FLUSH / 1024u; ++k) {
                    const u32 o = flushed + k * 1024u + (u32)lane * 16u;
                    *reinterpret_cast<uint4 *>(dst + o) = *reinterpret_cast<const uint4 *>(L.win + (o & INF_WIN_MASK));
                }
                flushed += INF_FLUSH;
            }
        };
        bool last = false;
        while (!last && !st) {
            [[maybe_unused]] const u64 is_tb = INF_CLOCK();
            INF_COUNT(2, 1);
            inf_refill(L, r, lane);
            last = inf_bits(r, 1) != 0;
            const u32 type = inf_bits(r, 2);
            if (type == 0) {
                // stored: to the byte boundary, LEN, NLEN, the bytes
                inf_bits(r, r.bn & 7u);
                inf_refill(L, r, lane);
                const u32 len = inf_bits(r, 16), nlen = inf_bits(r, 16);
                if ((len ^ 0xFFFFu) != nlen) {
                    st = 1;
                    break;
                }
                if (op + len > cap) {
                    st = 5;
                    break;
                }
                for (u32 i = 0; i < len; ++i) {
                    if (r.bn < 8) inf_refill(L, r, lane);
                    const u32 b = inf_bits(r, 8);
                    if (lane == 0) L.win[(op + i) & INF_WIN_MASK] = (unsigned char)b;
                    if (((op + i + 1u) & (INF_FLUSH - 1u)) == 0) flush(op + i + 1u);
                }
                op += len;
                if (r.ip - (r.bn >> 3) > r.n) st = 4;
                continue;
            }
            if (type == 3) {
                st = 1;
                break;
            }
            if (type == 1) {
                // fixed code
                for (u32 i = (u32)lane; i < 288u; i += 64) L.lens[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8;
                if (lane < 32) L.lens[288 + lane] = 5;  // (30 and 31 never occur in valid data: IE_BAD)
                if (!inf_build(L, 0, 288, 0, lane) || !inf_build(L, 288, 32, 1, lane)) {
                    st = 2;
                    break;
                }
            } else {
                const u32 hlit = inf_bits(r, 5) + 257u, hdist = inf_bits(r, 5) + 1u, hclen = inf_bits(r, 4) + 4u;
                if (hlit > 286u || hdist > 30u) {
                    st = 2;
                    break;
                }
                // the code-length code: 19 lengths of 3 bits
                inf_refill(L, r, lane);
                u32 cl_len[19];
#pragma unroll
                for (int i = 0; i < 19; ++i) cl_len[i] = 0;
                for (u32 i = 0; i < hclen; ++i) {
                    if (r.bn < 3) inf_refill(L, r, lane);
                    const u32 v = inf_bits(r, 3);
                    const u32 pos = c_clen_order[i];
#pragma unroll
                    for (int k = 0; k < 19; ++k) cl_len[k] = pos == (u32)k ? v : cl_len[k];
                }
                {
                    // its 7-bit table (uniform, 19 symbols)
                    u32 count[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                    for (int k = 0; k < 19; ++k)
#pragma unroll
                        for (int l = 1; l < 8; ++l) count[l] += cl_len[k] == (u32)l ? 1u : 0u;
                    int left = 1;
                    bool over = false;
#pragma unroll
                    for (int l = 1; l < 8; ++l) {
                        left = (left << 1) - (int)count[l];
                        over = over || left < 0;
                    }
                    if (over || left > 0) {
                        st = 2;
                        break;
                    }
                    u32 next_code[8];
                    u32 code = 0;
                    next_code[0] = 0;
#pragma unroll
                    for (int l = 1; l < 8; ++l) {
                        code = (code + (l > 1 ? count[l - 1] : 0u)) << 1;
                        next_code[l] = code;
                    }
                    for (u32 i = (u32)lane; i < 128u; i += 64) L.cl[i] = 0u;
#pragma unroll
                    for (int k = 0; k < 19; ++k) {
                        const u32 l = cl_len[k];
                        if (l) {
                            u32 c = 0;
#pragma unroll
                            for (int q = 1; q < 8; ++q) c = l == (u32)q ? next_code[q] : c;
#pragma unroll
                            for (int q = 1; q < 8; ++q) next_code[q] += l == (u32)q ? 1u : 0u;
                            const u32 rv = bitrev(c, l);
                            for (u32 x = rv + ((u32)lane << l); x < 128u; x += 64u << l) L.cl[x] = l | ((u32)k << 8);
                        }
                    }
                }
                // the literal/length and distance code lengths
                const u32 total = hlit + hdist;
                u32 i = 0, prev = 0;
                while (i < total && !st) {
                    if (r.bn < 16) inf_refill(L, r, lane);
                    const u32 e = uni(L.cl[(u32)r.bb & 127u]);
                    if (!e) {
                        st = 2;
                        break;
                    }
                    inf_bits(r, e & 15u);
                    const u32 sym = e >> 8;
                    if (sym < 16u) {
                        if (lane == 0) L.lens[i] = (unsigned char)sym;
                        prev = sym;
                        ++i;
                    } else {
                        u32 rep, val = 0;
                        if (sym == 16u) {
                            if (i == 0) {
                                st = 2;
                                break;
                            }
                            rep = 3u + inf_bits(r, 2);
                            val = prev;
                        } else if (sym == 17u) {
                            rep = 3u + inf_bits(r, 3);
                        } else {
                            rep = 11u + inf_bits(r, 7);
                        }
                        if (i + rep > total) {
                            st = 2;
                            break;
                        }
                        for (u32 k = (u32)lane; k < rep; k += 64) L.lens[i + k] = (unsigned char)val;
                        i += rep;
                        prev = val;
                    }
                }
                if (st) break;
                if (uni(L.lens[256]) == 0u) {  // no end-of-block code
                    st = 2;
                    break;
                }
                // (the distance lengths follow the literal/length ones in L.lens: build from [0, hlit) and [hlit, hlit + hdist))
                if (!inf_build(L, 0, hlit, 0, lane) || !inf_build(L, hlit, hdist, 1, lane)) {
                    st = 2;
                    break;
                }
            }
            // ---- the block's symbols
            [[maybe_unused]] const u64 is_ts = INF_CLOCK();
            INF_COUNT(3, is_ts - is_tb);
            // (a single wave issues one instruction every four to five cycles: what this loop costs is its INSTRUCTION COUNT, not the
            // LDS round trips -- the rare cases sit behind __builtin_expect, the capacity is checked where the output is flushed, a
            // literal is written by every lane, and 48 bits at the top cover a length code, its extra bits, a distance code and its
            // extra bits: 15 + 5 + 15 + 13)
            while (true) {
                if (r.bn < 48) inf_refill(L, r, lane);
                u32 e = uni(L.ll[(u32)r.bb & ((1u << INF_LL_BITS) - 1u)]);
                const u32 l = e & 15u;
                if (__builtin_expect(l == 0u, 0)) {
                    e = inf_slow(r, L.ll_cnt, L.ll_sym, 0);
                    INF_COUNT(7, 1);
                }
                r.bb >>= l;
                r.bn -= l;
                if (e & IE_LIT) {
                    L.win[op & INF_WIN_MASK] = (unsigned char)(e >> 16);  // (every lane: the same byte to the same address)
                    INF_COUNT(0, 1);
                    ++op;
                    if (__builtin_expect((op & (INF_FLUSH - 1u)) == 0u, 0)) {
                        if (op > cap) {
                            st = 5;
                            break;
                        }
                        flush(op);
                    }
                    continue;
                }
                if (__builtin_expect((e & (IE_EOB | IE_BAD)) != 0u, 0)) {
                    if (e & IE_BAD) st = 3;
                    break;
                }
                const u32 xb = (e >> 4) & 15u;
                const u32 len = (e >> 16) + ((u32)r.bb & ((1u << xb) - 1u));
                r.bb >>= xb;
                r.bn -= xb;
                u32 d = uni(L.dd[(u32)r.bb & ((1u << INF_D_BITS) - 1u)]);
                const u32 dl = d & 15u;
                if (__builtin_expect(dl == 0u, 0)) d = inf_slow(r, L.d_cnt, L.d_sym, 1);
                r.bb >>= dl;
                r.bn -= dl;
                const u32 dxb = (d >> 4) & 15u;
                const u32 dist = (d >> 16) + ((u32)r.bb & ((1u << dxb) - 1u));
                r.bb >>= dxb;
                r.bn -= dxb;
                if (__builtin_expect((d & IE_BAD) != 0u || dist > op, 0)) {
                    st = 3;
                    break;
                }
                // the copy: every byte comes from the `dist` bytes in front of the match (periodically when dist < len), so all
                // reads precede all writes and the lanes need not wait for each other
                const u32 from = op - dist;
                INF_COUNT(1, 1);
                INF_COUNT(6, len);
                if (__builtin_expect(dist >= len, 1)) {
                    for (u32 i0 = 0; i0 < len; i0 += 64) {
                        const u32 i = i0 + (u32)lane;
                        if (i < len) L.win[(op + i) & INF_WIN_MASK] = L.win[(from + i) & INF_WIN_MASK];
                    }
                } else {
                    unsigned char b[5];  // (len <= 258: at most five bytes per lane)
#pragma unroll
                    for (u32 k = 0; k < 5; ++k) {
                        const u32 i = (u32)lane + 64u * k;
                        b[k] = i < len ? L.win[(from + i % dist) & INF_WIN_MASK] : (unsigned char)0;
                    }
#pragma unroll
                    for (u32 k = 0; k < 5; ++k) {
                        const u32 i = (u32)lane + 64u * k;
                        if (i < len) L.win[(op + i) & INF_WIN_MASK] = b[k];
                    }
                }
                const u32 np = op + len;
                const bool crossed = ((np ^ op) >> 12) != 0u;  // INF_FLUSH == 4096
                op = np;
                if (__builtin_expect(crossed, 0)) {
                    if (op > cap) {
                        st = 5;
                        break;
                    }
                    flush(op);
                }
            }
            INF_COUNT(4, INF_CLOCK() - is_ts);
            if (r.ip - (r.bn >> 3) > r.n) st = st ? st : 4;
        }
        // the rest of the output, byte by byte
        if (!st && op > cap) st = 5;
        if (!st)
            for (u32 o = flushed + (u32)lane; o < op; o += 64) dst[o] = L.win[o & INF_WIN_MASK];
#if INF_STATS
        is_acc[5] = INF_CLOCK() - is_t0;
        if (lane == 0)
            for (int k = 0; k < 8; ++k) atomicAdd(&g_inf_stats[k], is_acc[k]);
#endif
        if (lane == 0) {
            out_len[sid] = op;
            consumed[sid] = r.ip - (r.bn >> 3);
            status[sid] = st;
        }
    }
}

gtars_status launch_inflate_streams(const unsigned char *comp, const u64 *in_off, const u32 *in_len, unsigned char *out, const u64 *out_off,
                                    const u32 *out_cap, u32 n_streams, u32 *out_len, u32 *consumed, u32 *status, hipStream_t st) {
    if (!n_streams) return GTARS_OK;
    int dev = 0, cus = 256;
    GT_HIP(hipGetDevice(&dev));
    GT_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const u32 grid = std::min<u32>(n_streams, (u32)cus * 4u);
    hipLaunchKernelGGL(k_inflate_streams, dim3(grid), dim3(64), 0, st, comp, in_off, in_len, out, out_off, out_cap, n_streams, out_len, consumed, status);
    GT_HIP(hipGetLastError());
    return GTARS_OK;
}

}  // namespace gtars

#if INF_STATS
extern "C" int gtars_debug_inflate_stats(unsigned long long *out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(gtars::g_inf_stats), 64) != hipSuccess) return 1;
    if (reset) {
        unsigned long long z[8] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(gtars::g_inf_stats), z, 64) != hipSuccess) return 1;
    }
    return 0;
}
#endif
// test / measurement entry (include/gtars_amd_debug.h): all pointers are device memory; see k_inflate_streams
extern "C" int gtars_debug_inflate_streams(const void *comp, const uint64_t *in_off, const uint32_t *in_len, void *out, const uint64_t *out_off,
                                           const uint32_t *out_cap, uint32_t n_streams, uint32_t *out_len, uint32_t *consumed, uint32_t *status,
                                           void *stream) {
    return (int)gtars::launch_inflate_streams((const unsigned char *)comp, (const gtars::u64 *)in_off, in_len, (unsigned char *)out,
                                              (const gtars::u64 *)out_off, out_cap, n_streams, out_len, consumed, status, (hipStream_t)stream);
}
