// inflate_fast.h -- a raw DEFLATE (RFC 1951) decoder for whole-buffer inputs: the host half of the fused fragment pipeline
// (gtars-fragsplit/src/split.rs:84-131 reads every fragment file through flate2's MultiGzDecoder, gtars-core/src/utils.rs:115-126).
//
// Why not zlib's inflate(): with the parse on the GPU the pipeline's floor IS the inflate (21 of 28 ms for config 5's 48 files on
// 16 host threads), and zlib 1.2.11's decoder is built for streaming -- a 32-bit bit reservoir refilled byte by byte, a sliding
// window, one symbol per loop turn.  Here the whole compressed file and the whole result are in memory, so
//   * the bit reservoir is 64 bits wide and refilled with ONE unaligned 8-byte load, branch-free (the input buffer is padded);
//   * the result buffer is the window: a match is copied from the bytes already written, eight at a time;
//   * the literal / length table is indexed with 11 bits, so that almost every symbol resolves in one lookup, and up to three
//     literals are emitted per refill;
//   * there is no CRC pass (the caller checks CRC-32 on the GPU, or with zlib's crc32 when it needs the host check).
// The decoder REFUSES rather than diagnoses: anything it does not like (an invalid or incomplete code, a distance in front of the
// member, input that ends early) returns false, and the caller reads the file again through zlib, whose checks and messages are
// the ones a user sees.  Every member's CRC-32 and length are verified by the caller, so a wrong decode cannot pass silently.
//
// Written from RFC 1951; table layout and refill scheme are the widely used ones (a packed 32-bit entry per table slot, "variant
// 4" of the branch-free refill).  Host code, no HIP.
#pragma once

#include <cstdint>
#include <cstring>
#include <string>

namespace gtars {
namespace fastinf {

constexpr unsigned LL_BITS = 11, OFF_BITS = 8, PRE_BITS = 7;
constexpr unsigned LL_SIZE = (1u << LL_BITS) + 288 * 16, OFF_SIZE = (1u << OFF_BITS) + 32 * 128, PRE_SIZE = 1u << PRE_BITS;
// a table entry: bits 0-5 bits to consume (a length's / distance's code AND its extra bits: their value is picked out of the bits
// as they were before, off the chain bit buffer -> lookup -> bit buffer) | 8-11 extra bits (or a sub-table's index width) | 12 end of block | 13 sub-table
// pointer | 14 exceptional (12, 13 or an unused code) | 15 literal | 16-31 base value / literal / sub-table start
constexpr uint32_t E_EOB = 1u << 12, E_SUB = 1u << 13, E_EXC = 1u << 14, E_LIT = 1u << 15;

inline uint64_t load64(const unsigned char *p) {
    uint64_t v;
    memcpy(&v, p, 8);
    return v;  // (little-endian hosts: x86-64, what the GPU boxes are)
}
inline void store64(unsigned char *p, uint64_t v) { memcpy(p, &v, 8); }

// kind 0: literal / length alphabet, 1: distance alphabet, 2: code-length alphabet
inline uint32_t symbol_entry(int kind, unsigned sym) {
    static const unsigned short len_base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const unsigned char len_extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const unsigned short off_base[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const unsigned char off_extra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    if (kind == 2) return (uint32_t)sym << 16;
    if (kind == 1) return sym < 30 ? ((uint32_t)off_base[sym] << 16) | ((uint32_t)off_extra[sym] << 8) : E_EXC;
    if (sym < 256) return E_LIT | ((uint32_t)sym << 16);
    if (sym == 256) return E_EXC | E_EOB;
    return sym < 286 ? ((uint32_t)len_base[sym - 257] << 16) | ((uint32_t)len_extra[sym - 257] << 8) : E_EXC;
}

// Decode table of a canonical Huffman code (RFC 1951 3.2.2) given as code lengths: `tb` index bits, longer codes through
// sub-tables behind the first 2^tb entries.  -> false: over-subscribed, or incomplete in a way zlib does not accept either (a
// distance code of no or one symbol is the exception: blocks of literals only).
inline bool build_table(const unsigned char *lens, unsigned n, int kind, unsigned tb, uint32_t *table, unsigned table_size) {
    unsigned count[16] = {0};
    for (unsigned i = 0; i < n; ++i) ++count[lens[i]];
    const unsigned used = n - count[0];
    if (used == 0 || (used == 1 && count[1] == 1)) {
        if (kind != 1) return false;
        for (unsigned i = 0; i < (1u << tb); ++i) table[i] = E_EXC | 1u;
        if (used)
            for (unsigned s = 0; s < n; ++s)
                if (lens[s])
                    for (unsigned i = 0; i < (1u << tb); i += 2) {  // (code "0")
                        const uint32_t e = symbol_entry(kind, s);
                        table[i] = e | (1u + ((e & E_EXC) ? 0u : (e >> 8) & 15u));
                    }
        return true;
    }
    int left = 1;
    unsigned max_len = 0;
    for (unsigned l = 1; l <= 15; ++l) {
        left = (left << 1) - (int)count[l];
        if (left < 0) return false;
        if (count[l]) max_len = l;
    }
    if (left != 0) return false;
    unsigned next_code[16];
    {
        unsigned code = 0;
        for (unsigned l = 1; l <= 15; ++l) {
            code = (code + count[l - 1] * (l > 1 ? 1u : 0u)) << 1;
            next_code[l] = code;
        }
    }
    auto reversed = [](unsigned code, unsigned len) {
        unsigned r = 0;
        for (unsigned i = 0; i < len; ++i) r |= ((code >> i) & 1u) << (len - 1 - i);
        return r;
    };
    // sub-tables: the longest code behind every tb-bit prefix
    unsigned char sub_len[1u << LL_BITS];
    const bool long_codes = max_len > tb;
    if (long_codes) {
        memset(sub_len, 0, (size_t)1 << tb);
        unsigned nc[16];
        memcpy(nc, next_code, sizeof nc);
        for (unsigned s = 0; s < n; ++s) {
            const unsigned l = lens[s];
            if (!l) continue;
            const unsigned r = reversed(nc[l]++, l);
            if (l > tb && sub_len[r & ((1u << tb) - 1u)] < l) sub_len[r & ((1u << tb) - 1u)] = (unsigned char)l;
        }
        unsigned next_free = 1u << tb;
        for (unsigned p = 0; p < (1u << tb); ++p)
            if (sub_len[p]) {
                const unsigned bits = sub_len[p] - tb;
                if (next_free + (1u << bits) > table_size) return false;
                table[p] = E_EXC | E_SUB | (next_free << 16) | (bits << 8) | tb;
                next_free += 1u << bits;
            }
    }
    for (unsigned s = 0; s < n; ++s) {
        const unsigned l = lens[s];
        if (!l) continue;
        const unsigned r = reversed(next_code[l]++, l);
        if (l <= tb) {
            uint32_t e = symbol_entry(kind, s);
            e |= l + (kind != 2 && !(e & (E_LIT | E_EXC)) ? (e >> 8) & 15u : 0u);  // (bits to consume: the code and its extra bits)
            for (unsigned i = r; i < (1u << tb); i += 1u << l) table[i] = e;
        } else {
            const uint32_t ptr = table[r & ((1u << tb) - 1u)];
            const unsigned start = ptr >> 16, bits = (ptr >> 8) & 15u;
            uint32_t e = symbol_entry(kind, s);
            e |= (l - tb) + (kind != 2 && !(e & (E_LIT | E_EXC)) ? (e >> 8) & 15u : 0u);
            for (unsigned i = r >> tb; i < (1u << bits); i += 1u << (l - tb)) table[start + i] = e;
        }
    }
    return true;
}

struct FixedTables {
    uint32_t ll[LL_SIZE], off[OFF_SIZE];
    FixedTables() {
        unsigned char lens[288 + 32];
        for (unsigned i = 0; i < 144; ++i) lens[i] = 8;
        for (unsigned i = 144; i < 256; ++i) lens[i] = 9;
        for (unsigned i = 256; i < 280; ++i) lens[i] = 7;
        for (unsigned i = 280; i < 288; ++i) lens[i] = 8;
        for (unsigned i = 0; i < 32; ++i) lens[288 + i] = 5;
        build_table(lens, 288, 0, LL_BITS, ll, LL_SIZE);
        build_table(lens + 288, 32, 1, OFF_BITS, off, OFF_SIZE);
    }
};

// One raw deflate stream from in[0, in_n) -- the buffer must be readable (any content) for 16 bytes beyond in_n -- appended to
// `out` at out_done (the string is grown as needed; its size is NOT trimmed: the caller resizes to out_done at the end).
// member_start: out position of the stream's first byte (no match may reach in front of it).  *in_used: bytes of `in` the stream
// occupied (rounded up to the byte).  -> false: refused (see the header).
// Out: a byte buffer with size(), resize(n) (contents kept; new bytes undefined or zero) and operator[] -- std::string, or the
// host layer's pinned text buffer
template <class Out>
static inline __attribute__((always_inline)) bool inflate_raw_body(const FixedTables &fixed, const unsigned char *in, size_t in_n, size_t *in_used,
                                                                   Out &out, size_t &out_done) {
    const size_t member_start = out_done;
    uint32_t dyn_ll[LL_SIZE], dyn_off[OFF_SIZE], pre[PRE_SIZE];
    const unsigned char *in_next = in, *const in_guard = in + in_n + 8;  // (a refill at in_guard still reads inside the padding)
    uint64_t bitbuf = 0;
    unsigned bitcnt = 0;
    constexpr size_t MARGIN = 258 + 8 + 64;
    if (out.size() < out_done + MARGIN) out.resize(out_done + MARGIN + (in_n << 2));
    unsigned char *out_base = (unsigned char *)&out[0];
    unsigned char *out_next = out_base + out_done, *out_guard = out_base + out.size() - MARGIN;
#define GTARS_INF_REFILL()                                  \
    do {                                                    \
        if (in_next > in_guard) return false;               \
        bitbuf |= load64(in_next) << bitcnt;                \
        in_next += (63u - bitcnt) >> 3;                     \
        bitcnt |= 56u;                                      \
    } while (0)
#define GTARS_INF_TAKE(n_) (bitbuf >>= (n_), bitcnt -= (n_))
    auto grow = [&]() {
        const size_t done = (size_t)(out_next - out_base);
        out.resize(out.size() + out.size() / 2 + (1u << 16));
        out_base = (unsigned char *)&out[0];
        out_next = out_base + done;
        out_guard = out_base + out.size() - MARGIN;
    };
    for (bool last = false; !last;) {
        GTARS_INF_REFILL();
        last = bitbuf & 1u;
        const unsigned type = (unsigned)(bitbuf >> 1) & 3u;
        GTARS_INF_TAKE(3);
        const uint32_t *ll, *off;
        if (type == 0) {
            // stored: to the byte boundary, LEN, ~LEN, bytes
            GTARS_INF_TAKE(bitcnt & 7u);
            const unsigned char *p = in_next - (bitcnt >> 3);
            bitbuf = 0, bitcnt = 0;
            if ((size_t)(p - in) + 4 > in_n) return false;
            const unsigned len = p[0] | ((unsigned)p[1] << 8), nlen = p[2] | ((unsigned)p[3] << 8);
            if ((len ^ nlen) != 0xFFFFu) return false;
            p += 4;
            if ((size_t)(p - in) + len > in_n) return false;
            while ((size_t)(out_base + out.size() - out_next) < len + MARGIN) grow();
            memcpy(out_next, p, len);
            out_next += len;
            in_next = p + len;
            continue;
        } else if (type == 1) {
            ll = fixed.ll, off = fixed.off;
        } else if (type == 2) {
            const unsigned hlit = ((unsigned)bitbuf & 31u) + 257u, hdist = ((unsigned)(bitbuf >> 5) & 31u) + 1u, hclen = ((unsigned)(bitbuf >> 10) & 15u) + 4u;
            GTARS_INF_TAKE(14);
            if (hlit > 286 || hdist > 30) return false;
            static const unsigned char order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
            unsigned char plen[19] = {0};
            GTARS_INF_REFILL();  // (>= 56 bits: 19 x 3 = 57 -- the first codes now, the rest behind another refill)
            for (unsigned i = 0; i < hclen; ++i) {
                if (i == 12) GTARS_INF_REFILL();
                plen[order[i]] = (unsigned char)(bitbuf & 7u);
                GTARS_INF_TAKE(3);
            }
            if (!build_table(plen, 19, 2, PRE_BITS, pre, PRE_SIZE)) return false;
            unsigned char lens[286 + 30 + 138];
            unsigned i = 0;
            const unsigned total = hlit + hdist;
            while (i < total) {
                GTARS_INF_REFILL();
                const uint32_t e = pre[bitbuf & (PRE_SIZE - 1u)];
                if (e & E_EXC) return false;
                GTARS_INF_TAKE(e & 63u);
                const unsigned sym = e >> 16;
                if (sym < 16) {
                    lens[i++] = (unsigned char)sym;
                } else if (sym == 16) {
                    if (!i) return false;
                    const unsigned rep = 3u + ((unsigned)bitbuf & 3u);
                    GTARS_INF_TAKE(2);
                    memset(lens + i, lens[i - 1], rep);
                    i += rep;
                } else if (sym == 17) {
                    const unsigned rep = 3u + ((unsigned)bitbuf & 7u);
                    GTARS_INF_TAKE(3);
                    memset(lens + i, 0, rep);
                    i += rep;
                } else {
                    const unsigned rep = 11u + ((unsigned)bitbuf & 127u);
                    GTARS_INF_TAKE(7);
                    memset(lens + i, 0, rep);
                    i += rep;
                }
            }
            if (i != total || !lens[256]) return false;
            if (!build_table(lens, hlit, 0, LL_BITS, dyn_ll, LL_SIZE)) return false;
            if (!build_table(lens + hlit, hdist, 1, OFF_BITS, dyn_off, OFF_SIZE)) return false;
            ll = dyn_ll, off = dyn_off;
        } else {
            return false;
        }
        // the block's symbols
        // (a length code + extra bits + a distance code + extra bits take at most 15 + 5 + 15 + 13 = 48 of the >= 56 bits a refill
        // leaves: a match that is the FIRST symbol behind a refill needs no second one; behind literals it does)
#define GTARS_INF_LENGTH()                                                                           \
    if (e & E_EXC) {                                                                                 \
        if (e & E_SUB) {                                                                             \
            GTARS_INF_TAKE(LL_BITS);                                                                 \
            e = ll[(e >> 16) + ((unsigned)bitbuf & ((1u << ((e >> 8) & 15u)) - 1u))];                \
            if (e & E_LIT) {                                                                         \
                GTARS_INF_TAKE(e & 63u);                                                             \
                *out_next++ = (unsigned char)(e >> 16);                                              \
                GTARS_INF_NEXT();                                                                    \
                continue;                                                                            \
            }                                                                                        \
        }                                                                                            \
        if (e & E_EXC) {                                                                             \
            if (!(e & E_EOB)) return false;                                                          \
            GTARS_INF_TAKE(e & 63u);                                                                 \
            break;                                                                                   \
        }                                                                                            \
    }                                                                                                \
    saved = bitbuf;                                                                                  \
    GTARS_INF_TAKE(e & 63u);                                                                         \
    xb = (e >> 8) & 15u;                                                                             \
    len = (e >> 16) + ((unsigned)(saved >> ((e & 63u) - xb)) & ((1u << xb) - 1u))
        // (the NEXT symbol's table entry is looked up before a match is copied: the lookup's latency hides behind the copy)
#define GTARS_INF_NEXT()                               \
    do {                                               \
        if (out_next > out_guard) grow();              \
        GTARS_INF_REFILL();                            \
        e = ll[bitbuf & ((1u << LL_BITS) - 1u)];       \
    } while (0)
        uint32_t e;
        GTARS_INF_NEXT();
        for (;;) {
            unsigned xb, len;
            uint64_t saved;
            if (e & E_LIT) {
                GTARS_INF_TAKE(e & 63u);
                *out_next++ = (unsigned char)(e >> 16);
                e = ll[bitbuf & ((1u << LL_BITS) - 1u)];
                if (e & E_LIT) {
                    GTARS_INF_TAKE(e & 63u);
                    *out_next++ = (unsigned char)(e >> 16);
                    e = ll[bitbuf & ((1u << LL_BITS) - 1u)];
                    if (e & E_LIT) {
                        GTARS_INF_TAKE(e & 63u);
                        *out_next++ = (unsigned char)(e >> 16);
                        GTARS_INF_NEXT();
                        continue;
                    }
                }
                // (two literals took at most 22 of the 56 bits: enough left for a length code and its extra bits, 20 at most)
                GTARS_INF_LENGTH();
                GTARS_INF_REFILL();
            } else {
                GTARS_INF_LENGTH();
            }
            e = off[bitbuf & ((1u << OFF_BITS) - 1u)];
            if (e & E_EXC) {
                if (!(e & E_SUB)) return false;
                GTARS_INF_TAKE(OFF_BITS);
                e = off[(e >> 16) + ((unsigned)bitbuf & ((1u << ((e >> 8) & 15u)) - 1u))];
                if (e & E_EXC) return false;
            }
            saved = bitbuf;
            GTARS_INF_TAKE(e & 63u);
            xb = (e >> 8) & 15u;
            const size_t dist = (e >> 16) + ((unsigned)(saved >> ((e & 63u) - xb)) & ((1u << xb) - 1u));
            if (dist > (size_t)(out_next - out_base) - member_start) return false;
            const unsigned char *src = out_next - dist;
            unsigned char *dst = out_next;
            out_next += len;
            GTARS_INF_REFILL();
            e = ll[bitbuf & ((1u << LL_BITS) - 1u)];
            if (dist >= 8) {
                // (the copy may run up to 15 bytes past the match: inside the margin, overwritten by what follows)
                store64(dst, load64(src));
                store64(dst + 8, load64(src + 8));
                if (len > 16) {
                    dst += 16, src += 16;
                    do {
                        store64(dst, load64(src));
                        store64(dst + 8, load64(src + 8));
                        dst += 16, src += 16;
                    } while (dst < out_next);
                }
            } else if (dist == 1) {
                const uint64_t v = 0x0101010101010101ull * src[0];
                do {
                    store64(dst, v);
                    dst += 8;
                } while (dst < out_next);
            } else {
                do {
                    *dst++ = *src++;
                } while (dst < out_next);
            }
            if (out_next > out_guard) {  // (the buffer moves: the entry in hand stays valid, the pointers are grow()'s business)
                grow();
            }
        }
#undef GTARS_INF_LENGTH
#undef GTARS_INF_NEXT
    }
    // what the stream occupied: up to the byte that holds its last bit
    GTARS_INF_TAKE(bitcnt & 7u);
    const unsigned char *p = in_next - (bitcnt >> 3);
    if (p < in || (size_t)(p - in) > in_n) return false;
    *in_used = (size_t)(p - in);
    out_done = (size_t)(out_next - out_base);
    return true;
#undef GTARS_INF_REFILL
#undef GTARS_INF_TAKE
}

// (the same body compiled twice: BMI2's shlx / shrx / bzhi -- variable shifts without the CL register, masks in one instruction --
// are worth 8 % of the decode on the hosts measured; picked at run time)
template <class Out>
__attribute__((target("bmi2"))) inline bool inflate_raw_bmi2(const FixedTables &fixed, const unsigned char *in, size_t in_n, size_t *in_used, Out &out,
                                                             size_t &out_done) {
    return inflate_raw_body(fixed, in, in_n, in_used, out, out_done);
}
template <class Out>
inline bool inflate_raw_plain(const FixedTables &fixed, const unsigned char *in, size_t in_n, size_t *in_used, Out &out, size_t &out_done) {
    return inflate_raw_body(fixed, in, in_n, in_used, out, out_done);
}
inline const FixedTables &fixed_tables() {
    static const FixedTables fixed;
    return fixed;
}
template <class Out>
inline bool inflate_raw(const unsigned char *in, size_t in_n, size_t *in_used, Out &out, size_t &out_done) {
    const FixedTables &fixed = fixed_tables();
    static const bool bmi2 = __builtin_cpu_supports("bmi2");
    return bmi2 ? inflate_raw_bmi2(fixed, in, in_n, in_used, out, out_done) : inflate_raw_plain(fixed, in, in_n, in_used, out, out_done);
}

}  // namespace fastinf
}  // namespace gtars
