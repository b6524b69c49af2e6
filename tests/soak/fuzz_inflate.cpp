// Soak fuzzer of gtars_amd/csrc/inflate_fast.h against zlib (host only; run it under the sanitizers):
//   g++ -O1 -g -fsanitize=address,undefined -o /tmp/fuzz_inflate tests/soak/fuzz_inflate.cpp -lz && /tmp/fuzz_inflate 3000
// Every case: random data of a random kind, deflated raw by zlib with random level / strategy / window / memory level and random
// flush points (empty stored blocks, block boundaries at odd places), decoded by inflate_raw and compared byte for byte; then the
// same stream with random bytes flipped and with its tail cut off -- the decoder may refuse or decode garbage (the caller's CRC
// check catches that), but must stay inside its buffers (the exact-size heap copies below are what ASan watches).
#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../../gtars_amd/csrc/inflate_fast.h"

static unsigned long long rs = 88172645463325252ull;
static unsigned rnd() {
    rs ^= rs << 13, rs ^= rs >> 7, rs ^= rs << 17;
    return (unsigned)(rs >> 11);
}

static std::string make_data() {
    const unsigned kind = rnd() % 8, n = rnd() % 6 ? rnd() % 70000 : rnd() % 1500000;
    std::string d;
    d.reserve(n + 64);
    switch (kind) {
        case 0: for (unsigned i = 0; i < n; ++i) d.push_back((char)rnd()); break;
        case 1: for (unsigned i = 0; i < n; ++i) d.push_back((char)('a' + rnd() % 3)); break;
        case 2: d.assign(n, 'x'); break;
        case 3: { const unsigned p = 1 + rnd() % 9; for (unsigned i = 0; i < n; ++i) d.push_back((char)('0' + i % p)); } break;
        case 4: while (d.size() < n) { char b[96]; const int k = snprintf(b, sizeof b, "chr%u\t%u\t%u\tBC%05u\t%u\n", 1 + rnd() % 22, rnd() % 100000000, rnd() % 100000000, rnd() % 500, rnd() % 4); d.append(b, k); } break;
        case 5: for (unsigned i = 0; i < n; ++i) d.push_back((char)(rnd() % 16 ? 'q' : rnd())); break;
        case 6: { std::string w; for (unsigned i = 0; i < 40; ++i) w.push_back((char)rnd()); while (d.size() < n) d.append(w, 0, 1 + rnd() % 40); } break;
        default: for (unsigned i = 0; i < n; ++i) d.push_back((char)(i < 300 ? rnd() : d[i - 1 - rnd() % 300])); break;
    }
    return d;
}

static std::vector<unsigned char> deflate_raw(const std::string &d) {
    z_stream z;
    memset(&z, 0, sizeof z);
    static const int strategies[5] = {Z_DEFAULT_STRATEGY, Z_FILTERED, Z_HUFFMAN_ONLY, Z_RLE, Z_FIXED};
    deflateInit2(&z, (int)(rnd() % 10), Z_DEFLATED, -(9 + (int)(rnd() % 7)), 1 + (int)(rnd() % 9), strategies[rnd() % 5]);
    std::vector<unsigned char> out(deflateBound(&z, d.size()) + 4096 + d.size() / 8);
    z.next_out = out.data();
    z.avail_out = (uInt)out.size();
    size_t at = 0;
    while (at < d.size()) {
        const size_t piece = rnd() % 4 ? d.size() - at : 1 + rnd() % (d.size() - at);
        z.next_in = (Bytef *)d.data() + at;
        z.avail_in = (uInt)piece;
        static const int flushes[3] = {Z_SYNC_FLUSH, Z_FULL_FLUSH, Z_BLOCK};
        if (deflate(&z, at + piece == d.size() ? Z_NO_FLUSH : flushes[rnd() % 3]) != Z_OK) abort();
        at += piece;
    }
    z.avail_in = 0;
    if (deflate(&z, Z_FINISH) != Z_STREAM_END) abort();
    out.resize(z.total_out);
    deflateEnd(&z);
    return out;
}

int main(int argc, char **argv) {
    const long cases = argc > 1 ? atol(argv[1]) : 1000;
    if (argc > 2) rs ^= strtoull(argv[2], nullptr, 10) * 0x9E3779B97F4A7C15ull;
    unsigned long long bytes = 0, refused = 0, garbage = 0;
    for (long c = 0; c < cases; ++c) {
        const std::string d = make_data();
        const std::vector<unsigned char> comp = deflate_raw(d);
        auto decode = [&](const unsigned char *src, size_t n, std::string &out, size_t &done, size_t &used) {
            std::unique_ptr<unsigned char[]> in(new unsigned char[n + 16]);  // (exactly the promised padding)
            memcpy(in.get(), src, n);
            memset(in.get() + n, 0, 16);
            out.clear();
            if (rnd() & 1) out.resize(d.size() + 512);  // (sized like the caller's ISIZE guess, or grown from nothing)
            done = 0;
            return gtars::fastinf::inflate_raw(in.get(), n, &used, out, done);
        };
        std::string out;
        size_t done = 0, used = 0;
        if (!decode(comp.data(), comp.size(), out, done, used) || done != d.size() || memcmp(out.data(), d.data(), done) || used != comp.size()) {
            fprintf(stderr, "MISMATCH in case %ld (%zu bytes, %zu compressed; decoded %zu, used %zu)\n", c, d.size(), comp.size(), done, used);
            return 1;
        }
        bytes += d.size();
        for (int k = 0; k < 6 && !comp.empty(); ++k) {
            std::vector<unsigned char> bad = comp;
            if (k < 4) {
                for (unsigned f = 0, nf = 1 + rnd() % 3; f < nf; ++f) bad[rnd() % bad.size()] ^= (unsigned char)(1u << (rnd() % 8));
            } else {
                bad.resize(rnd() % bad.size());
            }
            const bool ok = decode(bad.data(), bad.size(), out, done, used);
            refused += !ok;
            garbage += ok && (done != d.size() || memcmp(out.data(), d.data(), done));
        }
    }
    printf("fuzz_inflate: %ld cases, %.1f MB decoded bit-exact; damaged streams: %llu refused, %llu decoded to something else (the caller's CRC check)\n",
           cases, bytes / 1e6, refused, garbage);
    return 0;
}
