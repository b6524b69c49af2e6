// host.cpp -- host ("string world") layer of libgtars_amd.so: BED / BED.gz
// parsing, RegionSet, Universe + Tokenizer, fragment files, .gtok, IGD
// databases from BED files.  Declared in include/gtars_amd_host.h; every
// function cites the reference code whose behaviour it reproduces.  All
// overlap / tokenization compute goes through the engine C ABI (gtars_amd.h),
// i.e. the HIP kernels -- nothing here searches intervals on the CPU.
#include <dirent.h>
#include <sys/stat.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <limits>
#include <map>
#include <memory>
#include <mutex>
#include <condition_variable>
#include <numeric>
#include <set>
#include <string>
#include <string_view>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/gtars_amd_host.h"
#include "frag_device.h"
#include "inflate_fast.h"

namespace gtars {
gtars_status fail(gtars_status st, const std::string &msg);
const char *cfg_get(const char *name);  // snapshot of the GTARS_* environment (common.h)

// runs f(), turning C++ exceptions into a status (nothing may unwind through the extern "C" boundary)
template <class F>
static gtars_status guarded(F &&f) {
    try {
        return f();
    } catch (const std::bad_alloc &) {
        return fail(GTARS_ERR_INTERNAL, "out of host memory");
    } catch (const std::exception &e) {
        return fail(GTARS_ERR_INTERNAL, std::string("internal error: ") + e.what());
    }
}
}
using gtars::cfg_get;
using gtars::fail;

namespace {

// Host threads worth starting: hardware threads, capped by the container's CPU quota (cgroup v2 cpu.max) and by `cap`, and --
// when this process is one of several ranks of a launcher on this node (LOCAL_WORLD_SIZE, set by torch.distributed.run) -- its
// share of them: the fragment pipeline is host-bound, and eight ranks that each start every thread the node has fight over the
// same cores.  GTARS_HOST_THREADS overrides (a launcher may set it per rank).
unsigned host_thread_budget(unsigned cap) {
    if (const char *e = cfg_get("GTARS_HOST_THREADS")) return (unsigned)std::max(1, atoi(e));
    unsigned nt = std::thread::hardware_concurrency();
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        long long quota = 0, period = 0;
        if (fscanf(f, "%lld %lld", &quota, &period) == 2 && quota > 0 && period > 0)
            nt = std::min<unsigned>(nt, (unsigned)((quota + period - 1) / period));
        fclose(f);
    }
    if (const char *e = getenv("LOCAL_WORLD_SIZE")) {
        const int ranks = atoi(e);
        if (ranks > 1) nt = std::max(1u, nt / (unsigned)ranks);
    }
    return std::max(1u, std::min(nt, cap));
}

}  // namespace

namespace {

// ------------------------------------------------------------------ file IO

bool ends_with(const std::string &s, const char *suf) {
    const size_t n = strlen(suf);
    return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
}

std::string extension_of(const std::string &path) {
    // std::path::Path::extension of the final component
    const size_t slash = path.find_last_of('/');
    const std::string base = slash == std::string::npos ? path : path.substr(slash + 1);
    const size_t dot = base.find_last_of('.');
    if (dot == std::string::npos || dot == 0) return "";
    return base.substr(dot + 1);
}

std::string file_stem(const std::string &path) {
    const size_t slash = path.find_last_of('/');
    const std::string base = slash == std::string::npos ? path : path.substr(slash + 1);
    const size_t dot = base.find_last_of('.');
    if (dot == std::string::npos || dot == 0) return base;
    return base.substr(0, dot);
}

std::string parent_dir(const std::string &path) {
    const size_t slash = path.find_last_of('/');
    if (slash == std::string::npos) return "";
    if (slash == 0) return "/";
    return path.substr(0, slash);
}

std::string base_name(const std::string &path) {
    const size_t slash = path.find_last_of('/');
    return slash == std::string::npos ? path : path.substr(slash + 1);
}

bool is_regular_file(const std::string &p) {
    struct stat st;
    return stat(p.c_str(), &st) == 0 && S_ISREG(st.st_mode);
}

// get_dynamic_reader (gtars-core/src/utils.rs:115-126): gzip iff the extension is "gz"
// (MultiGzDecoder: concatenated members, which zlib's gzread handles too).
// The compressed file is read whole and inflated by zlib's inflate() straight into the result (round 5).  Round 4 went through
// gzopen / gzread + std::string::append: on a 1-MB fragment file that layer costs half as much again as the inflate itself
// (19-21 ms against 13.7 for 3.5 MB of text on this image's zlib 1.2.11: the gz layer copies every byte out of its own buffer, and
// the string grows by reallocation) -- and inflate is what bounds the fused fragment pipeline now that its parse runs on the GPU.
// Behaviour: concatenated members are decoded one after the other (flate2's MultiGzDecoder, utils.rs:115-126), bytes behind the
// last member that do not start another one are an error as in the reference ("invalid gzip header"; gzread would ignore them), a
// file without the gzip magic is passed through as it is, the CRC and length of every member are checked (inflate does, with
// the gzip wrapper).
// the file's bytes, followed by 16 zero bytes the decoders may read into (inflate_fast.h); n = the file's length
bool read_file_padded(const std::string &path, std::string &raw, size_t &n) {
    raw.clear();
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return false;
    struct stat sb;
    if (fstat(fileno(f), &sb) == 0 && sb.st_size > 0) raw.reserve((size_t)sb.st_size + 17);
    char buf[1 << 16];
    size_t k;
    while ((k = fread(buf, 1, sizeof buf, f)) > 0) raw.append(buf, k);
    fclose(f);
    n = raw.size();
    raw.append(16, '\0');
    return true;
}

// A gzip file inflated member by member with RAW inflate -- no CRC on the host: zlib's crc32 is a quarter of its inflate time
// (1.8 of 7.4 ms per 3.5 MB of text on the GPU box's host), and the device path of the fused fragment pipeline ships the inflated
// bytes to the GPU anyway, which checks every member's CRC-32 there (fragparse.hip).  The gzip framing (RFC 1952: header with its
// optional fields -- bgzip's extra field, names, comments, a header CRC --, deflate stream, CRC-32 + ISIZE trailer) is parsed here;
// ISIZE is checked here.  -> false for anything but a clean sequence of members (not gzip, truncated, garbage behind the last
// member, a length that does not match ...): the caller then reads the file with read_all, i.e. zlib's own checks and messages.
// `p` must be readable for 16 bytes beyond n (read_file_padded).  The deflate streams are decoded by inflate_fast.h (round 5; zlib's
// raw inflate with GTARS_ZLIB_INFLATE, the A/B switch).
// Out: std::string, or the device path's pinned TextBuf.
template <class Out>
bool inflate_gzip_members_raw(const unsigned char *p, size_t n, Out &out, std::vector<gtars::FragGzMember> &members) {
    out.clear();
    members.clear();
    if (n < 18 || p[0] != 0x1f || p[1] != 0x8b) return false;
    size_t guess = n * 4;
    {
        const unsigned char *t = p + n - 4;
        const size_t isize = (size_t)t[0] | ((size_t)t[1] << 8) | ((size_t)t[2] << 16) | ((size_t)t[3] << 24);
        // (trusted only up to 16x the compressed size: a crafted trailer must not pin gigabytes per loader thread before a byte is
        // decoded -- the output grows on demand beyond the guess)
        if (isize >= n / 2 && isize <= n * 1024) guess = std::min<size_t>(isize, 16 * n + (64u << 10));
    }
    out.resize(guess + 512);
    const bool use_zlib = cfg_get("GTARS_ZLIB_INFLATE") != nullptr;
    z_stream z;
    memset(&z, 0, sizeof z);
    if (use_zlib && inflateInit2(&z, -MAX_WBITS) != Z_OK) return false;
    struct End {
        z_stream &z;
        bool on;
        ~End() {
            if (on) inflateEnd(&z);
        }
    } end{z, use_zlib};
    size_t at = 0, out_done = 0;
    while (at < n) {
        if (n - at < 18 || p[at] != 0x1f || p[at + 1] != 0x8b || p[at + 2] != 8) return false;  // (also: bytes behind the last member)
        const unsigned flg = p[at + 3];
        if (flg & 0xE0) return false;  // reserved bits
        size_t h = at + 10;
        if (flg & 4) {  // FEXTRA
            if (h + 2 > n) return false;
            h += 2 + ((size_t)p[h] | ((size_t)p[h + 1] << 8));
        }
        for (unsigned bit : {8u, 16u})  // FNAME, FCOMMENT: zero-terminated
            if (flg & bit) {
                while (h < n && p[h]) ++h;
                ++h;
            }
        if (flg & 2) {  // FHCRC: the low half of the header's CRC-32 (flate2 and zlib both check it)
            if (h + 2 > n) return false;
            const uint32_t want = (uint32_t)p[h] | ((uint32_t)p[h + 1] << 8);
            if ((crc32(crc32(0L, Z_NULL, 0), p + at, (uInt)(h - at)) & 0xFFFFu) != want) return false;
            h += 2;
        }
        if (h + 8 > n) return false;
        const size_t member_out0 = out_done;
        size_t t;  // the trailer
        if (!use_zlib) {
            size_t used = 0;
            if (!gtars::fastinf::inflate_raw(p + h, n - h, &used, out, out_done)) return false;
            t = h + used;
        } else {
            if (inflateReset(&z) != Z_OK) return false;
            z.next_in = (Bytef *)(p + h);
            size_t in_left = n - h;
            z.avail_in = (uInt)std::min<size_t>(in_left, 0x7FFFFFFFu);
            for (;;) {
                if (out_done == out.size()) out.resize(out.size() + out.size() / 2 + (1 << 16));
                z.next_out = (Bytef *)&out[out_done];
                const size_t room = std::min<size_t>(out.size() - out_done, 0x7FFFFFFFu);
                z.avail_out = (uInt)room;
                const uInt in_before = z.avail_in;
                const int r = inflate(&z, Z_NO_FLUSH);
                out_done += room - z.avail_out;
                in_left -= in_before - z.avail_in;
                if (r == Z_STREAM_END) break;
                if (r != Z_OK && !(r == Z_BUF_ERROR && z.avail_out == 0)) return false;
                if (z.avail_in == 0) {
                    if (!in_left) return false;  // the input ends inside the member
                    z.avail_in = (uInt)std::min<size_t>(in_left, 0x7FFFFFFFu);
                }
            }
            t = n - in_left;
        }
        if (t + 8 > n) return false;
        const uint32_t crc = (uint32_t)p[t] | ((uint32_t)p[t + 1] << 8) | ((uint32_t)p[t + 2] << 16) | ((uint32_t)p[t + 3] << 24);
        const uint32_t isize = (uint32_t)p[t + 4] | ((uint32_t)p[t + 5] << 8) | ((uint32_t)p[t + 6] << 16) | ((uint32_t)p[t + 7] << 24);
        if (isize != (uint32_t)(out_done - member_out0)) return false;
        members.push_back(gtars::FragGzMember{member_out0, out_done - member_out0, crc});
        at = t + 8;
    }
    out.resize(out_done);
    return true;
}

bool read_all(const std::string &path, std::string &out, std::string &err) {
    out.clear();
    std::string raw;
    size_t n_raw = 0;
    if (!read_file_padded(path, raw, n_raw)) {
        err = "Failed to open file: \"" + path + "\": " + strerror(errno);
        return false;
    }
    if (extension_of(path) != "gz" || n_raw < 2 || (unsigned char)raw[0] != 0x1f || (unsigned char)raw[1] != 0x8b) {
        raw.resize(n_raw);
        out.swap(raw);  // plain text (or a ".gz" without the magic: zlib's transparent mode)
        return true;
    }
    if (!cfg_get("GTARS_ZLIB_INFLATE")) {
        // the whole-buffer decoder (inflate_fast.h) + zlib's crc32 over every member; whatever it refuses, and any mismatch, goes
        // through zlib below -- its checks, its messages
        std::vector<gtars::FragGzMember> members;
        if (inflate_gzip_members_raw((const unsigned char *)raw.data(), n_raw, out, members)) {
            bool ok = true;
            for (const gtars::FragGzMember &g : members) {
                uLong c = crc32(0L, Z_NULL, 0);
                for (uint64_t at = 0; at < g.len; at += 1u << 30)
                    c = crc32(c, (const Bytef *)out.data() + g.off + at, (uInt)std::min<uint64_t>(g.len - at, 1u << 30));
                ok = ok && (uint32_t)c == g.crc;
            }
            if (ok) return true;
        }
        out.clear();
    }
    raw.resize(n_raw);
    // the last member's ISIZE (uncompressed length mod 2^32) sizes the result; several members, or a lie, just grow it
    size_t guess = raw.size() * 4;
    if (raw.size() >= 18) {
        const unsigned char *t = (const unsigned char *)raw.data() + raw.size() - 4;
        const size_t isize = (size_t)t[0] | ((size_t)t[1] << 8) | ((size_t)t[2] << 16) | ((size_t)t[3] << 24);
        if (isize >= raw.size() / 2 && isize <= raw.size() * 1024) guess = std::min<size_t>(isize, 16 * raw.size() + (64u << 10));
    }
    out.resize(guess + 64);
    z_stream z;
    memset(&z, 0, sizeof z);
    if (inflateInit2(&z, 16 + MAX_WBITS) != Z_OK) {
        err = "gzip read error: cannot initialise zlib";
        return false;
    }
    z.next_in = (Bytef *)raw.data();
    z.avail_in = (uInt)std::min<size_t>(raw.size(), 0x7FFFFFFFu);
    size_t in_done = 0, out_done = 0;
    for (;;) {
        if (out_done == out.size()) out.resize(out.size() + out.size() / 2 + (1 << 16));
        z.next_out = (Bytef *)&out[out_done];
        const size_t room = std::min<size_t>(out.size() - out_done, 0x7FFFFFFFu);
        z.avail_out = (uInt)room;
        const uInt in_before = z.avail_in;
        const int r = inflate(&z, Z_NO_FLUSH);
        out_done += room - z.avail_out;
        in_done += in_before - z.avail_in;
        if (r == Z_STREAM_END) {
            // another member?  flate2's MultiGzDecoder (what get_dynamic_reader wraps, utils.rs:115-126) parses a header out of
            // whatever follows a member's trailer and FAILS on bytes that are none ("invalid gzip header"); zlib's gzread would
            // ignore them -- round 5 did, round 6 follows the reference.
            if (in_done == raw.size()) break;
            if (raw.size() - in_done >= 2 && (unsigned char)raw[in_done] == 0x1f && (unsigned char)raw[in_done + 1] == 0x8b) {
                inflateReset(&z);
                z.next_in = (Bytef *)raw.data() + in_done;
                z.avail_in = (uInt)std::min<size_t>(raw.size() - in_done, 0x7FFFFFFFu);
                continue;
            }
            err = "gzip read error: invalid gzip header";
            inflateEnd(&z);
            return false;
        }
        if (r == Z_OK || (r == Z_BUF_ERROR && z.avail_out == 0)) {
            if (z.avail_in == 0 && in_done < raw.size()) {  // (files beyond 2 GiB compressed: the next piece)
                z.next_in = (Bytef *)raw.data() + in_done;
                z.avail_in = (uInt)std::min<size_t>(raw.size() - in_done, 0x7FFFFFFFu);
                continue;
            }
            if (z.avail_in == 0 && z.avail_out != 0) {  // the input ended inside a member
                err = "gzip read error: unexpected end of file";
                inflateEnd(&z);
                return false;
            }
            continue;
        }
        err = std::string("gzip read error: ") + (z.msg ? z.msg : zError(r));
        if (r == Z_BUF_ERROR) err = "gzip read error: unexpected end of file";
        inflateEnd(&z);
        return false;
    }
    inflateEnd(&z);
    out.resize(out_done);
    return true;
}

// BufRead::lines(): split on '\n', strip one trailing '\r'; no empty last line after a final '\n'
struct LineIter {
    const std::string &s;
    size_t pos = 0;
    explicit LineIter(const std::string &str) : s(str) {}
    bool next(std::string &line) {
        if (pos >= s.size()) return false;
        size_t nl = s.find('\n', pos);
        size_t end = nl == std::string::npos ? s.size() : nl;
        size_t e2 = end;
        if (nl != std::string::npos && e2 > pos && s[e2 - 1] == '\r') --e2;
        line.assign(s, pos, e2 - pos);
        pos = nl == std::string::npos ? s.size() : nl + 1;
        return true;
    }
};

std::vector<std::string> split_char(const std::string &s, char sep) {
    std::vector<std::string> out;
    size_t pos = 0;
    for (;;) {
        size_t k = s.find(sep, pos);
        if (k == std::string::npos) {
            out.push_back(s.substr(pos));
            break;
        }
        out.push_back(s.substr(pos, k - pos));
        pos = k + 1;
    }
    return out;
}

// str::split_whitespace (ASCII subset: BED files)
std::vector<std::string> split_ws(const std::string &s) {
    std::vector<std::string> out;
    size_t i = 0;
    while (i < s.size()) {
        while (i < s.size() && isspace((unsigned char)s[i])) ++i;
        size_t j = i;
        while (j < s.size() && !isspace((unsigned char)s[j])) ++j;
        if (j > i) out.push_back(s.substr(i, j - i));
        i = j;
    }
    return out;
}

// str::parse::<u32>(): optional '+', ASCII digits, must fit
bool parse_u32(const std::string &s, uint32_t &out) {
    size_t i = 0;
    if (!s.empty() && s[0] == '+') i = 1;
    if (i >= s.size()) return false;
    uint64_t v = 0;
    for (; i < s.size(); ++i) {
        if (s[i] < '0' || s[i] > '9') return false;
        v = v * 10 + (uint64_t)(s[i] - '0');
        if (v > 0xFFFFFFFFull) return false;
    }
    out = (uint32_t)v;
    return true;
}

// str::parse::<i32>()
bool parse_i32(const std::string &s, int32_t &out) {
    size_t i = 0;
    bool neg = false;
    if (!s.empty() && (s[0] == '+' || s[0] == '-')) {
        neg = s[0] == '-';
        i = 1;
    }
    if (i >= s.size()) return false;
    int64_t v = 0;
    for (; i < s.size(); ++i) {
        if (s[i] < '0' || s[i] > '9') return false;
        v = v * 10 + (s[i] - '0');
        if (v > 2147483648ll) return false;
    }
    if (neg) v = -v;
    if (v < -2147483648ll || v > 2147483647ll) return false;
    out = (int32_t)v;
    return true;
}

bool parse_f64(const std::string &s_in, double &out) {
    std::string s = s_in;
    while (!s.empty() && isspace((unsigned char)s.back())) s.pop_back();
    size_t b = 0;
    while (b < s.size() && isspace((unsigned char)s[b])) ++b;
    s = s.substr(b);
    if (s.empty() || s.find_first_of("xX") != std::string::npos) return false;
    char *end = nullptr;
    errno = 0;
    out = strtod(s.c_str(), &end);
    return end && *end == '\0';
}

char *dup_cstr(const std::string &s) {
    char *p = (char *)malloc(s.size() + 1);
    if (p) memcpy(p, s.c_str(), s.size() + 1);
    return p;
}

// string -> dense id dictionary in first-seen order
struct Dict {
    std::unordered_map<std::string, uint32_t> ids;
    std::vector<std::string> names;
    uint32_t get_or_add(const std::string &s) {
        auto it = ids.find(s);
        if (it != ids.end()) return it->second;
        const uint32_t id = (uint32_t)names.size();
        ids.emplace(s, id);
        names.push_back(s);
        return id;
    }
    int64_t find(const std::string &s) const {
        auto it = ids.find(s);
        return it == ids.end() ? -1 : (int64_t)it->second;
    }
};

// open-addressing string -> id table; keys are views into `names` (stable: deque)
struct ViewDict {
    std::deque<std::string> names;
    std::vector<uint32_t> slots;  // id + 1, 0 = empty
    size_t mask = 0;
    static uint64_t hash(const char *p, size_t n) {
        uint64_t h = 1469598103934665603ull;
        for (size_t i = 0; i < n; ++i) h = (h ^ (unsigned char)p[i]) * 1099511628211ull;
        return h ^ (h >> 29);
    }
    void grow() {
        const size_t cap = slots.empty() ? 64 : slots.size() * 2;
        slots.assign(cap, 0);
        mask = cap - 1;
        for (uint32_t id = 0; id < names.size(); ++id) {
            size_t k = hash(names[id].data(), names[id].size()) & mask;
            while (slots[k]) k = (k + 1) & mask;
            slots[k] = id + 1;
        }
    }
    uint32_t get_or_add(const char *p, size_t n) {
        if (names.size() * 2 >= slots.size()) grow();
        size_t k = hash(p, n) & mask;
        while (slots[k]) {
            const std::string &nm = names[slots[k] - 1];
            if (nm.size() == n && memcmp(nm.data(), p, n) == 0) return slots[k] - 1;
            k = (k + 1) & mask;
        }
        names.emplace_back(p, n);
        slots[k] = (uint32_t)names.size();
        return (uint32_t)names.size() - 1;
    }
};

inline bool is_ws(char ch) { return ch == ' ' || (ch >= '\t' && ch <= '\r'); }  // isspace() in the C locale

// str::parse::<u32>(): optional '+', ASCII digits, must fit
inline bool parse_u32_view(const char *p, size_t n, uint32_t &out) {
    size_t i = (n && p[0] == '+') ? 1 : 0;
    if (i >= n) return false;
    uint64_t v = 0;
    for (; i < n; ++i) {
        const unsigned d = (unsigned char)p[i] - '0';
        if (d > 9) return false;
        v = v * 10 + d;
        if (v > 0xFFFFFFFFull) return false;
    }
    out = (uint32_t)v;
    return true;
}


}  // namespace

// =============================================================== RegionSet

struct gtars_regionset {
    Dict chroms;
    std::vector<uint32_t> chrom_ids, starts, ends;
    std::string rest_arena;          // NUL-terminated `rest` strings back to back
    std::vector<uint64_t> rest_off;  // offset of region i's rest in the arena
    std::vector<uint8_t> has_rest;
    std::string header;
    bool has_header = false;
    size_t size() const { return starts.size(); }
};

namespace {

struct ParsedRegion {
    std::string chr;
    uint32_t start, end;
    std::string rest;
    bool has_rest;
};

void regionset_assign(gtars_regionset *rs, std::vector<ParsedRegion> &regs) {
    const size_t n = regs.size();
    rs->chrom_ids.resize(n);
    rs->starts.resize(n);
    rs->ends.resize(n);
    rs->rest_off.assign(n, 0);
    rs->has_rest.resize(n);
    for (size_t i = 0; i < n; ++i) {
        rs->chrom_ids[i] = rs->chroms.get_or_add(regs[i].chr);
        rs->starts[i] = regs[i].start;
        rs->ends[i] = regs[i].end;
        rs->has_rest[i] = regs[i].has_rest;
        if (regs[i].has_rest) {
            rs->rest_off[i] = rs->rest_arena.size();
            rs->rest_arena.append(regs[i].rest);
            rs->rest_arena.push_back('\0');
        }
    }
}

// encode the chromosomes of `q` in the id space of dictionary `d` (unknown -> GTARS_UNKNOWN_CHROM)
std::vector<uint32_t> translate_chroms(const gtars_regionset *q, const Dict &d) {
    std::vector<uint32_t> map(q->chroms.names.size());
    for (size_t i = 0; i < map.size(); ++i) {
        const int64_t id = d.find(q->chroms.names[i]);
        map[i] = id < 0 ? GTARS_UNKNOWN_CHROM : (uint32_t)id;
    }
    std::vector<uint32_t> out(q->size());
    for (size_t i = 0; i < out.size(); ++i) out[i] = map[q->chrom_ids[i]];
    return out;
}

}  // namespace

extern "C" uint32_t gtars_host_threads(uint32_t cap) { return host_thread_budget(cap ? cap : 0xFFFFFFFFu); }

extern "C" gtars_status gtars_read_file(const char *path, char **out, uint64_t *out_n) {
    if (!path || !out || !out_n) return fail(GTARS_ERR_INVALID_ARG, "gtars_read_file: null argument");
    *out = nullptr;
    *out_n = 0;
    std::string data, err;
    if (!read_all(path, data, err)) return fail(GTARS_ERR_IO, err);
    char *buf = (char *)malloc(data.size() + 1);
    if (!buf) return fail(GTARS_ERR_INTERNAL, "gtars_read_file: out of memory");
    memcpy(buf, data.data(), data.size());
    buf[data.size()] = 0;
    *out = buf;
    *out_n = data.size();
    return GTARS_OK;
}

extern "C" {

// RegionSet::try_from(&Path) -- gtars-core/src/models/region_set.rs:52-186
static gtars_status gtars_regionset_from_bed_impl(const char *path, gtars_regionset_t **out) {
    if (!path || !out) return fail(GTARS_ERR_INVALID_ARG, "NULL argument");
    *out = nullptr;
    const std::string p(path);
    if (!is_regular_file(p))
        return fail(GTARS_ERR_IO, "The file " + p + " does not exist or is not a file (http input is not supported)");
    std::string data, err;
    if (!read_all(p, data, err)) return fail(GTARS_ERR_IO, err);

    // The text is cut at line ends into one chunk per host thread and scanned in place (no per-line or
    // per-field allocation); `rest` stays a view into the text until the final arena is written.
    struct Chunk {
        std::vector<uint32_t> c, s, e, rest_len;
        std::vector<uint64_t> rest_off;
        ViewDict chroms;
        std::string header;
        int err = 0;  // 1: start, 2: end
        std::string err_line;
    };
    auto parse_chunk = [&data](size_t lo, size_t hi, bool file_start, Chunk &out) {
        const char *base = data.data();
        const char *p = base + lo, *end = base + hi;
        const size_t guess = (hi - lo) / 24 + 16;
        out.c.reserve(guess); out.s.reserve(guess); out.e.reserve(guess); out.rest_off.reserve(guess); out.rest_len.reserve(guess);
        bool first_line = file_start;
        const char *last_p = nullptr; size_t last_n = 0; uint32_t last_id = 0;
        while (p < end) {
            const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
            const char *le = nl ? nl : end;
            const char *next = nl ? nl + 1 : end;
            if (nl && le > p && le[-1] == '\r') --le;
            const size_t n = (size_t)(le - p);
            if ((n >= 7 && memcmp(p, "browser", 7) == 0) || (n >= 5 && memcmp(p, "track", 5) == 0) || (n && *p == '#')) {
                out.header.append(p, n);
                first_line = false;
                p = next;
                continue;
            }
            const char *t1 = (const char *)memchr(p, '\t', n);
            const char *t2 = t1 ? (const char *)memchr(t1 + 1, '\t', (size_t)(le - t1 - 1)) : nullptr;
            const char *t3 = t2 ? (const char *)memchr(t2 + 1, '\t', (size_t)(le - t2 - 1)) : nullptr;
            const char *f2e = t3 ? t3 : le;
            uint32_t sv = 0, ev = 0;
            if (first_line) {
                first_line = false;
                // column headers like `chr start end ...` without '#'
                if (t2 && !parse_u32_view(t1 + 1, (size_t)(t2 - t1 - 1), sv)) {
                    out.header.append(p, n);
                    p = next;
                    continue;
                }
            }
            if (!t2 || !parse_u32_view(t1 + 1, (size_t)(t2 - t1 - 1), sv)) {
                out.err = 1; out.err_line.assign(p, n); return;
            }
            if (!parse_u32_view(t2 + 1, (size_t)(f2e - t2 - 1), ev)) {
                out.err = 2; out.err_line.assign(p, n); return;
            }
            const size_t cn = (size_t)(t1 - p);
            if (!(last_p && last_n == cn && memcmp(last_p, p, cn) == 0)) {
                last_id = out.chroms.get_or_add(p, cn); last_p = p; last_n = cn;
            }
            out.c.push_back(last_id); out.s.push_back(sv); out.e.push_back(ev);
            out.rest_off.push_back(t3 ? (uint64_t)(t3 + 1 - base) : 0);
            out.rest_len.push_back(t3 ? (uint32_t)(le - t3 - 1) : 0u);
            p = next;
        }
    };
    unsigned nt = host_thread_budget(32);
    nt = (unsigned)std::min<size_t>(nt, data.size() / (1u << 20) + 1);
    std::vector<size_t> cut(nt + 1, data.size());
    cut[0] = 0;
    for (unsigned i = 1; i < nt; ++i) {
        const size_t nl = data.find('\n', data.size() / nt * i);
        cut[i] = nl == std::string::npos ? data.size() : nl + 1;
    }
    for (unsigned i = 1; i <= nt; ++i) cut[i] = std::max(cut[i], cut[i - 1]);
    std::vector<Chunk> chunks(nt);
    {
        std::vector<std::thread> th;
        for (unsigned i = 1; i < nt; ++i) th.emplace_back([&, i] { parse_chunk(cut[i], cut[i + 1], false, chunks[i]); });
        parse_chunk(cut[0], cut[1], true, chunks[0]);
        for (auto &t : th) t.join();
    }
    std::string header;
    size_t n = 0;
    for (unsigned i = 0; i < nt; ++i) {
        if (chunks[i].err)
            return fail(GTARS_ERR_PARSE, std::string("Error in parsing ") + (chunks[i].err == 1 ? "start" : "end") +
                                             " position: \"" + chunks[i].err_line + "\"");
        header += chunks[i].header;
        n += chunks[i].c.size();
    }
    if (n == 0) return fail(GTARS_ERR_EMPTY, "EmptyRegionSet: " + p);
    // chromosome names -> rank in byte order; RegionSet::sort (region_set.rs:502-505) is stable by (chr, start)
    ViewDict all;
    std::vector<std::vector<uint32_t>> cmap(nt);
    for (unsigned i = 0; i < nt; ++i)
        for (const std::string &nm : chunks[i].chroms.names) cmap[i].push_back(all.get_or_add(nm.data(), nm.size()));
    const uint32_t n_names = (uint32_t)all.names.size();
    std::vector<uint32_t> by_name(n_names), rank(n_names);
    std::iota(by_name.begin(), by_name.end(), 0u);
    std::sort(by_name.begin(), by_name.end(), [&](uint32_t a, uint32_t b) { return all.names[a] < all.names[b]; });
    for (uint32_t r = 0; r < n_names; ++r) rank[by_name[r]] = r;
    std::vector<uint32_t> col_c(n), col_s(n), col_e(n), col_rl(n);
    std::vector<uint64_t> col_ro(n);
    {
        size_t o = 0;
        for (unsigned i = 0; i < nt; ++i) {
            const Chunk &ck = chunks[i];
            for (size_t k = 0; k < ck.c.size(); ++k, ++o) {
                col_c[o] = rank[cmap[i][ck.c[k]]];
                col_s[o] = ck.s[k];
                col_e[o] = ck.e[k];
                col_ro[o] = ck.rest_off[k];
                col_rl[o] = ck.rest_len[k];
            }
        }
    }
    // stable LSD radix sort of the row indices: start (two 16-bit passes), then chromosome rank
    std::vector<uint32_t> perm(n), tmp(n);
    std::iota(perm.begin(), perm.end(), 0u);
    {
        std::vector<size_t> cnt;
        auto pass = [&](auto key_of, size_t n_keys) {
            cnt.assign(n_keys + 1, 0);
            for (size_t i = 0; i < n; ++i) ++cnt[(size_t)key_of(perm[i]) + 1];
            for (size_t k = 0; k < n_keys; ++k) cnt[k + 1] += cnt[k];
            for (size_t i = 0; i < n; ++i) tmp[cnt[key_of(perm[i])]++] = perm[i];
            perm.swap(tmp);
        };
        pass([&](uint32_t i) { return col_s[i] & 0xFFFFu; }, 1u << 16);
        pass([&](uint32_t i) { return col_s[i] >> 16; }, 1u << 16);
        pass([&](uint32_t i) { return col_c[i]; }, n_names);
    }
    auto *rs = new gtars_regionset();
    for (uint32_t r = 0; r < n_names; ++r) rs->chroms.get_or_add(all.names[by_name[r]]);  // id = rank
    rs->chrom_ids.resize(n); rs->starts.resize(n); rs->ends.resize(n); rs->rest_off.assign(n, 0); rs->has_rest.resize(n);
    size_t arena = 0;
    for (size_t i = 0; i < n; ++i) arena += col_rl[i] ? col_rl[i] + 1 : 0;
    rs->rest_arena.reserve(arena);
    for (size_t i = 0; i < n; ++i) {
        const uint32_t k = perm[i];
        rs->chrom_ids[i] = col_c[k];
        rs->starts[i] = col_s[k];
        rs->ends[i] = col_e[k];
        rs->has_rest[i] = col_rl[k] != 0;
        if (col_rl[k]) {
            rs->rest_off[i] = rs->rest_arena.size();
            rs->rest_arena.append(data.data() + col_ro[k], col_rl[k]);
            rs->rest_arena.push_back('\0');
        }
    }
    rs->header = header;
    rs->has_header = !header.empty();
    *out = rs;
    return GTARS_OK;
}

// From<Vec<Region>> -- region_set.rs:212-220 (not sorted)
gtars_status gtars_regionset_from_arrays(const char *const *chrs, const uint32_t *starts, const uint32_t *ends,
                                         const char *const *rest, uint64_t n, gtars_regionset_t **out) {
    if (!out || (n && (!chrs || !starts || !ends))) return fail(GTARS_ERR_INVALID_ARG, "NULL argument");
    *out = nullptr;
    std::vector<ParsedRegion> regs(n);
    for (uint64_t i = 0; i < n; ++i) {
        if (!chrs[i]) return fail(GTARS_ERR_INVALID_ARG, "NULL chromosome name");
        regs[i].chr = chrs[i];
        regs[i].start = starts[i];
        regs[i].end = ends[i];
        regs[i].has_rest = rest && rest[i] && rest[i][0];
        if (regs[i].has_rest) regs[i].rest = rest[i];
    }
    auto *rs = new gtars_regionset();
    regionset_assign(rs, regs);
    *out = rs;
    return GTARS_OK;
}

void gtars_regionset_free(gtars_regionset_t *rs) { delete rs; }
uint64_t gtars_regionset_len(const gtars_regionset_t *rs) { return rs ? rs->size() : 0; }
const char *gtars_regionset_header(const gtars_regionset_t *rs) {
    return rs && rs->has_header ? rs->header.c_str() : nullptr;
}
uint32_t gtars_regionset_n_chrom(const gtars_regionset_t *rs) { return rs ? (uint32_t)rs->chroms.names.size() : 0; }
const char *gtars_regionset_chrom_name(const gtars_regionset_t *rs, uint32_t id) {
    return rs && id < rs->chroms.names.size() ? rs->chroms.names[id].c_str() : nullptr;
}
const uint32_t *gtars_regionset_chrom_ids(const gtars_regionset_t *rs) { return rs ? rs->chrom_ids.data() : nullptr; }
const uint32_t *gtars_regionset_starts(const gtars_regionset_t *rs) { return rs ? rs->starts.data() : nullptr; }
const uint32_t *gtars_regionset_ends(const gtars_regionset_t *rs) { return rs ? rs->ends.data() : nullptr; }
const char *gtars_regionset_rest(const gtars_regionset_t *rs, uint64_t i) {
    return rs && i < rs->size() && rs->has_rest[i] ? rs->rest_arena.data() + rs->rest_off[i] : nullptr;
}

// generate_region_to_id_map (gtars-core/src/utils.rs:202-214): dense ids in first-seen order over the whole Region
// (chr, start, end, rest) -- what gtars-scoring's ConsensusSet stores as the interval payload (files.rs:60-83)
gtars_status gtars_regionset_dense_ids(const gtars_regionset_t *rs, uint32_t **out_ids, uint32_t *out_n_ids) {
    return gtars::guarded([&]() -> gtars_status {
        if (!rs || !out_ids) return fail(GTARS_ERR_INVALID_ARG, "NULL argument");
        *out_ids = nullptr;
        const size_t n = rs->size();
        uint32_t *ids = (uint32_t *)malloc(std::max<size_t>(n, 1) * sizeof(uint32_t));
        if (!ids) return fail(GTARS_ERR_INTERNAL, "out of host memory");
        struct Key {
            uint32_t c, s, e;
            bool has;
            std::string_view rest;
            bool operator==(const Key &o) const { return c == o.c && s == o.s && e == o.e && has == o.has && rest == o.rest; }
        };
        struct Hash {
            size_t operator()(const Key &k) const {
                uint64_t h = ((uint64_t)k.c * 0x9E3779B97F4A7C15ull) ^ (((uint64_t)k.s << 32) | k.e);
                h ^= std::hash<std::string_view>()(k.rest) + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2) + (k.has ? 1 : 0);
                return (size_t)h;
            }
        };
        std::unordered_map<Key, uint32_t, Hash> seen;
        seen.reserve(n * 2 + 16);
        for (size_t i = 0; i < n; ++i) {
            Key k{rs->chrom_ids[i], rs->starts[i], rs->ends[i], rs->has_rest[i] != 0,
                  rs->has_rest[i] ? std::string_view(rs->rest_arena.data() + rs->rest_off[i]) : std::string_view()};
            ids[i] = seen.emplace(k, (uint32_t)seen.size()).first->second;
        }
        *out_ids = ids;
        if (out_n_ids) *out_n_ids = (uint32_t)seen.size();
        return GTARS_OK;
    });
}

}  // extern "C"

namespace {

// IndexedRegionSet::new(other) on the GPU (build_indexed_overlapper, multi_chrom_overlapper.rs:325-351)
struct ScopedIndex {
    gtars_index_t *ix = nullptr;
    ~ScopedIndex() { gtars_index_free(ix); }
};

// count / any / sorted-unique indices do not expose the enumeration order, so whatever `kind` the caller
// names (IndexedRegionSet defaults to AIList, indexed_region_set.rs:111-113) the Bits form is built: same hit
// sets, no AIList decomposition at build, and the blocked structure's kernels (k_count_lds, k_tok_lds).
gtars_status build_other_index(const gtars_regionset *other, int kind, ScopedIndex &out) {
    if (kind != GTARS_KIND_BITS && kind != GTARS_KIND_AILIST) return fail(GTARS_ERR_INVALID_ARG, "unknown index kind");
    kind = GTARS_KIND_BITS;
    return gtars_index_build(other->chrom_ids.data(), other->starts.data(), other->ends.data(), nullptr,
                             other->size(), (uint32_t)other->chroms.names.size(), kind, &out.ix);
}

}  // namespace

extern "C" {

gtars_status gtars_regionset_count_overlaps(const gtars_regionset_t *self, const gtars_regionset_t *other,
                                            int kind, int has_min, int32_t min_overlap, uint32_t *counts) {
    if (!self || !other) return fail(GTARS_ERR_INVALID_ARG, "NULL region set");
    ScopedIndex ix;
    gtars_status st = build_other_index(other, kind, ix);
    if (st) return st;
    const std::vector<uint32_t> qc = translate_chroms(self, other->chroms);
    return gtars_count_overlaps(ix.ix, qc.data(), self->starts.data(), self->ends.data(), self->size(), has_min,
                                min_overlap, counts);
}

gtars_status gtars_regionset_any_overlaps(const gtars_regionset_t *self, const gtars_regionset_t *other, int kind,
                                          int has_min, int32_t min_overlap, uint8_t *out) {
    if (!self || !other) return fail(GTARS_ERR_INVALID_ARG, "NULL region set");
    ScopedIndex ix;
    gtars_status st = build_other_index(other, kind, ix);
    if (st) return st;
    const std::vector<uint32_t> qc = translate_chroms(self, other->chroms);
    return gtars_any_overlaps(ix.ix, qc.data(), self->starts.data(), self->ends.data(), self->size(), has_min,
                              min_overlap, out);
}

gtars_status gtars_regionset_find_overlaps(const gtars_regionset_t *self, const gtars_regionset_t *other,
                                           int kind, int has_min, int32_t min_overlap, uint64_t *offsets,
                                           uint32_t **out_idx, uint64_t *out_n) {
    if (!self || !other) return fail(GTARS_ERR_INVALID_ARG, "NULL region set");
    ScopedIndex ix;
    gtars_status st = build_other_index(other, kind, ix);
    if (st) return st;
    const std::vector<uint32_t> qc = translate_chroms(self, other->chroms);
    return gtars_find_overlap_indices(ix.ix, qc.data(), self->starts.data(), self->ends.data(), self->size(),
                                      has_min, min_overlap, offsets, out_idx, out_n);
}

}  // extern "C"

// ================================================================ Tokenizer

struct gtars_tokenizer {
    // Universe (gtars-tokenizers/src/universe/mod.rs:35-42)
    std::vector<std::string> regions;
    std::unordered_map<std::string, uint32_t> region_to_id;
    std::vector<std::string> vocab_order;  // keys of region_to_id in insertion order
    std::unordered_map<uint32_t, std::string> id_to_region;
    std::unordered_map<std::string, std::string> names;
    std::unordered_map<std::string, double> scores;
    bool has_scores = false;
    std::string special[7];  // unk,pad,mask,cls,eos,bos,sep
    int kind = GTARS_KIND_BITS;
    Dict chroms;
    gtars_index_t *index = nullptr;
    uint32_t unk_id = 0;
    // the chromosome dictionary as a device table (fragparse.hip), made by the first fused fragment pipeline call on a device and
    // kept with the tokenizer: building it per call cost 23 ms of a 65-ms call (device allocations synchronise)
    mutable std::mutex frag_mu;
    mutable gtars::FragChroms *frag_chroms = nullptr;
    // the device the tokenizer's index lives on: where the fused fragment pipeline runs (its loaders' pinned blocks, its device
    // threads, this table), whatever the calling thread's current device is
    int device() const {
        const int d = gtars_index_device(index);
        return d >= 0 ? d : gtars::frag_current_device();
    }
    // Built once, on the index's device, and never freed under a running call (round 5 kept one table keyed by the CALLER's current
    // device and freed it when a call arrived from another device -- under a concurrent call that still used it).
    gtars_status device_chroms(gtars::FragChroms **out) const {
        std::lock_guard<std::mutex> lk(frag_mu);
        if (!frag_chroms) {
            const int before = gtars::frag_current_device(), want = device();
            if (before != want) {
                gtars_status st = gtars::frag_select_device(want);
                if (st) return st;
            }
            gtars_status st = gtars::frag_chroms_create(chroms.names, &frag_chroms);
            if (before != want && before >= 0) (void)gtars::frag_select_device(before);
            if (st) return st;
        }
        *out = frag_chroms;
        return GTARS_OK;
    }
};

namespace {

const char *kSpecialSlots[7] = {"unk", "pad", "mask", "cls", "eos", "bos", "sep"};
const char *kSpecialDefaults[7] = {"<unk>", "<pad>", "<mask>", "<cls>", "<eos>", "<bos>", "<sep>"};

// ---- minimal TOML reader for TokenizerConfig (gtars-tokenizers/src/config.rs:27-41):
//   universe = "<path>"            (required)
//   tokenizer_type = "bits"|"ailist"   (optional; anything else is an error, tokenizer.rs:336-341)
//   special_tokens = [ {name="unk", token="<UNKNOWN>"}, ... ]  or  [[special_tokens]] tables
// Unknown keys are ignored (serde default).
struct TomlCfg {
    bool has_universe = false;
    std::string universe;
    bool has_type = false;
    std::string type;
    std::vector<std::pair<std::string, std::string>> specials;  // (name, token)
};

struct TomlParser {
    const std::string &s;
    size_t i = 0;
    std::string err;
    explicit TomlParser(const std::string &str) : s(str) {}

    void skip_ws_inline() {
        while (i < s.size() && (s[i] == ' ' || s[i] == '\t')) ++i;
    }
    void skip_ws_all() {
        for (;;) {
            while (i < s.size() && (s[i] == ' ' || s[i] == '\t' || s[i] == '\r' || s[i] == '\n')) ++i;
            if (i < s.size() && s[i] == '#') {
                while (i < s.size() && s[i] != '\n') ++i;
                continue;
            }
            break;
        }
    }
    bool parse_string(std::string &out) {
        if (i >= s.size()) return false;
        const char q = s[i];
        if (q != '"' && q != '\'') return false;
        ++i;
        out.clear();
        while (i < s.size() && s[i] != q) {
            if (q == '"' && s[i] == '\\' && i + 1 < s.size()) {
                const char c = s[i + 1];
                switch (c) {
                    case 'n': out += '\n'; break;
                    case 't': out += '\t'; break;
                    case 'r': out += '\r'; break;
                    case '"': out += '"'; break;
                    case '\\': out += '\\'; break;
                    default: err = "unsupported escape in TOML string"; return false;
                }
                i += 2;
            } else {
                if (s[i] == '\n') {
                    err = "newline in TOML string";
                    return false;
                }
                out += s[i++];
            }
        }
        if (i >= s.size()) {
            err = "unterminated TOML string";
            return false;
        }
        ++i;
        return true;
    }
    bool parse_key(std::string &k) {
        skip_ws_inline();
        k.clear();
        if (i < s.size() && (s[i] == '"' || s[i] == '\'')) return parse_string(k);
        while (i < s.size() && (isalnum((unsigned char)s[i]) || s[i] == '_' || s[i] == '-')) k += s[i++];
        return !k.empty();
    }
    // skip any value we do not care about (string / number / bool / array / inline table)
    bool skip_value() {
        skip_ws_inline();
        if (i >= s.size()) return false;
        if (s[i] == '"' || s[i] == '\'') {
            std::string t;
            return parse_string(t);
        }
        if (s[i] == '[' || s[i] == '{') {
            const char open = s[i], close = open == '[' ? ']' : '}';
            int depth = 0;
            while (i < s.size()) {
                if (s[i] == '"' || s[i] == '\'') {
                    std::string t;
                    if (!parse_string(t)) return false;
                    continue;
                }
                if (s[i] == '#') {
                    while (i < s.size() && s[i] != '\n') ++i;
                    continue;
                }
                if (s[i] == open) ++depth;
                if (s[i] == close) {
                    --depth;
                    if (depth == 0) {
                        ++i;
                        return true;
                    }
                }
                ++i;
            }
            return false;
        }
        while (i < s.size() && s[i] != '\n' && s[i] != '#' && s[i] != ',' && s[i] != '}' && s[i] != ']') ++i;
        return true;
    }
    bool parse_inline_special(TomlCfg &cfg) {
        // { name = "unk", token = "<X>" }
        if (s[i] != '{') {
            err = "special_tokens entries must be inline tables";
            return false;
        }
        ++i;
        std::string name, token;
        bool hn = false, ht = false;
        for (;;) {
            skip_ws_all();
            if (i < s.size() && s[i] == '}') {
                ++i;
                break;
            }
            std::string k;
            if (!parse_key(k)) {
                err = "bad key in special_tokens entry";
                return false;
            }
            skip_ws_inline();
            if (i >= s.size() || s[i] != '=') {
                err = "expected '=' in special_tokens entry";
                return false;
            }
            ++i;
            skip_ws_inline();
            std::string v;
            if (!parse_string(v)) {
                if (err.empty()) err = "special_tokens values must be strings";
                return false;
            }
            if (k == "name") {
                name = v;
                hn = true;
            } else if (k == "token") {
                token = v;
                ht = true;
            }
            skip_ws_all();
            if (i < s.size() && s[i] == ',') ++i;
        }
        if (!hn || !ht) {
            err = "special_tokens entry needs name and token";
            return false;
        }
        cfg.specials.emplace_back(name, token);
        return true;
    }
    bool parse(TomlCfg &cfg) {
        bool in_special_table = false, in_other_table = false;
        std::string cur_name, cur_token;
        bool cn = false, ct = false;
        auto flush_table = [&]() -> bool {
            if (in_special_table) {
                if (!cn || !ct) {
                    err = "special_tokens entry needs name and token";
                    return false;
                }
                cfg.specials.emplace_back(cur_name, cur_token);
            }
            in_special_table = false;
            cn = ct = false;
            return true;
        };
        for (;;) {
            skip_ws_all();
            if (i >= s.size()) break;
            if (s[i] == '[') {
                if (!flush_table()) return false;
                const bool dbl = i + 1 < s.size() && s[i + 1] == '[';
                i += dbl ? 2 : 1;
                std::string k;
                if (!parse_key(k)) {
                    err = "bad TOML table header";
                    return false;
                }
                skip_ws_inline();
                while (i < s.size() && s[i] == ']') ++i;
                in_special_table = dbl && k == "special_tokens";
                in_other_table = !in_special_table;
                continue;
            }
            std::string k;
            if (!parse_key(k)) {
                err = "bad TOML key";
                return false;
            }
            skip_ws_inline();
            if (i >= s.size() || s[i] != '=') {
                err = "expected '=' after TOML key '" + k + "'";
                return false;
            }
            ++i;
            skip_ws_inline();
            if (in_special_table) {
                std::string v;
                if (!parse_string(v)) {
                    if (err.empty()) err = "special_tokens values must be strings";
                    return false;
                }
                if (k == "name") {
                    cur_name = v;
                    cn = true;
                } else if (k == "token") {
                    cur_token = v;
                    ct = true;
                }
                continue;
            }
            if (in_other_table) {
                if (!skip_value()) {
                    err = "bad TOML value";
                    return false;
                }
                continue;
            }
            if (k == "universe" || k == "tokenizer_type") {
                std::string v;
                if (!parse_string(v)) {
                    if (err.empty()) err = "'" + k + "' must be a string";
                    return false;
                }
                if (k == "universe") {
                    cfg.universe = v;
                    cfg.has_universe = true;
                } else {
                    cfg.type = v;
                    cfg.has_type = true;
                }
            } else if (k == "special_tokens") {
                if (i >= s.size() || s[i] != '[') {
                    err = "special_tokens must be an array";
                    return false;
                }
                ++i;
                for (;;) {
                    skip_ws_all();
                    if (i < s.size() && s[i] == ']') {
                        ++i;
                        break;
                    }
                    if (i >= s.size()) {
                        err = "unterminated special_tokens array";
                        return false;
                    }
                    if (!parse_inline_special(cfg)) return false;
                    skip_ws_all();
                    if (i < s.size() && s[i] == ',') ++i;
                }
            } else {
                if (!skip_value()) {
                    err = "bad TOML value";
                    return false;
                }
            }
        }
        return flush_table();
    }
};

// TokenizerInputFileType::from_path (config.rs:74-95): 0 toml, 1 bed, 2 bed.gz, -1 invalid
int input_file_type(const std::string &path) {
    const std::string ext = extension_of(path);
    if (ext == "gz") {
        const std::string stem = file_stem(path);
        return extension_of(stem) == "bed" ? 2 : -1;
    }
    if (ext == "toml") return 0;
    if (ext == "bed") return 1;
    return -1;
}

// Universe::try_from(&Path) -- universe/mod.rs:123-197, universe/utils.rs:7-19
gtars_status load_universe(const std::string &path, gtars_tokenizer *t) {
    std::string data, err;
    if (!read_all(path, data, err)) return fail(GTARS_ERR_IO, err);
    LineIter it(data);
    std::string line;
    std::vector<std::string> lines;
    while (it.next(line)) lines.push_back(line);
    if (lines.empty()) return fail(GTARS_ERR_PARSE, "Could not determine the universe type from the file");
    const std::string &first = lines[0];
    if (first.compare(0, 5, "track") == 0)
        return fail(GTARS_ERR_PARSE, "Could not determine the universe type from the file");
    const size_t nparts = split_char(first, '\t').size();
    if (nparts == 3) {
        for (const std::string &l : lines) {
            const std::vector<std::string> parts = split_ws(l);
            if (parts.size() != 3) return fail(GTARS_ERR_PARSE, "Error parsing line: " + l);
            t->regions.push_back(parts[0] + ":" + parts[1] + "-" + parts[2]);
        }
    } else if (nparts >= 5) {
        t->has_scores = true;
        for (const std::string &l : lines) {
            const std::vector<std::string> parts = split_char(l, '\t');
            if (parts.size() < 5) return fail(GTARS_ERR_PARSE, "Error parsing line: " + l);
            const std::string region = parts[0] + ":" + parts[1] + "-" + parts[2];
            double score;
            if (!parse_f64(parts[4], score)) return fail(GTARS_ERR_PARSE, "bad score in universe line: " + l);
            t->regions.push_back(region);
            t->names[region] = parts[3];
            t->scores[region] = score;
        }
    } else {
        return fail(GTARS_ERR_PARSE, "Could not determine the universe type from the file");
    }
    // generate_region_string_to_id_map (gtars-core/src/utils.rs:240-252): dense ids over DISTINCT strings
    for (const std::string &r : t->regions) {
        if (t->region_to_id.find(r) == t->region_to_id.end()) {
            const uint32_t id = (uint32_t)t->region_to_id.size();
            t->region_to_id.emplace(r, id);
            t->vocab_order.push_back(r);
        }
    }
    // generate_id_to_region_string_map (utils.rs:259-271): id i <- regions[i], one per LINE
    for (size_t i = 0; i < t->regions.size(); ++i) t->id_to_region.emplace((uint32_t)i, t->regions[i]);
    return GTARS_OK;
}

// Universe::add_token_to_universe (universe/mod.rs:51-56)
void add_token(gtars_tokenizer *t, const std::string &tok) {
    const uint32_t new_id = (uint32_t)t->region_to_id.size();
    if (t->region_to_id.find(tok) == t->region_to_id.end()) t->vocab_order.push_back(tok);
    t->region_to_id[tok] = new_id;
    t->id_to_region[new_id] = tok;
    t->regions.push_back(tok);
}

// prepare_universe_and_special_tokens + create_tokenize_core_from_universe (utils/mod.rs:34-99)
gtars_status finish_tokenizer(const std::string &universe_path, gtars_tokenizer *t) {
    gtars_status st = load_universe(universe_path, t);
    if (st) return st;
    const size_t n_real = t->regions.size();
    for (int k = 0; k < 7; ++k) add_token(t, t->special[k]);  // add_special_tokens (universe/mod.rs:114-120)
    std::vector<uint32_t> ch, s, e, v;
    ch.reserve(n_real);
    for (const std::string &region : t->regions) {
        bool is_special = false;
        for (int k = 0; k < 7; ++k) is_special = is_special || region == t->special[k];
        if (is_special) continue;
        // region.split(":") -> parts[0], parts[1]; parts[1].split("-") -> [0], [1]
        const std::vector<std::string> parts = split_char(region, ':');
        if (parts.size() < 2) return fail(GTARS_ERR_PARSE, "universe region is not chr:start-end: " + region);
        const std::vector<std::string> se = split_char(parts[1], '-');
        uint32_t start, end;
        if (se.size() < 2 || !parse_u32(se[0], start) || !parse_u32(se[1], end))
            return fail(GTARS_ERR_PARSE, "universe region has non-numeric coordinates: " + region);
        ch.push_back(t->chroms.get_or_add(parts[0]));
        s.push_back(start);
        e.push_back(end);
        v.push_back(t->region_to_id[region]);
    }
    t->unk_id = t->region_to_id[t->special[0]];
    return gtars_index_build(ch.data(), s.data(), e.data(), v.data(), ch.size(), (uint32_t)t->chroms.names.size(),
                             t->kind, &t->index);
}

gtars_status tokenizer_from_config(const std::string &path, gtars_tokenizer_t **out) {
    std::string data, err;
    {
        FILE *f = fopen(path.c_str(), "rb");
        if (!f) return fail(GTARS_ERR_IO, "Failed to open file: \"" + path + "\": " + strerror(errno));
        char buf[65536];
        size_t n;
        while ((n = fread(buf, 1, sizeof buf, f)) > 0) data.append(buf, n);
        fclose(f);
    }
    TomlCfg cfg;
    TomlParser p(data);
    if (!p.parse(cfg)) return fail(GTARS_ERR_CONFIG, "invalid tokenizer config " + path + ": " + p.err);
    if (!cfg.has_universe) return fail(GTARS_ERR_CONFIG, "invalid tokenizer config " + path + ": missing field `universe`");
    auto *t = new gtars_tokenizer();
    for (int k = 0; k < 7; ++k) t->special[k] = kSpecialDefaults[k];
    for (const auto &sp : cfg.specials) {
        int slot = -1;
        for (int k = 0; k < 7; ++k)
            if (sp.first == kSpecialSlots[k]) slot = k;
        if (slot < 0) {
            delete t;
            return fail(GTARS_ERR_CONFIG, "invalid tokenizer config: unknown special token name `" + sp.first + "`");
        }
        t->special[slot] = sp.second;
    }
    if (cfg.has_type) {
        if (cfg.type == "bits")
            t->kind = GTARS_KIND_BITS;
        else if (cfg.type == "ailist")
            t->kind = GTARS_KIND_AILIST;
        else {
            delete t;
            return fail(GTARS_ERR_CONFIG, "Invalid tokenizer type in config file: `" + cfg.type + "`");
        }
    }
    // the universe path is relative to the config file's directory (tokenizer.rs:64-65)
    const std::string dir = parent_dir(path);
    std::string upath = cfg.universe;
    if (!(upath.size() && upath[0] == '/')) upath = dir.empty() ? upath : (dir == "/" ? "/" + upath : dir + "/" + upath);
    gtars_status st = finish_tokenizer(upath, t);
    if (st) {
        gtars_tokenizer_free(t);
        return st;
    }
    *out = t;
    return GTARS_OK;
}

gtars_status tokenizer_from_bed(const std::string &path, gtars_tokenizer_t **out) {
    auto *t = new gtars_tokenizer();
    for (int k = 0; k < 7; ++k) t->special[k] = kSpecialDefaults[k];
    t->kind = GTARS_KIND_BITS;  // tokenizer.rs:93
    gtars_status st = finish_tokenizer(path, t);
    if (st) {
        gtars_tokenizer_free(t);
        return st;
    }
    *out = t;
    return GTARS_OK;
}

// core of Tokenizer::encode on chromosome ids of this tokenizer's dictionary
gtars_status encode_core(const gtars_tokenizer *t, const uint32_t *qc, const uint32_t *qs, const uint32_t *qe,
                         uint64_t n, uint32_t **out_ids, uint64_t *out_n) {
    *out_ids = nullptr;
    *out_n = 0;
    uint32_t *ids = nullptr;
    uint64_t h = 0;
    gtars_status st = gtars_tokenize(t->index, qc, qs, qe, n, nullptr, &ids, &h);
    if (st) return st;
    if (h == 0) {
        // tokenizer.rs:158-160: nothing overlapped in the whole batch -> [unk]
        gtars_free(ids);
        ids = (uint32_t *)malloc(sizeof(uint32_t));
        if (!ids) return fail(GTARS_ERR_INTERNAL, "out of host memory");
        ids[0] = t->unk_id;
        h = 1;
    }
    *out_ids = ids;
    *out_n = h;
    return GTARS_OK;
}

}  // namespace

extern "C" {

gtars_status gtars_tokenizer_from_auto(const char *path, gtars_tokenizer_t **out) {
    if (!path || !out) return fail(GTARS_ERR_INVALID_ARG, "NULL argument");
    *out = nullptr;
    const std::string p(path);
    switch (input_file_type(p)) {
        case 0: return tokenizer_from_config(p, out);
        case 1:
        case 2: return tokenizer_from_bed(p, out);
        default:
            return fail(GTARS_ERR_CONFIG,
                        "Missing or invalid file extension in tokenizer config file. It must be `toml`, `bed` or `bed.gz`");
    }
}

gtars_status gtars_tokenizer_from_config(const char *path, gtars_tokenizer_t **out) {
    if (!path || !out) return fail(GTARS_ERR_INVALID_ARG, "NULL argument");
    *out = nullptr;
    return tokenizer_from_config(path, out);
}

gtars_status gtars_tokenizer_from_bed(const char *path, gtars_tokenizer_t **out) {
    if (!path || !out) return fail(GTARS_ERR_INVALID_ARG, "NULL argument");
    *out = nullptr;
    return tokenizer_from_bed(path, out);
}

void gtars_tokenizer_free(gtars_tokenizer_t *t) {
    if (!t) return;
    gtars_index_free(t->index);
    gtars::frag_chroms_free(t->frag_chroms);
    delete t;
}

uint64_t gtars_tokenizer_vocab_size(const gtars_tokenizer_t *t) { return t ? t->region_to_id.size() : 0; }
int gtars_tokenizer_kind(const gtars_tokenizer_t *t) { return t ? t->kind : -1; }

const char *gtars_tokenizer_id_to_token(const gtars_tokenizer_t *t, uint32_t id) {
    if (!t) return nullptr;
    auto it = t->id_to_region.find(id);
    return it == t->id_to_region.end() ? nullptr : it->second.c_str();
}

int64_t gtars_tokenizer_token_to_id(const gtars_tokenizer_t *t, const char *token) {
    if (!t || !token) return -1;
    auto it = t->region_to_id.find(token);
    return it == t->region_to_id.end() ? -1 : (int64_t)it->second;
}

const char *gtars_tokenizer_vocab_token(const gtars_tokenizer_t *t, uint64_t i, uint32_t *id) {
    if (!t || i >= t->vocab_order.size()) return nullptr;
    const std::string &k = t->vocab_order[i];
    if (id) *id = t->region_to_id.at(k);
    return k.c_str();
}

const char *gtars_tokenizer_special_token(const gtars_tokenizer_t *t, int which) {
    return t && which >= 0 && which < 7 ? t->special[which].c_str() : nullptr;
}

const char *gtars_tokenizer_region_name(const gtars_tokenizer_t *t, const char *region) {
    if (!t || !region) return nullptr;
    auto it = t->names.find(region);
    return it == t->names.end() ? nullptr : it->second.c_str();
}

double gtars_tokenizer_region_score(const gtars_tokenizer_t *t, const char *region) {
    if (!t || !region) return std::numeric_limits<double>::quiet_NaN();
    auto it = t->scores.find(region);
    return it == t->scores.end() ? std::numeric_limits<double>::quiet_NaN() : it->second;
}

int64_t gtars_tokenizer_chrom_id(const gtars_tokenizer_t *t, const char *chr) {
    return t && chr ? t->chroms.find(chr) : -1;
}
uint32_t gtars_tokenizer_n_chrom(const gtars_tokenizer_t *t) { return t ? (uint32_t)t->chroms.names.size() : 0; }
const char *gtars_tokenizer_chrom_name(const gtars_tokenizer_t *t, uint32_t id) {
    return t && id < t->chroms.names.size() ? t->chroms.names[id].c_str() : nullptr;
}
const gtars_index_t *gtars_tokenizer_index(const gtars_tokenizer_t *t) { return t ? t->index : nullptr; }

gtars_status gtars_tokenizer_encode_regionset(const gtars_tokenizer_t *t, const gtars_regionset_t *rs,
                                              uint32_t **out_ids, uint64_t *out_n) {
    if (!t || !rs || !out_ids || !out_n) return fail(GTARS_ERR_INVALID_ARG, "NULL argument");
    const std::vector<uint32_t> qc = translate_chroms(rs, t->chroms);
    return encode_core(t, qc.data(), rs->starts.data(), rs->ends.data(), rs->size(), out_ids, out_n);
}

gtars_status gtars_tokenizer_encode_arrays(const gtars_tokenizer_t *t, const char *const *chrs,
                                           const uint32_t *starts, const uint32_t *ends, uint64_t n,
                                           uint32_t **out_ids, uint64_t *out_n) {
    if (!t || !out_ids || !out_n || (n && (!chrs || !starts || !ends))) return fail(GTARS_ERR_INVALID_ARG, "NULL argument");
    std::vector<uint32_t> qc(n);
    // consecutive regions usually share a chromosome: one-entry cache in front of the hash lookup
    const char *last = nullptr;
    uint32_t last_id = GTARS_UNKNOWN_CHROM;
    for (uint64_t i = 0; i < n; ++i) {
        if (!chrs[i]) return fail(GTARS_ERR_INVALID_ARG, "NULL chromosome name");
        if (!last || strcmp(last, chrs[i]) != 0) {
            const int64_t id = t->chroms.find(chrs[i]);
            last_id = id < 0 ? GTARS_UNKNOWN_CHROM : (uint32_t)id;
            last = chrs[i];
        }
        qc[i] = last_id;
    }
    return encode_core(t, qc.data(), starts, ends, n, out_ids, out_n);
}

gtars_status gtars_tokenizer_encode_ids(const gtars_tokenizer_t *t, const uint32_t *chrom_ids,
                                        const uint32_t *starts, const uint32_t *ends, uint64_t n,
                                        uint64_t *offsets, uint32_t **out_ids, uint64_t *out_n) {
    if (!t || !out_ids || !out_n) return fail(GTARS_ERR_INVALID_ARG, "NULL argument");
    return gtars_tokenize(t->index, chrom_ids, starts, ends, n, offsets, out_ids, out_n);
}

// tokenize_fragment_file -- gtars-tokenizers/src/utils/fragments.rs:12-82
// ------------------------------------------------------------ fragment files: SoA ingest
// (gtars-tokenizers/src/utils/fragments.rs:12-56 parse_fragment_line)
// The decompressed text is cut at line ends into one chunk per host thread; every chunk is scanned in
// place (no per-line or per-field allocation) into SoA columns with chunk-local dictionaries, which are
// then merged in chunk order -- so dictionary ids are in first-seen order, exactly as a serial pass.
namespace {

struct FragChunk {
    std::vector<uint32_t> c, s, e, b;
    ViewDict chroms, barcodes;
    size_t n_lines = 0;           // every line of the chunk, comments included (for error line numbers)
    int err = 0;                  // 0 ok, 1 < 5 fields, 2 bad start, 3 bad end, 4 bad read support (strict mode)
    size_t err_line = 0;          // line index inside the chunk
};

void parse_fragment_chunk(const char *p, const char *end, FragChunk &out, bool check_support = false) {
    const size_t guess = (size_t)(end - p) / 30 + 16;
    out.c.reserve(guess); out.s.reserve(guess); out.e.reserve(guess); out.b.reserve(guess);
    uint32_t last_c = 0, last_b = 0;
    const char *last_c_p = nullptr, *last_b_p = nullptr;
    size_t last_c_n = 0, last_b_n = 0;
    while (p < end) {
        const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
        const char *le = nl ? nl : end;
        const char *next = nl ? nl + 1 : end;
        if (nl && le > p && le[-1] == '\r') --le;  // BufRead::lines strips "\r\n"
        const size_t ln = out.n_lines++;
        if (le > p && *p == '#') { p = next; continue; }
        const char *f[5]; size_t fl[5]; int nf = 0;
        const char *q = p;
        while (q < le && nf < 5) {
            while (q < le && is_ws(*q)) ++q;
            const char *st = q;
            while (q < le && !is_ws(*q)) ++q;
            if (q > st) { f[nf] = st; fl[nf] = (size_t)(q - st); ++nf; }
        }
        uint32_t sv, ev;
        if (nf < 5) { out.err = 1; out.err_line = ln; return; }
        if (!parse_u32_view(f[1], fl[1], sv)) { out.err = 2; out.err_line = ln; return; }
        if (!parse_u32_view(f[2], fl[2], ev)) { out.err = 3; out.err_line = ln; return; }
        if (check_support) {  // Fragment::from_str also parses the read support (gtars-core/src/models/fragments.rs:31-33)
            uint32_t sup;
            if (!parse_u32_view(f[4], fl[4], sup)) { out.err = 4; out.err_line = ln; return; }
        }
        // fragment files are sorted by chromosome and barcodes repeat: try the previous line's ids first
        if (!(last_c_p && last_c_n == fl[0] && memcmp(last_c_p, f[0], fl[0]) == 0)) {
            last_c = out.chroms.get_or_add(f[0], fl[0]); last_c_p = f[0]; last_c_n = fl[0];
        }
        if (!(last_b_p && last_b_n == fl[3] && memcmp(last_b_p, f[3], fl[3]) == 0)) {
            last_b = out.barcodes.get_or_add(f[3], fl[3]); last_b_p = f[3]; last_b_n = fl[3];
        }
        out.c.push_back(last_c); out.s.push_back(sv); out.e.push_back(ev); out.b.push_back(last_b);
        p = next;
    }
}

struct FragTable {
    std::vector<uint32_t> c, s, e, b;
    std::vector<std::string> chroms, barcodes;
};

gtars_status read_fragment_table(const char *path, FragTable &ft, bool check_support = false) {
    std::string data, err;
    if (!read_all(path, data, err)) return fail(GTARS_ERR_IO, err);
    unsigned nt = host_thread_budget(32);
    nt = (unsigned)std::min<size_t>(nt, data.size() / (1u << 20) + 1);
    std::vector<size_t> cut(nt + 1, data.size());
    cut[0] = 0;
    for (unsigned i = 1; i < nt; ++i) {
        size_t pos = data.size() / nt * i;
        const size_t nl = data.find('\n', pos);
        cut[i] = nl == std::string::npos ? data.size() : nl + 1;
    }
    for (unsigned i = 1; i <= nt; ++i) cut[i] = std::max(cut[i], cut[i - 1]);
    std::vector<FragChunk> chunks(nt);
    {
        std::vector<std::thread> th;
        for (unsigned i = 1; i < nt; ++i)
            th.emplace_back([&, i] { parse_fragment_chunk(data.data() + cut[i], data.data() + cut[i + 1], chunks[i], check_support); });
        parse_fragment_chunk(data.data() + cut[0], data.data() + cut[1], chunks[0], check_support);
        for (auto &t : th) t.join();
    }
    size_t line0 = 0, total = 0;
    for (unsigned i = 0; i < nt; ++i) {
        if (chunks[i].err) {
            const std::string ln = std::to_string(line0 + chunks[i].err_line);
            if (chunks[i].err == 1) return fail(GTARS_ERR_PARSE, "Invalid fragment file detected at line: " + ln);
            if (chunks[i].err == 4) return fail(GTARS_ERR_PARSE, "invalid digit found in string (read support, line " + ln + ")");
            return fail(GTARS_ERR_PARSE, std::string("Failed to parse ") + (chunks[i].err == 2 ? "start" : "end") +
                                             " position at line " + ln);
        }
        line0 += chunks[i].n_lines;
        total += chunks[i].c.size();
    }
    // merge the dictionaries in chunk order (first-seen order of a serial pass), then the columns
    ViewDict gc, gb;
    std::vector<std::vector<uint32_t>> mapc(nt), mapb(nt);
    for (unsigned i = 0; i < nt; ++i) {
        for (const std::string &n : chunks[i].chroms.names) mapc[i].push_back(gc.get_or_add(n.data(), n.size()));
        for (const std::string &n : chunks[i].barcodes.names) mapb[i].push_back(gb.get_or_add(n.data(), n.size()));
    }
    ft.c.resize(total); ft.s.resize(total); ft.e.resize(total); ft.b.resize(total);
    std::vector<size_t> base(nt + 1, 0);
    for (unsigned i = 0; i < nt; ++i) base[i + 1] = base[i] + chunks[i].c.size();
    {
        auto merge = [&](unsigned i) {
            const FragChunk &ck = chunks[i];
            const size_t o = base[i], m = ck.c.size();
            for (size_t k = 0; k < m; ++k) {
                ft.c[o + k] = mapc[i][ck.c[k]];
                ft.b[o + k] = mapb[i][ck.b[k]];
            }
            if (m) {
                memcpy(&ft.s[o], ck.s.data(), m * sizeof(uint32_t));
                memcpy(&ft.e[o], ck.e.data(), m * sizeof(uint32_t));
            }
        };
        std::vector<std::thread> th;
        for (unsigned i = 1; i < nt; ++i) th.emplace_back(merge, i);
        merge(0);
        for (auto &t : th) t.join();
    }
    ft.chroms.assign(gc.names.begin(), gc.names.end());
    ft.barcodes.assign(gb.names.begin(), gb.names.end());
    return GTARS_OK;
}

}  // namespace

struct gtars_fragments {
    FragTable t;
    std::vector<const char *> chrom_ptrs, barcode_ptrs;
};

// ------------------------------------------------------------ BED3 text mode of the overlaprs front end
// gtars-cli/src/overlaprs/handlers.rs:64-92, 123-139: EVERY line is a record (no header or comment skipping), fields are
// split on TAB only, the first three are chr / start / end with start and end through str::parse::<u32>; what follows is
// ignored.  Same chunked in-place scan as the fragment reader; the columns come back in a gtars_fragments_t without barcodes.
namespace {
void parse_bed3_chunk(const char *p, const char *end, FragChunk &out) {
    const size_t guess = (size_t)(end - p) / 20 + 16;
    out.c.reserve(guess); out.s.reserve(guess); out.e.reserve(guess);
    uint32_t last_c = 0;
    const char *last_c_p = nullptr;
    size_t last_c_n = 0;
    while (p < end) {
        const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
        const char *le = nl ? nl : end;
        const char *next = nl ? nl + 1 : end;
        if (nl && le > p && le[-1] == '\r') --le;  // BufRead::lines strips "\r\n"
        const size_t ln = out.n_lines++;
        const char *t1 = (const char *)memchr(p, '\t', (size_t)(le - p));
        if (!t1) { out.err = 1; out.err_line = ln; return; }            // Missing start field
        const char *t2 = (const char *)memchr(t1 + 1, '\t', (size_t)(le - t1 - 1));
        if (!t2) { out.err = 2; out.err_line = ln; return; }            // Missing end field
        const char *t3 = (const char *)memchr(t2 + 1, '\t', (size_t)(le - t2 - 1));
        const char *e_end = t3 ? t3 : le;
        uint32_t sv, ev;
        if (!parse_u32_view(t1 + 1, (size_t)(t2 - t1 - 1), sv)) { out.err = 3; out.err_line = ln; return; }
        if (!parse_u32_view(t2 + 1, (size_t)(e_end - t2 - 1), ev)) { out.err = 4; out.err_line = ln; return; }
        const size_t cl = (size_t)(t1 - p);
        if (!(last_c_p && last_c_n == cl && memcmp(last_c_p, p, cl) == 0)) {
            last_c = out.chroms.get_or_add(p, cl); last_c_p = p; last_c_n = cl;
        }
        out.c.push_back(last_c); out.s.push_back(sv); out.e.push_back(ev);
        p = next;
    }
}

gtars_status read_bed3_lines(const char *path, FragTable &ft) {
    std::string data, err;
    if (!read_all(path, data, err)) return fail(GTARS_ERR_IO, err);
    unsigned nt = host_thread_budget(32);
    nt = (unsigned)std::min<size_t>(nt, data.size() / (1u << 20) + 1);
    std::vector<size_t> cut(nt + 1, data.size());
    cut[0] = 0;
    for (unsigned i = 1; i < nt; ++i) {
        const size_t nl = data.find('\n', data.size() / nt * i);
        cut[i] = nl == std::string::npos ? data.size() : nl + 1;
    }
    for (unsigned i = 1; i <= nt; ++i) cut[i] = std::max(cut[i], cut[i - 1]);
    std::vector<FragChunk> chunks(nt);
    {
        std::vector<std::thread> th;
        for (unsigned i = 1; i < nt; ++i)
            th.emplace_back([&, i] { parse_bed3_chunk(data.data() + cut[i], data.data() + cut[i + 1], chunks[i]); });
        parse_bed3_chunk(data.data() + cut[0], data.data() + cut[1], chunks[0]);
        for (auto &t : th) t.join();
    }
    size_t line0 = 0, total = 0;
    for (unsigned i = 0; i < nt; ++i) {
        if (chunks[i].err) {
            const std::string where = std::string(path) + ":" + std::to_string(line0 + chunks[i].err_line + 1) + ": ";
            if (chunks[i].err <= 2) return fail(GTARS_ERR_PARSE, where + (chunks[i].err == 1 ? "Missing start field" : "Missing end field"));
            return fail(GTARS_ERR_PARSE, where + "invalid digit found in string (" + (chunks[i].err == 3 ? "start" : "end") + ")");
        }
        line0 += chunks[i].n_lines;
        total += chunks[i].c.size();
    }
    ViewDict gc;
    std::vector<std::vector<uint32_t>> mapc(nt);
    for (unsigned i = 0; i < nt; ++i)
        for (const std::string &n : chunks[i].chroms.names) mapc[i].push_back(gc.get_or_add(n.data(), n.size()));
    ft.c.resize(total); ft.s.resize(total); ft.e.resize(total); ft.b.assign(total, 0);
    size_t o = 0;
    for (unsigned i = 0; i < nt; ++i) {
        const FragChunk &ck = chunks[i];
        const size_t m = ck.c.size();
        for (size_t k = 0; k < m; ++k) ft.c[o + k] = mapc[i][ck.c[k]];
        if (m) {
            memcpy(&ft.s[o], ck.s.data(), m * sizeof(uint32_t));
            memcpy(&ft.e[o], ck.e.data(), m * sizeof(uint32_t));
        }
        o += m;
    }
    ft.chroms.assign(gc.names.begin(), gc.names.end());
    ft.barcodes.clear();
    return GTARS_OK;
}
}  // namespace

static gtars_status gtars_fragments_read_impl(const char *path, gtars_fragments_t **out) {
    if (!path || !out) return fail(GTARS_ERR_INVALID_ARG, "NULL argument");
    *out = nullptr;
    std::unique_ptr<gtars_fragments> f(new gtars_fragments());
    gtars_status st = read_fragment_table(path, f->t);
    if (st) return st;
    *out = f.release();
    return GTARS_OK;
}
gtars_status gtars_fragments_read_strict(const char *path, gtars_fragments_t **out) {
    if (!path || !out) return fail(GTARS_ERR_INVALID_ARG, "NULL argument");
    *out = nullptr;
    std::unique_ptr<gtars_fragments> f(new gtars_fragments());
    gtars_status st = read_fragment_table(path, f->t, true);
    if (st) return st;
    *out = f.release();
    return GTARS_OK;
}
void gtars_fragments_free(gtars_fragments_t *f) { delete f; }
uint64_t gtars_fragments_len(const gtars_fragments_t *f) { return f ? f->t.c.size() : 0; }
uint32_t gtars_fragments_n_chrom(const gtars_fragments_t *f) { return f ? (uint32_t)f->t.chroms.size() : 0; }
uint32_t gtars_fragments_n_barcodes(const gtars_fragments_t *f) { return f ? (uint32_t)f->t.barcodes.size() : 0; }
const char *gtars_fragments_chrom_name(const gtars_fragments_t *f, uint32_t id) {
    return f && id < f->t.chroms.size() ? f->t.chroms[id].c_str() : nullptr;
}
const char *gtars_fragments_barcode_name(const gtars_fragments_t *f, uint32_t id) {
    return f && id < f->t.barcodes.size() ? f->t.barcodes[id].c_str() : nullptr;
}
const uint32_t *gtars_fragments_chrom_ids(const gtars_fragments_t *f) { return f ? f->t.c.data() : nullptr; }
const uint32_t *gtars_fragments_starts(const gtars_fragments_t *f) { return f ? f->t.s.data() : nullptr; }
const uint32_t *gtars_fragments_ends(const gtars_fragments_t *f) { return f ? f->t.e.data() : nullptr; }
const uint32_t *gtars_fragments_barcode_ids(const gtars_fragments_t *f) { return f ? f->t.b.data() : nullptr; }

static gtars_status gtars_tokenizer_tokenize_fragment_file_impl(const gtars_tokenizer_t *t, const char *path,
                                                    gtars_fragment_tokens_t **out) {
    if (!t || !path || !out) return fail(GTARS_ERR_INVALID_ARG, "NULL argument");
    *out = nullptr;
    FragTable frag;
    {
        gtars_status st0 = read_fragment_table(path, frag);
        if (st0) return st0;
    }
    // the file's chromosome dictionary -> the tokenizer's
    std::vector<uint32_t> cmap(frag.chroms.size());
    for (size_t i = 0; i < cmap.size(); ++i) {
        const int64_t cid = t->chroms.find(frag.chroms[i]);
        cmap[i] = cid < 0 ? GTARS_UNKNOWN_CHROM : (uint32_t)cid;
    }
    std::vector<uint32_t> &qc = frag.c, &qs = frag.s, &qe = frag.e, &bc = frag.b;
    for (uint32_t &v : qc) v = cmap[v];
    const uint64_t n = qc.size();
    std::vector<uint64_t> off(n + 1, 0);
    uint32_t *ids = nullptr;
    uint64_t h = 0;
    gtars_status st = gtars_tokenize(t->index, qc.data(), qs.data(), qe.data(), n, off.data(), &ids, &h);
    if (st) return st;
    // one tokenize() per fragment: a fragment without hits contributes exactly one unk id
    const uint64_t nb = frag.barcodes.size();
    std::vector<uint64_t> cnt(nb + 1, 0);
    for (uint64_t i = 0; i < n; ++i) {
        const uint64_t k = off[i + 1] - off[i];
        cnt[bc[i] + 1] += k ? k : 1;
    }
    for (uint64_t b = 0; b < nb; ++b) cnt[b + 1] += cnt[b];
    auto *ft = (gtars_fragment_tokens_t *)calloc(1, sizeof(gtars_fragment_tokens_t));
    ft->n_barcodes = nb;
    ft->barcodes = (char **)calloc(nb ? nb : 1, sizeof(char *));
    ft->offsets = (uint64_t *)malloc((nb + 1) * sizeof(uint64_t));
    ft->ids = (uint32_t *)malloc((cnt[nb] ? cnt[nb] : 1) * sizeof(uint32_t));
    memcpy(ft->offsets, cnt.data(), (nb + 1) * sizeof(uint64_t));
    for (uint64_t b = 0; b < nb; ++b) ft->barcodes[b] = dup_cstr(frag.barcodes[b]);
    std::vector<uint64_t> fill(cnt.begin(), cnt.end() - 1);
    for (uint64_t i = 0; i < n; ++i) {
        uint64_t &w = fill[bc[i]];
        if (off[i + 1] == off[i]) {
            ft->ids[w++] = t->unk_id;
        } else {
            for (uint64_t k = off[i]; k < off[i + 1]; ++k) ft->ids[w++] = ids[k];
        }
    }
    gtars_free(ids);
    *out = ft;
    return GTARS_OK;
}

void gtars_fragment_tokens_free(gtars_fragment_tokens_t *ft) {
    if (!ft) return;
    for (uint64_t b = 0; b < ft->n_barcodes; ++b) free(ft->barcodes[b]);
    free(ft->barcodes);
    free(ft->offsets);
    free(ft->ids);
    free(ft);
}

gtars_status gtars_fragment_tokens_barcodes_joined(const gtars_fragment_tokens_t *ft, char **out, uint64_t *out_len) {
    if (!ft || !out || !out_len) return fail(GTARS_ERR_INVALID_ARG, "NULL argument");
    size_t total = 0;
    std::vector<size_t> len(ft->n_barcodes);
    for (uint64_t b = 0; b < ft->n_barcodes; ++b) total += (len[b] = strlen(ft->barcodes[b])) + 1;
    char *buf = (char *)malloc(total ? total : 1);
    if (!buf) return fail(GTARS_ERR_INTERNAL, "out of host memory");
    size_t at = 0;
    for (uint64_t b = 0; b < ft->n_barcodes; ++b) {
        memcpy(buf + at, ft->barcodes[b], len[b]);
        at += len[b];
        buf[at++] = '\n';
    }
    *out = buf;
    *out_len = total ? total - 1 : 0;
    return GTARS_OK;
}

// ===================================================================== gtok

// write_tokens_to_gtok -- gtars-io/src/gtok.rs:125-165
gtars_status gtars_gtok_write(const char *path, const uint32_t *tokens, uint64_t n) {
    if (!path || (n && !tokens)) return fail(GTARS_ERR_INVALID_ARG, "NULL argument");
    // create_dir_all(parent)
    const std::string dir = parent_dir(path);
    if (!dir.empty() && dir != "/") {
        std::string acc;
        for (const std::string &comp : split_char(dir, '/')) {
            if (comp.empty()) {
                acc += "/";
                continue;
            }
            acc += (acc.empty() || acc.back() == '/') ? comp : "/" + comp;
            if (mkdir(acc.c_str(), 0777) != 0 && errno != EEXIST)
                return fail(GTARS_ERR_IO, "Failed to create parent directories for gtok file: " + acc);
        }
    }
    FILE *f = fopen(path, "wb");
    if (!f) return fail(GTARS_ERR_IO, std::string("Failed to create gtok file: ") + path);
    bool small = true;
    for (uint64_t i = 0; i < n; ++i) small = small && tokens[i] <= 0xFFFFu;
    fwrite("GTOK", 1, 4, f);
    const unsigned char flag = small ? 0x01 : 0x02;
    fwrite(&flag, 1, 1, f);
    if (small) {
        std::vector<uint16_t> v(n);
        for (uint64_t i = 0; i < n; ++i) v[i] = (uint16_t)tokens[i];
        if (n) fwrite(v.data(), 2, n, f);  // little-endian host
    } else if (n) {
        fwrite(tokens, 4, n, f);
    }
    fclose(f);
    return GTARS_OK;
}

// read_tokens_from_gtok -- gtars-io/src/gtok.rs:174-210
gtars_status gtars_gtok_read(const char *path, uint32_t **out_tokens, uint64_t *out_n) {
    if (!path || !out_tokens || !out_n) return fail(GTARS_ERR_INVALID_ARG, "NULL argument");
    *out_tokens = nullptr;
    *out_n = 0;
    FILE *f = fopen(path, "rb");
    if (!f) return fail(GTARS_ERR_IO, std::string("Failed to open file: ") + path);
    std::string data;
    char buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) data.append(buf, n);
    fclose(f);
    if (data.size() < 5 || data.compare(0, 4, "GTOK") != 0)
        return fail(GTARS_ERR_PARSE, "File doesn't appear to be a valid .gtok file.");
    const unsigned char flag = (unsigned char)data[4];
    const size_t body = data.size() - 5;
    uint64_t cnt;
    uint32_t *res;
    if (flag == 0x01) {
        cnt = body / 2;
        res = (uint32_t *)malloc((cnt ? cnt : 1) * 4);
        for (uint64_t i = 0; i < cnt; ++i) {
            uint16_t v;
            memcpy(&v, data.data() + 5 + 2 * i, 2);
            res[i] = v;
        }
    } else if (flag == 0x02) {
        cnt = body / 4;
        res = (uint32_t *)malloc((cnt ? cnt : 1) * 4);
        if (cnt) memcpy(res, data.data() + 5, cnt * 4);
    } else {
        return fail(GTARS_ERR_PARSE, "Invalid data format flag found in gtok file");
    }
    *out_tokens = res;
    *out_n = cnt;
    return GTARS_OK;
}

}  // extern "C"

// =================================================================== IGD DB

struct gtars_igddb {
    Dict chroms;
    struct FileInfo {
        std::string filename;
        uint32_t num_regions;
        double avg_width;
    };
    std::vector<FileInfo> files;
    gtars_igd_t *igd = nullptr;
};

namespace {

// str::parse::<i32>(): optional sign, ASCII digits, must fit
inline bool parse_i32_view(const char *p, size_t n, int32_t &out) {
    size_t i = 0;
    bool neg = false;
    if (n && (p[0] == '+' || p[0] == '-')) {
        neg = p[0] == '-';
        i = 1;
    }
    if (i >= n) return false;
    int64_t v = 0;
    for (; i < n; ++i) {
        const unsigned d = (unsigned char)p[i] - '0';
        if (d > 9) return false;
        v = v * 10 + d;
        if (v > 2147483648ll) return false;
    }
    if (neg) v = -v;
    if (v < -2147483648ll || v > 2147483647ll) return false;
    out = (int32_t)v;
    return true;
}

// One BED file of an IGD database, parsed in place.  Igd::parse_bed_line (gtars-igd/src/igd.rs:850-867):
// tab-split; chrom, start, end (i32) required; chrom shorter than 40 bytes and end > 0; column 5 = score
// if it parses, else -1.  Unparsable lines are skipped.
struct IgdBedFile {
    bool readable = false, has_valid = false;
    uint32_t count = 0;          // lines with start >= 0 (FileInfo.num_regions, igd.rs:213-217)
    uint64_t total_width = 0;
    std::vector<uint32_t> c;     // file-local chromosome codes of the KEPT records (start < end)
    std::vector<int32_t> s, e, v;
    ViewDict chroms;             // in first-kept order
};

void parse_igd_bed_file(const std::string &path, IgdBedFile &out) {
    std::string data, err;
    if (!read_all(path, data, err)) return;  // Err(_) => continue (igd.rs:203-206)
    out.readable = true;
    const char *p = data.data(), *end = p + data.size();
    const size_t guess = data.size() / 24 + 16;
    out.c.reserve(guess); out.s.reserve(guess); out.e.reserve(guess); out.v.reserve(guess);
    const char *last_p = nullptr; size_t last_n = 0; uint32_t last_id = 0;
    while (p < end) {
        const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
        const char *le = nl ? nl : end;
        const char *next = nl ? nl + 1 : end;
        if (nl && le > p && le[-1] == '\r') --le;
        const char *line = p;
        p = next;
        const char *t1 = (const char *)memchr(line, '\t', (size_t)(le - line));
        if (!t1) continue;
        const char *t2 = (const char *)memchr(t1 + 1, '\t', (size_t)(le - t1 - 1));
        if (!t2) continue;
        const char *t3 = (const char *)memchr(t2 + 1, '\t', (size_t)(le - t2 - 1));
        int32_t sv, ev, score = -1;
        if (!parse_i32_view(t1 + 1, (size_t)(t2 - t1 - 1), sv)) continue;
        if (!parse_i32_view(t2 + 1, (size_t)((t3 ? t3 : le) - t2 - 1), ev)) continue;
        const size_t cn = (size_t)(t1 - line);
        if (cn >= 40 || ev <= 0) continue;
        if (t3) {
            const char *t4 = (const char *)memchr(t3 + 1, '\t', (size_t)(le - t3 - 1));
            if (t4) {
                const char *t5 = (const char *)memchr(t4 + 1, '\t', (size_t)(le - t4 - 1));
                int32_t v;
                if (parse_i32_view(t4 + 1, (size_t)((t5 ? t5 : le) - t4 - 1), v)) score = v;
            }
        }
        out.has_valid = true;
        if (sv >= 0) {
            // Igd::add creates the contig only for a record it keeps (igd.rs:114-133); count and
            // total_width are updated regardless (igd.rs:213-217)
            if (sv < ev) {
                if (!(last_p && last_n == cn && memcmp(last_p, line, cn) == 0)) {
                    last_id = out.chroms.get_or_add(line, cn); last_p = line; last_n = cn;
                }
                // the dictionary copies the name, so the view into `data` may die with this function
                out.c.push_back(last_id); out.s.push_back(sv); out.e.push_back(ev); out.v.push_back(score);
            }
            out.count += 1;
            out.total_width += (uint64_t)(int64_t)(ev - sv);
        }
    }
}

}  // namespace

extern "C" {

// Igd::from_bed_files -- gtars-igd/src/igd.rs:191-242.  Files are independent: they are parsed by a pool
// of host threads and merged in argument order (contigs and file indices come out as in a serial pass).
static gtars_status gtars_igddb_from_bed_files_impl(const char *const *paths, uint64_t n_paths, gtars_igddb_t **out) {
    if (!out || (n_paths && !paths)) return fail(GTARS_ERR_INVALID_ARG, "NULL argument");
    *out = nullptr;
    std::vector<IgdBedFile> parsed(n_paths);
    {
        unsigned nt = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(host_thread_budget(64), n_paths));
        std::atomic<uint64_t> next{0};
        auto work = [&] {
            for (uint64_t i = next.fetch_add(1); i < n_paths; i = next.fetch_add(1))
                parse_igd_bed_file(paths[i] ? paths[i] : "", parsed[i]);
        };
        std::vector<std::thread> th;
        for (unsigned i = 1; i < nt; ++i) th.emplace_back(work);
        work();
        for (auto &t : th) t.join();
    }
    auto *db = new gtars_igddb();
    uint64_t total = 0;
    for (const IgdBedFile &f : parsed) total += f.c.size();
    std::vector<uint32_t> ch, fi;
    std::vector<int32_t> st, en, va;
    ch.reserve(total); fi.reserve(total); st.reserve(total); en.reserve(total); va.reserve(total);
    for (uint64_t pi = 0; pi < n_paths; ++pi) {
        IgdBedFile &f = parsed[pi];
        // unreadable files and files without a parseable line are skipped (igd.rs:203-206, 223-225)
        if (!f.readable || !f.has_valid) continue;
        const uint32_t file_idx = (uint32_t)db->files.size();
        std::vector<uint32_t> cmap;
        for (const std::string &nm : f.chroms.names) cmap.push_back(db->chroms.get_or_add(nm));
        for (size_t k = 0; k < f.c.size(); ++k) ch.push_back(cmap[f.c[k]]);
        st.insert(st.end(), f.s.begin(), f.s.end());
        en.insert(en.end(), f.e.begin(), f.e.end());
        va.insert(va.end(), f.v.begin(), f.v.end());
        fi.insert(fi.end(), f.c.size(), file_idx);
        db->files.push_back({base_name(paths[pi] ? paths[pi] : ""), f.count, f.count ? (double)f.total_width / (double)f.count : 0.0});
        std::vector<uint32_t>().swap(f.c);  // release as we go
        std::vector<int32_t>().swap(f.s); std::vector<int32_t>().swap(f.e); std::vector<int32_t>().swap(f.v);
    }
    gtars_status s2 = gtars_igd_build(ch.data(), st.data(), en.data(), va.data(), fi.data(), ch.size(),
                                      (uint32_t)db->chroms.names.size(), (uint32_t)db->files.size(), &db->igd);
    if (s2) {
        gtars_igddb_free(db);
        return s2;
    }
    *out = db;
    return GTARS_OK;
}

// Igd::from_bed_dir -- gtars-igd/src/igd.rs:170-188
gtars_status gtars_igddb_from_bed_dir(const char *dir, gtars_igddb_t **out) {
    if (!dir || !out) return fail(GTARS_ERR_INVALID_ARG, "NULL argument");
    *out = nullptr;
    DIR *d = opendir(dir);
    if (!d) return fail(GTARS_ERR_IO, std::string("cannot read directory: ") + dir);
    std::vector<std::string> files;
    while (dirent *ent = readdir(d)) {
        const std::string name = ent->d_name;
        if (name == "." || name == "..") continue;
        const std::string full = std::string(dir) + (ends_with(dir, "/") ? "" : "/") + name;
        const std::string ext = extension_of(name);
        if ((ext == "bed" || ext == "gz") && is_regular_file(full)) files.push_back(full);
    }
    closedir(d);
    std::sort(files.begin(), files.end());
    std::vector<const char *> ptrs;
    for (const std::string &f : files) ptrs.push_back(f.c_str());
    return gtars_igddb_from_bed_files(ptrs.data(), ptrs.size(), out);
}

void gtars_igddb_free(gtars_igddb_t *db) {
    if (!db) return;
    gtars_igd_free(db->igd);
    delete db;
}

uint32_t gtars_igddb_n_files(const gtars_igddb_t *db) { return db ? (uint32_t)db->files.size() : 0; }
uint32_t gtars_igddb_n_contigs(const gtars_igddb_t *db) { return db ? (uint32_t)db->chroms.names.size() : 0; }
const char *gtars_igddb_file_name(const gtars_igddb_t *db, uint32_t i) {
    return db && i < db->files.size() ? db->files[i].filename.c_str() : nullptr;
}
uint32_t gtars_igddb_file_num_regions(const gtars_igddb_t *db, uint32_t i) {
    return db && i < db->files.size() ? db->files[i].num_regions : 0;
}
double gtars_igddb_file_avg_width(const gtars_igddb_t *db, uint32_t i) {
    return db && i < db->files.size() ? db->files[i].avg_width : 0.0;
}
int64_t gtars_igddb_chrom_id(const gtars_igddb_t *db, const char *chr) { return db && chr ? db->chroms.find(chr) : -1; }
const char *gtars_igddb_chrom_name(const gtars_igddb_t *db, uint32_t id) {
    return db && id < db->chroms.names.size() ? db->chroms.names[id].c_str() : nullptr;
}
const gtars_igd_t *gtars_igddb_engine(const gtars_igddb_t *db) { return db ? db->igd : nullptr; }

gtars_status gtars_igddb_count_regionset(const gtars_igddb_t *db, const gtars_regionset_t *rs, int32_t min_overlap,
                                         int binary, uint64_t *hits) {
    if (!db || !rs) return fail(GTARS_ERR_INVALID_ARG, "NULL argument");
    const std::vector<uint32_t> qc = translate_chroms(rs, db->chroms);
    return gtars_igd_count(db->igd, qc.data(), rs->starts.data(), rs->ends.data(), rs->size(), min_overlap, binary,
                           hits);
}

}  // extern "C"

// ===================================================================== .igd files
extern "C" {

static gtars_status gtars_igddb_from_arrays_impl(const char *const *chrom_names, uint32_t n_chrom, const uint32_t *chrom,
                                     const int32_t *start, const int32_t *end, const int32_t *value,
                                     const uint32_t *file_idx, uint64_t n, const char *const *file_names,
                                     const uint32_t *num_regions, const double *avg_width, uint32_t n_files,
                                     gtars_igddb_t **out) {
    if (!out || (n_chrom && !chrom_names) || (n && (!chrom || !start || !end || !file_idx)))
        return fail(GTARS_ERR_INVALID_ARG, "NULL argument");
    *out = nullptr;
    std::unique_ptr<gtars_igddb> db(new gtars_igddb());
    for (uint32_t c = 0; c < n_chrom; ++c)
        if (db->chroms.get_or_add(chrom_names[c] ? chrom_names[c] : "") != c)
            return fail(GTARS_ERR_INVALID_ARG, "duplicate chromosome name");
    uint32_t nf = n_files;
    for (uint64_t i = 0; i < n; ++i) nf = std::max(nf, file_idx[i] + 1);
    for (uint32_t f = 0; f < n_files; ++f)
        db->files.push_back({file_names && file_names[f] ? file_names[f] : "", num_regions ? num_regions[f] : 0u,
                             avg_width ? avg_width[f] : 0.0});
    gtars_status st = gtars_igd_build(chrom, start, end, value, file_idx, n, n_chrom, nf, &db->igd);
    if (st) return st;
    *out = db.release();
    return GTARS_OK;
}

static gtars_status gtars_igddb_save_impl(const gtars_igddb_t *db, const char *path, int32_t nbp) {
    if (!db || !path) return fail(GTARS_ERR_INVALID_ARG, "NULL argument");
    if (nbp <= 0) nbp = 16384;
    const uint64_t n = gtars_igd_len(db->igd);
    std::vector<uint32_t> c(n), f(n);
    std::vector<int32_t> s(n), e(n), v(n);
    gtars_status st = gtars_igd_export(db->igd, c.data(), s.data(), e.data(), v.data(), f.data());
    if (st) return st;
    const uint32_t n_ctg = (uint32_t)db->chroms.names.size();
    // tiles per contig = highest tile any record touches + 1 (Igd::add grows the tile vector, igd.rs:135-150)
    std::vector<int32_t> n_tiles(n_ctg, 0);
    for (uint64_t i = 0; i < n; ++i) n_tiles[c[i]] = std::max(n_tiles[c[i]], (e[i] - 1) / nbp + 1);
    std::vector<uint64_t> toff(n_ctg + 1, 0);
    for (uint32_t k = 0; k < n_ctg; ++k) toff[k + 1] = toff[k] + (uint64_t)n_tiles[k];
    // the stored order is (contig, start, insertion); a tile's records are those that touch it, in that order
    std::vector<int32_t> counts(toff[n_ctg], 0);
    for (uint64_t i = 0; i < n; ++i)
        for (int32_t t = s[i] / nbp; t <= (e[i] - 1) / nbp; ++t) counts[toff[c[i]] + (uint64_t)t]++;
    std::vector<uint64_t> pos(counts.size() + 1, 0);
    for (size_t t = 0; t < counts.size(); ++t) pos[t + 1] = pos[t] + (uint64_t)counts[t];
    std::vector<int32_t> rec(pos.back() * 4);
    {
        std::vector<uint64_t> fill(pos.begin(), pos.end() - 1);
        for (uint64_t i = 0; i < n; ++i)
            for (int32_t t = s[i] / nbp; t <= (e[i] - 1) / nbp; ++t) {
                int32_t *r = &rec[fill[toff[c[i]] + (uint64_t)t]++ * 4];
                r[0] = (int32_t)f[i];
                r[1] = s[i];
                r[2] = e[i];
                r[3] = v[i];
            }
    }
    const std::string p = path;
    {
        const std::string parent = parent_dir(p);
        std::string acc;
        for (size_t i = 0; i <= parent.size(); ++i) {
            if ((i == parent.size() || parent[i] == '/') && !acc.empty()) (void)mkdir(acc.c_str(), 0777);
            if (i < parent.size()) acc.push_back(parent[i]);
        }
    }
    FILE *fh = fopen(p.c_str(), "wb");
    if (!fh) return fail(GTARS_ERR_IO, "cannot create " + p + ": " + strerror(errno));
    const int32_t hdr[3] = {nbp, 1, (int32_t)n_ctg};
    bool ok = fwrite(hdr, 4, 3, fh) == 3;
    ok = ok && (n_ctg == 0 || fwrite(n_tiles.data(), 4, n_ctg, fh) == n_ctg);
    ok = ok && (counts.empty() || fwrite(counts.data(), 4, counts.size(), fh) == counts.size());
    for (uint32_t k = 0; ok && k < n_ctg; ++k) {
        char name[40] = {0};
        memcpy(name, db->chroms.names[k].data(), std::min<size_t>(40, db->chroms.names[k].size()));
        ok = fwrite(name, 1, 40, fh) == 40;
    }
    ok = ok && (rec.empty() || fwrite(rec.data(), 4, rec.size(), fh) == rec.size());
    ok = (fclose(fh) == 0) && ok;
    if (!ok) return fail(GTARS_ERR_IO, "write to " + p + " failed");
    // companion .tsv: Path::with_extension("tsv")
    const std::string stem_path = (parent_dir(p).empty() ? "" : parent_dir(p) + "/") + file_stem(p);
    FILE *th = fopen((stem_path + ".tsv").c_str(), "w");
    if (!th) return fail(GTARS_ERR_IO, "cannot create " + stem_path + ".tsv");
    fprintf(th, "Index\tFile\tNumber of Regions\tAvg size\n");
    for (size_t i = 0; i < db->files.size(); ++i)
        fprintf(th, "%zu\t%s\t%u\t%.2f\n", i, db->files[i].filename.c_str(), db->files[i].num_regions, db->files[i].avg_width);
    fclose(th);
    return GTARS_OK;
}

static gtars_status gtars_igddb_load_impl(const char *path, gtars_igddb_t **out, int32_t *nbp_out) {
    if (!path || !out) return fail(GTARS_ERR_INVALID_ARG, "NULL argument");
    *out = nullptr;
    FILE *fh = fopen(path, "rb");
    if (!fh) return fail(GTARS_ERR_IO, std::string("Failed to open file: \"") + path + "\": " + strerror(errno));
    std::string data;
    {
        std::vector<char> buf(1 << 20);
        size_t k;
        while ((k = fread(buf.data(), 1, buf.size(), fh)) > 0) data.append(buf.data(), k);
        fclose(fh);
    }
    auto need = [&](size_t pos, size_t bytes) { return pos + bytes <= data.size(); };
    if (!need(0, 12)) return fail(GTARS_ERR_PARSE, "truncated .igd header");
    int32_t hdr[3];
    memcpy(hdr, data.data(), 12);
    const int32_t nbp = hdr[0], g_type = hdr[1], n_ctg = hdr[2];
    if (nbp <= 0 || n_ctg < 0 || (g_type != 0 && g_type != 1)) return fail(GTARS_ERR_PARSE, "bad .igd header");
    size_t pos = 12;
    if (!need(pos, (size_t)n_ctg * 4)) return fail(GTARS_ERR_PARSE, "truncated .igd (tile counts)");
    std::vector<int32_t> n_tiles(n_ctg);
    memcpy(n_tiles.data(), data.data() + pos, (size_t)n_ctg * 4);
    pos += (size_t)n_ctg * 4;
    uint64_t total_tiles = 0;
    for (int32_t t : n_tiles) {
        if (t < 0) return fail(GTARS_ERR_PARSE, "bad .igd tile count");
        total_tiles += (uint64_t)t;
    }
    if (!need(pos, total_tiles * 4)) return fail(GTARS_ERR_PARSE, "truncated .igd (record counts)");
    std::vector<int32_t> counts(total_tiles);
    memcpy(counts.data(), data.data() + pos, total_tiles * 4);
    pos += total_tiles * 4;
    std::vector<std::string> names;
    if (!need(pos, (size_t)n_ctg * 40)) return fail(GTARS_ERR_PARSE, "truncated .igd (contig names)");
    for (int32_t k = 0; k < n_ctg; ++k) {
        const char *nm = data.data() + pos + (size_t)k * 40;
        names.emplace_back(nm, strnlen(nm, 40));
    }
    pos += (size_t)n_ctg * 40;
    const size_t w = g_type == 0 ? 3 : 4;
    uint64_t nrec = 0;
    for (int32_t cnt : counts) {
        if (cnt < 0) return fail(GTARS_ERR_PARSE, "bad .igd record count");
        nrec += (uint64_t)cnt;
    }
    if (!need(pos, nrec * w * 4)) return fail(GTARS_ERR_PARSE, "truncated .igd (records)");
    std::vector<uint32_t> c, f;
    std::vector<int32_t> s, e, v;
    {
        const char *rp = data.data() + pos;
        uint64_t tile = 0;
        for (int32_t k = 0; k < n_ctg; ++k)
            for (int32_t t = 0; t < n_tiles[k]; ++t, ++tile)
                for (int32_t r = 0; r < counts[tile]; ++r, rp += w * 4) {
                    int32_t rr[4] = {0, 0, 0, 0};
                    memcpy(rr, rp, w * 4);
                    if (rr[1] / nbp != t) continue;  // a replica: the record starts in an earlier tile
                    c.push_back((uint32_t)k);
                    f.push_back((uint32_t)rr[0]);
                    s.push_back(rr[1]);
                    e.push_back(rr[2]);
                    v.push_back(rr[3]);
                }
    }
    std::unique_ptr<gtars_igddb> db(new gtars_igddb());
    for (const std::string &nm : names) db->chroms.get_or_add(nm);
    // companion .tsv (igd.rs:386-410): Index, File, Number of Regions, Avg size; unparsable numbers read as 0
    {
        const std::string p = path;
        const std::string tsv = (parent_dir(p).empty() ? "" : parent_dir(p) + "/") + file_stem(p) + ".tsv";
        std::string text, err;
        if (is_regular_file(tsv) && read_all(tsv + "", text, err)) {
            size_t lp = 0, li = 0;
            while (lp < text.size()) {
                size_t nl = text.find('\n', lp);
                if (nl == std::string::npos) nl = text.size();
                std::string line = text.substr(lp, nl - lp);
                lp = nl + 1;
                if (li++ == 0) continue;
                std::vector<std::string> fld;
                size_t a = 0;
                for (;;) {
                    const size_t t = line.find('\t', a);
                    fld.push_back(line.substr(a, t == std::string::npos ? std::string::npos : t - a));
                    if (t == std::string::npos) break;
                    a = t + 1;
                }
                if (fld.size() < 4) continue;
                auto trim = [](std::string x) {
                    size_t b = 0, e2 = x.size();
                    while (b < e2 && is_ws(x[b])) ++b;
                    while (e2 > b && is_ws(x[e2 - 1])) --e2;
                    return x.substr(b, e2 - b);
                };
                uint32_t nr = 0;
                const std::string nrs = trim(fld[2]);
                if (!parse_u32_view(nrs.data(), nrs.size(), nr)) nr = 0;
                char *endp = nullptr;
                const std::string aws = trim(fld[3]);
                double aw = strtod(aws.c_str(), &endp);
                if (endp == aws.c_str() || *endp) aw = 0.0;
                db->files.push_back({trim(fld[1]), nr, aw});
            }
        }
    }
    uint32_t nf = (uint32_t)db->files.size();
    for (uint32_t x : f) nf = std::max(nf, x + 1);
    gtars_status st = gtars_igd_build(c.data(), s.data(), e.data(), v.data(), f.data(), c.size(), (uint32_t)n_ctg, nf, &db->igd);
    if (st) return st;
    if (nbp_out) *nbp_out = nbp;
    *out = db.release();
    return GTARS_OK;
}

}  // extern "C"

// ===================================================================== gtars-fragsplit
namespace {

// remove_all_extensions (gtars-core/src/utils.rs:372-387): the file name without ANY extension
std::string remove_all_extensions(const std::string &path) {
    std::string stem = file_stem(path);
    for (;;) {
        const size_t dot = stem.find_last_of('.');
        if (dot == std::string::npos || dot == 0) return stem;  // Path::extension() is None
        stem = stem.substr(0, dot);
    }
}

struct SvHash {
    size_t operator()(const std::string &s) const noexcept {
        uint64_t h = 1469598103934665603ull;  // FNV-1a
        for (unsigned char ch : s) h = (h ^ ch) * 1099511628211ull;
        return (size_t)h;
    }
};

gtars_status list_regular_files(const std::string &dir, std::vector<std::string> &out) {
    DIR *d = opendir(dir.c_str());
    if (!d) return fail(GTARS_ERR_IO, "There was an error reading the specifed fragment file directory: \"" + dir + "\"");
    while (struct dirent *e = readdir(d)) {
        const std::string name = e->d_name;
        if (name == "." || name == "..") continue;
        const std::string p = dir + "/" + name;
        if (is_regular_file(p)) out.push_back(p);
    }
    closedir(d);
    std::sort(out.begin(), out.end());
    return GTARS_OK;
}

}  // namespace

struct gtars_barcode_map {
    std::unordered_map<std::string, uint32_t, SvHash> map;  // "stem+barcode" -> cluster index (into labels)
    std::vector<std::string> labels;                        // byte order
    // the keys in byte order (pointers into `map`: node keys do not move): the keys of ONE fragment file -- those that start
    // with "stem+" -- are a range of it, which split_one_file turns into a small per-file table keyed by the barcode alone
    std::vector<std::pair<const std::string *, uint32_t>> sorted;
    void index_keys() {
        sorted.clear();
        sorted.reserve(map.size());
        for (const auto &kv : map) sorted.emplace_back(&kv.first, kv.second);
        std::sort(sorted.begin(), sorted.end(), [](const auto &a, const auto &b) { return *a.first < *b.first; });
    }
};

namespace {
// barcode -> (cluster, id of the barcode among its cluster's barcodes in this file, in first-seen order) for ONE fragment file:
// what "{stem}+{barcode}" resolves to in the map (split.rs:100-106), looked up by the barcode's bytes alone.  A few hundred
// entries that stay in the parsing core's cache, where the map of a 10,000-file folder (4M keys) is a cache miss per line; and
// one lookup per line instead of two (cluster, then barcode id).
struct FileBarcodes {
    struct Entry {
        const char *bc = nullptr;
        uint32_t n = 0, cluster = 0, local = 0xFFFFFFFFu;
    };
    std::vector<Entry> slots;
    size_t mask = 0;
    void build(const gtars_barcode_map &m, const std::string &prefix) {  // prefix = stem + "+"
        auto lo = std::lower_bound(m.sorted.begin(), m.sorted.end(), prefix,
                                   [](const auto &a, const std::string &p) { return *a.first < p; });
        auto hi = lo;
        while (hi != m.sorted.end() && hi->first->size() >= prefix.size() && hi->first->compare(0, prefix.size(), prefix) == 0) ++hi;
        size_t cap = 16;
        while (cap < (size_t)(hi - lo) * 2) cap <<= 1;
        slots.assign(cap, Entry());
        mask = cap - 1;
        for (auto it = lo; it != hi; ++it) {
            Entry e;
            e.bc = it->first->data() + prefix.size();
            e.n = (uint32_t)(it->first->size() - prefix.size());
            e.cluster = it->second;
            size_t k = ViewDict::hash(e.bc, e.n) & mask;
            while (slots[k].bc) k = (k + 1) & mask;
            slots[k] = e;
        }
    }
    Entry *find(const char *p, size_t n) {
        size_t k = ViewDict::hash(p, n) & mask;
        while (slots[k].bc) {
            if (slots[k].n == n && memcmp(slots[k].bc, p, n) == 0) return &slots[k];
            k = (k + 1) & mask;
        }
        return nullptr;
    }
};
}  // namespace

extern "C" {

gtars_status gtars_barcode_map_from_file(const char *path, gtars_barcode_map_t **out) {
    if (!path || !out) return fail(GTARS_ERR_INVALID_ARG, "NULL argument");
    *out = nullptr;
    FILE *probe = fopen(path, "rb");
    if (!probe) return fail(GTARS_ERR_IO, std::string("Couldn't open file: \"") + path + "\"");
    fclose(probe);
    std::string data, err;
    {
        // BufReader::new(File::open(..)): never gunzipped, whatever the extension
        FILE *f = fopen(path, "rb");
        std::vector<char> buf(1 << 20);
        size_t n;
        while ((n = fread(buf.data(), 1, buf.size(), f)) > 0) data.append(buf.data(), n);
        fclose(f);
    }
    std::vector<std::pair<std::string, std::string>> rows;
    const char *p = data.data(), *end = p + data.size();
    while (p < end) {
        const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
        const char *le = nl ? nl : end;
        const char *next = nl ? nl + 1 : end;
        if (nl && le > p && le[-1] == '\r') --le;
        const char *q = p;
        std::string f[2];
        int nf = 0;
        while (q < le && nf < 2) {
            while (q < le && is_ws(*q)) ++q;
            const char *st = q;
            while (q < le && !is_ws(*q)) ++q;
            if (q > st) f[nf++].assign(st, (size_t)(q - st));
        }
        if (nf < 2)
            return fail(GTARS_ERR_PARSE, "Invalid line format: Expected two tab-separated values, found: \"" +
                                             std::string(p, (size_t)(le - p)) + "\"");
        rows.emplace_back(std::move(f[0]), std::move(f[1]));
        p = next;
    }
    std::unique_ptr<gtars_barcode_map> m(new gtars_barcode_map());
    std::set<std::string> labels;
    for (auto &r : rows) labels.insert(r.second);
    m->labels.assign(labels.begin(), labels.end());
    std::unordered_map<std::string, uint32_t> lid;
    for (uint32_t i = 0; i < m->labels.size(); ++i) lid[m->labels[i]] = i;
    m->map.reserve(rows.size() * 2);
    for (auto &r : rows) m->map[r.first] = lid[r.second];  // HashMap::insert: the later line wins
    m->index_keys();
    *out = m.release();
    return GTARS_OK;
}
void gtars_barcode_map_free(gtars_barcode_map_t *m) { delete m; }
uint64_t gtars_barcode_map_len(const gtars_barcode_map_t *m) { return m ? m->map.size() : 0; }
uint32_t gtars_barcode_map_n_clusters(const gtars_barcode_map_t *m) { return m ? (uint32_t)m->labels.size() : 0; }
const char *gtars_barcode_map_cluster_label(const gtars_barcode_map_t *m, uint32_t i) {
    return m && i < m->labels.size() ? m->labels[i].c_str() : nullptr;
}
const char *gtars_barcode_map_lookup(const gtars_barcode_map_t *m, const char *key) {
    if (!m || !key) return nullptr;
    auto it = m->map.find(key);
    return it == m->map.end() ? nullptr : m->labels[it->second].c_str();
}

}  // extern "C"

namespace {

// One input file routed by cluster.  TEXT: the output lines per cluster; COLS: the routed lines as fragment columns
// (chromosome / barcode names as views into `data`, which stays alive with the result).
struct SplitFile {
    std::string data;
    std::vector<std::string> text;                           // [n_clusters]
    // fused pipeline: per cluster the routed fragments as columns -- chromosome ids of the tokenizer's dictionary, barcode ids
    // LOCAL to this file and cluster (first-seen order; `barcodes.names`).  Filled by the thread that parses the file, while the
    // line is in its cache: resolving the strings per cluster afterwards, by other threads, cost two cache misses per row (as
    // much wall time as gunzip + parse: 54 of 129 ms for 48 files x 1e5 fragments).
    struct Cols {
        std::vector<uint32_t> c, s, e, b;
        ViewDict barcodes;
    };
    std::vector<Cols> cols;                                  // [n_clusters]
    uint64_t n_reads = 0, n_written = 0;
    gtars_status st = GTARS_OK;
    std::string err;
};

void split_one_file(const std::string &path, const gtars_barcode_map &m, bool want_text, SplitFile &out, const Dict *chroms) {
    std::string err;
    if (!read_all(path, out.data, err)) {
        out.st = GTARS_ERR_IO;
        out.err = err;
        return;
    }
    const size_t nc = m.labels.size();
    if (want_text) out.text.resize(nc); else out.cols.resize(nc);
    const char *lc = nullptr;  // the last chromosome name looked up (fragment files are sorted: it rarely changes)
    size_t lcn = 0;
    uint32_t lcid = 0;
    FileBarcodes mine;
    mine.build(m, remove_all_extensions(path) + "+");
    const char *p = out.data.data(), *end = p + out.data.size();
    size_t index = 0;
    while (p < end) {
        const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
        const char *le = nl ? nl : end;
        const char *next = nl ? nl + 1 : end;
        if (nl && le > p && le[-1] == '\r') --le;
        const char *f[5];
        size_t fl[5];
        int nf = 0;
        const char *q = p;
        while (q < le && nf < 5) {
            while (q < le && is_ws(*q)) ++q;
            const char *st = q;
            while (q < le && !is_ws(*q)) ++q;
            if (q > st) { f[nf] = st; fl[nf] = (size_t)(q - st); ++nf; }
        }
        if (nf < 5) {
            out.st = GTARS_ERR_PARSE;
            out.err = "Failed to parse fragments file at line " + std::to_string(index) + ": " + std::string(p, (size_t)(le - p));
            return;
        }
        if (FileBarcodes::Entry *hit = mine.find(f[3], fl[3])) {  // else: most likely a cell dropped in QC
            const uint32_t cl = hit->cluster;
            if (want_text) {
                std::string &t = out.text[cl];
                for (int k = 0; k < 5; ++k) {
                    t.append(f[k], fl[k]);
                    t.push_back(k == 4 ? '\n' : '\t');
                }
            } else if (f[0][0] != '#') {  // tokenize_fragment_file skips '#' lines of the cluster file
                uint32_t rs = 0, re = 0;
                if (!parse_u32_view(f[1], fl[1], rs) || !parse_u32_view(f[2], fl[2], re)) {
                    out.st = GTARS_ERR_PARSE;
                    out.err = std::string("Failed to parse ") + (parse_u32_view(f[1], fl[1], rs) ? "end" : "start") +
                              " position of a routed fragment (" + path + " line " + std::to_string(index) + ")";
                    return;
                }
                if (!(lc && lcn == fl[0] && memcmp(lc, f[0], lcn) == 0)) {
                    const int64_t cid = chroms ? chroms->find(std::string(f[0], fl[0])) : -1;
                    lcid = cid < 0 ? GTARS_UNKNOWN_CHROM : (uint32_t)cid;
                    lc = f[0];
                    lcn = fl[0];
                }
                SplitFile::Cols &k = out.cols[cl];
                k.c.push_back(lcid);
                k.s.push_back(rs);
                k.e.push_back(re);
                if (hit->local == 0xFFFFFFFFu) {  // the barcode's first line in this file
                    hit->local = (uint32_t)k.barcodes.names.size();
                    k.barcodes.names.emplace_back(f[3], fl[3]);
                }
                k.b.push_back(hit->local);
            }
            ++out.n_written;
        }
        ++out.n_reads;
        ++index;
        p = next;
    }
}

// One input file for the DEVICE path of the fused pipeline (fragparse.hip): the inflated text as it is -- the GPU splits and
// parses it -- and the file's barcode table in the device's format (FileBarcodes above, keyed by the barcode alone).
// The inflated text of a file in a block of the pinned pool (frag_device.h): the host thread's decoder writes into memory the GPU's
// copy engine reads directly.  The interface the decoders need of a std::string; no zero fill on resize; a buffer that must
// grow moves to a larger block.
struct TextBuf {
    gtars::HostBlock b;
    size_t n = 0;
    size_t size() const { return n; }
    bool empty() const { return n == 0; }
    char *data() const { return (char *)b.p; }
    char &operator[](size_t i) const { return ((char *)b.p)[i]; }
    char back() const { return ((char *)b.p)[n - 1]; }
    void clear() { n = 0; }
    void reserve(size_t m) {
        if (m <= b.cap) return;
        gtars::HostBlock nb;
        if (!nb.alloc(m + m / 4)) throw std::bad_alloc();
        if (n) memcpy(nb.p, b.p, n);
        b = std::move(nb);
    }
    void resize(size_t m) {
        reserve(m);
        n = m;
    }
    void push_back(char c) {
        reserve(n + 1);
        ((char *)b.p)[n++] = c;
    }
    void assign(const std::string &s) {
        resize(s.size());
        if (n) memcpy(b.p, s.data(), n);
    }
    void release() {
        b.reset();
        n = 0;
    }
};

struct TextFile {
    TextBuf data;
    std::vector<gtars::FragGzMember> members;  // gzip members whose CRC-32 the device still has to check
    std::vector<gtars::FragSlot> slots;
    std::string keys;
    gtars_status st = GTARS_OK;
    std::string err;
};

void load_text_file(const std::string &path, const gtars_barcode_map &m, TextFile &out) {
    std::string err;
    bool have = false;
    if (extension_of(path) == "gz" && !cfg_get("GTARS_FRAG_HOST_CRC")) {  // (the switch: A/B and tests)
        std::string raw;
        size_t n_raw = 0;
        if (read_file_padded(path, raw, n_raw)) have = inflate_gzip_members_raw((const unsigned char *)raw.data(), n_raw, out.data, out.members);
    }
    if (!have) {
        out.members.clear();
        std::string plain;
        if (!read_all(path, plain, err)) {  // (everything else, and whatever the raw reader did not like: zlib's own checks)
            out.st = GTARS_ERR_IO;
            out.err = err;
            return;
        }
        out.data.assign(plain);
    }
    if (!out.data.empty() && out.data.back() != '\n') out.data.push_back('\n');  // (BufRead::lines: a last line without one still counts)
    const std::string prefix = remove_all_extensions(path) + "+";
    auto lo = std::lower_bound(m.sorted.begin(), m.sorted.end(), prefix, [](const auto &a, const std::string &p) { return *a.first < p; });
    auto hi = lo;
    while (hi != m.sorted.end() && hi->first->size() >= prefix.size() && hi->first->compare(0, prefix.size(), prefix) == 0) ++hi;
    size_t cap = 1;
    while (cap < (size_t)(hi - lo) * 2 + 1) cap <<= 1;
    out.slots.assign(cap, gtars::FragSlot{0, 0, 0, 0});
    for (auto it = lo; it != hi; ++it) {
        const char *bc = it->first->data() + prefix.size();
        const uint32_t n = (uint32_t)(it->first->size() - prefix.size());
        if (!n) continue;  // (a field is never empty)
        uint32_t k = gtars::frag_hash(bc, n) & (uint32_t)(cap - 1);
        while (out.slots[k].len) k = (k + 1) & (uint32_t)(cap - 1);
        out.slots[k] = gtars::FragSlot{(uint32_t)out.keys.size(), n, it->second, 0};
        out.keys.append(bc, n);
    }
}

// The files inflated by a pool of host threads that never waits for a wave to complete (round 5, second cut): threads take the
// next file as they finish one, at most `window` files ahead of what has been handed on, and the CALLING thread hands batches of
// consecutive loaded files to `sink(first_file_index, batch)` in file order -- a new batch whenever `wait_idle()` says that the
// consumer (the device thread) has nothing to do, holding whatever has been loaded by then: batches size themselves to the
// consumer's pace, and behind the last file only the files loaded since the previous batch are left to do.  `failed()`: the
// consumer has met an error (stop producing).  A file that could not be read ends the stream with ITS error -- every earlier file
// has been handed on by then, so an earlier file's parse error (known once the consumer is done) still comes first.
// `too_large`: a single file beyond the device parser's text limit.
template <class Sink, class Idle, class Failed>
gtars_status stream_text_files(const std::vector<std::string> &files, const gtars_barcode_map &m, uint64_t byte_limit, int device, Sink &&sink,
                               Idle &&wait_idle, Failed &&failed, bool *too_large) {
    const size_t n = files.size();
    if (!n) return GTARS_OK;
    const unsigned nt = (unsigned)std::max<size_t>(1, std::min<size_t>(host_thread_budget(64), n));
    const size_t window = (size_t)nt * 4, max_batch = std::min<size_t>((size_t)nt * 4, 60000);
    // A batch's text: 24 MB at most.  The loaders finish their last files together -- a third of a 48-file folder is ready at once --
    // and ONE batch of it keeps one device thread busy for 3.6 ms behind the last file while the other has nothing to do; smaller
    // batches run side by side and the last one is short (same box, median of 7 calls: 14.7 ms uncapped, 14.6 at 40 MB, 12.8 at
    // 20 MB; below ~10 MB a batch's fixed costs, ~0.5 ms of launches and waits, would show).
    const char *bmb = cfg_get("GTARS_FRAG_BATCH_MB");
    const uint64_t batch_bytes = (bmb ? (uint64_t)std::max(1ll, atoll(bmb)) : 24ull) << 20;
    std::vector<TextFile> tf(n);
    std::vector<char> done(n, 0);
    std::mutex mx;
    std::condition_variable cvx;
    size_t consumed = 0;
    bool stop = false;
    std::atomic<size_t> next{0};
    auto loader = [&] {
        // (the texts are inflated into pinned blocks: allocated against the CALLER's device, not this new thread's default device 0)
        (void)gtars::frag_select_device(device);
        for (;;) {
            const size_t i = next.fetch_add(1);
            if (i >= n) return;
            {
                std::unique_lock<std::mutex> lk(mx);
                cvx.wait(lk, [&] { return stop || i < consumed + window; });
                if (stop) return;
            }
            try {
                load_text_file(files[i], m, tf[i]);
            } catch (const std::exception &ex) {  // (out of memory on a loader thread: the file's error, not the process's end)
                tf[i].st = GTARS_ERR_INTERNAL;
                tf[i].err = std::string("fragment pipeline: ") + ex.what();
            }
            {
                std::lock_guard<std::mutex> lk(mx);
                done[i] = 1;
            }
            cvx.notify_all();
        }
    };
    std::vector<std::thread> th;
    for (unsigned k = 0; k < nt; ++k) th.emplace_back(loader);
    struct Join {
        std::vector<std::thread> &th;
        std::mutex &mx;
        std::condition_variable &cvx;
        bool &stop;
        ~Join() {
            {
                std::lock_guard<std::mutex> lk(mx);
                stop = true;
            }
            cvx.notify_all();
            for (auto &t : th)
                if (t.joinable()) t.join();
        }
    } join{th, mx, cvx, stop};
    gtars_status st = GTARS_OK;
    size_t lo = 0;
    while (lo < n) {
        wait_idle();
        if (failed()) break;
        size_t hi = lo;
        uint64_t bytes = 0;
        {
            std::unique_lock<std::mutex> lk(mx);
            cvx.wait(lk, [&] { return done[lo] != 0; });
            // (a batch is at least one file -- of up to byte_limit --, and grows while it stays below batch_bytes)
            while (hi < n && done[hi] && !tf[hi].st && hi - lo < max_batch &&
                   (hi == lo ? tf[hi].data.size() < byte_limit : bytes + tf[hi].data.size() < std::min(byte_limit, batch_bytes))) {
                bytes += tf[hi].data.size();
                ++hi;
            }
        }
        if (hi == lo) {
            if (tf[lo].st) {
                st = fail(tf[lo].st, tf[lo].err);
            } else {
                *too_large = true;
                st = fail(GTARS_ERR_INTERNAL, "fragment file too large for the device parser");
            }
            break;
        }
        std::vector<TextFile> batch;
        batch.reserve(hi - lo);
        for (size_t i = lo; i < hi; ++i) batch.push_back(std::move(tf[i]));
        st = sink(lo, batch);
        if (st) break;
        lo = hi;
        {
            std::lock_guard<std::mutex> lk(mx);
            consumed = lo;
        }
        cvx.notify_all();
    }
    return st;
}

// files in waves of `wave` (parsed in parallel), handed to `sink(first_file_index, wave_results)` in file order
template <class Sink>
gtars_status for_each_split_wave(const std::vector<std::string> &files, const gtars_barcode_map &m, bool want_text, const Dict *chroms,
                                 Sink &&sink) {
    const unsigned nt = (unsigned)std::max<size_t>(1, std::min<size_t>(host_thread_budget(64), files.size()));
    const size_t wave = (size_t)nt * 2;
    for (size_t base = 0; base < files.size(); base += wave) {
        const size_t n = std::min(wave, files.size() - base);
        std::vector<SplitFile> res(n);
        std::atomic<size_t> next{0};
        auto work = [&] {
            for (size_t i = next.fetch_add(1); i < n; i = next.fetch_add(1)) split_one_file(files[base + i], m, want_text, res[i], chroms);
        };
        std::vector<std::thread> th;
        for (unsigned k = 1; k < nt; ++k) th.emplace_back(work);
        work();
        for (auto &t : th) t.join();
        for (size_t i = 0; i < n; ++i)
            if (res[i].st) return fail(res[i].st, res[i].err);
        gtars_status st = sink(base, res);
        if (st) return st;
    }
    return GTARS_OK;
}

}  // namespace

extern "C" {

static gtars_status gtars_fragsplit_impl(const char *files_dir, const gtars_barcode_map_t *m, const char *out_dir, uint64_t *n_reads,
                             uint64_t *n_written) {
    if (!files_dir || !m || !out_dir) return fail(GTARS_ERR_INVALID_ARG, "NULL argument");
    std::vector<std::string> files;
    gtars_status st = list_regular_files(files_dir, files);
    if (st) return st;
    {
        // fs::create_dir_all
        std::string acc;
        const std::string od = out_dir;
        for (size_t i = 0; i <= od.size(); ++i) {
            if (i == od.size() || od[i] == '/') {
                if (!acc.empty() && mkdir(acc.c_str(), 0777) != 0 && errno != EEXIST)
                    return fail(GTARS_ERR_IO, "There was an error creating the output directory: \"" + od + "\"");
            }
            if (i < od.size()) acc.push_back(od[i]);
        }
    }
    const size_t nc = m->labels.size();
    std::vector<gzFile> outs(nc, nullptr);
    auto close_all = [&] {
        for (gzFile f : outs)
            if (f) gzclose(f);
    };
    for (size_t c = 0; c < nc; ++c) {
        const std::string p = std::string(out_dir) + "/cluster_" + m->labels[c] + ".bed.gz";
        outs[c] = gzopen(p.c_str(), "wb6");  // flate2 Compression::default()
        if (!outs[c]) {
            close_all();
            return fail(GTARS_ERR_IO, "cannot create " + p);
        }
        gzbuffer(outs[c], 1 << 18);
    }
    uint64_t reads = 0, written = 0;
    st = for_each_split_wave(files, *m, true, nullptr, [&](size_t, std::vector<SplitFile> &res) -> gtars_status {
        // every cluster's stream is compressed by one thread per wave, the wave's files in order
        std::atomic<size_t> next{0};
        std::atomic<int> bad{0};
        auto work = [&] {
            for (size_t c = next.fetch_add(1); c < nc; c = next.fetch_add(1))
                for (SplitFile &f : res)
                    if (!f.text[c].empty() && gzwrite(outs[c], f.text[c].data(), (unsigned)f.text[c].size()) <= 0) bad = 1;
        };
        const unsigned nt = (unsigned)std::max<size_t>(1, std::min<size_t>(host_thread_budget(64), nc));
        std::vector<std::thread> th;
        for (unsigned k = 1; k < nt; ++k) th.emplace_back(work);
        work();
        for (auto &t : th) t.join();
        for (SplitFile &f : res) {
            reads += f.n_reads;
            written += f.n_written;
        }
        return bad ? fail(GTARS_ERR_IO, "write to a cluster file failed") : GTARS_OK;
    });
    close_all();
    if (st) return st;
    if (n_reads) *n_reads = reads;
    if (n_written) *n_written = written;
    return GTARS_OK;
}

// the pipeline over an explicit list of files, visited in the order given (the directory form passes its regular files in byte
// order of their names; a rank of the sharded driver passes its run of that list)
static thread_local double g_frag_stages[12];
void gtars_fragsplit_last_stages(double *out12) {
    if (out12) memcpy(out12, g_frag_stages, sizeof g_frag_stages);
}

static gtars_status fragsplit_tokenize_mode(const gtars_tokenizer_t *t, const std::vector<std::string> &files, const gtars_barcode_map_t *m,
                                            gtars_fragment_tokens_t ***out, uint64_t *n_reads, bool allow_device, bool *redo_on_host);

static gtars_status fragsplit_tokenize_core(const gtars_tokenizer_t *t, const std::vector<std::string> &files, const gtars_barcode_map_t *m,
                                            gtars_fragment_tokens_t ***out, uint64_t *n_reads) {
    bool redo = false;
    gtars_status st = fragsplit_tokenize_mode(t, files, m, out, n_reads, true, &redo);
    // (a wave of more than 3.5 GiB of text -- beyond the device parser's 32-bit positions: the whole call on the host parser)
    if (redo) st = fragsplit_tokenize_mode(t, files, m, out, n_reads, false, &redo);
    return st;
}

static gtars_status fragsplit_tokenize_mode(const gtars_tokenizer_t *t, const std::vector<std::string> &files, const gtars_barcode_map_t *m,
                                            gtars_fragment_tokens_t ***out, uint64_t *n_reads, bool allow_device, bool *redo_on_host) {
    gtars_status st = GTARS_OK;
    *redo_on_host = false;
    const double t_enter = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    const size_t nc = m->labels.size();
    // per cluster: fragment columns in the order the cluster file would have them (files in order, lines in order),
    // chromosome ids of the TOKENIZER's dictionary, barcode ids in first-seen order
    struct Cluster {
        std::vector<uint32_t> b;  // barcode ids (the cluster's first-seen order) of its fragments, wave after wave
        ViewDict barcodes;
    };
    // a (file, barcode) run of regrouped ids in a device wave's answer: which line opens it, the file and the barcode's slot in the
    // file's table, where the run lies in the wave's id array
    struct Run {
        uint32_t line, file, slot, start, len;
    };
    // one wave's routed fragments as tokenizer input, cluster after cluster, and what the tokenizer made of them
    struct Wave {
        std::vector<uint64_t> coff;     // [n_clusters + 1]
        uint64_t n = 0;                 // fragments
        // columns and CSR offsets [n + 1]: plain allocations -- a std::vector would zero 20 bytes per fragment on the thread that
        // sits between two waves (10 waves of 2.4M fragments: 60 of the 100 ms that stage took)
        std::unique_ptr<uint32_t[]> c, s, e;
        std::unique_ptr<uint64_t[]> off;
        uint32_t *ids = nullptr;
        gtars_status st = GTARS_OK;
        std::string err;
        // device path (fragparse.hip): the wave's files as text until the GPU has them, afterwards their barcode tables; file and
        // barcode slot of every tokenized fragment (cluster-major like the CSR)
        bool device = false;
        size_t first_file = 0;
        std::vector<TextFile> tf;
        gtars::FragWaveOut dev;  // the device's answer: ids regrouped by (file, barcode), where every run starts, which line opens it
        // ... cut into its runs per cluster, in line order: made by the device thread that received the answer, while later waves are
        // still inflating (round 6: this pass over every slot of every file ran on the calling thread behind the last wave --
        // 11 ms of a 38-ms call on a folder of 1000 files)
        std::vector<std::vector<Run>> runs;
        uint64_t reads = 0;
        double td[5] = {0, 0, 0, 0, 0};
    };
    std::vector<Cluster> cl(nc);
    std::deque<Wave> waves;
    // The device path: the host threads only inflate; line splitting, field parsing, barcode and chromosome lookup, the grouping by
    // cluster and the tokenizer run on the GPU (fragparse.hip).  GTARS_FRAG_HOST_PARSE=1 keeps round 4's host parse (A/B, tests).
    const bool device_path = allow_device && !cfg_get("GTARS_FRAG_HOST_PARSE") && nc < 65000;
    gtars::FragChroms *d_chroms = nullptr;  // (the tokenizer's: calls of several threads on one tokenizer share it read-only)
    if (device_path) {
        st = t->device_chroms(&d_chroms);
        if (st) return st;
    }
    const int pipeline_device = t->device();  // the INDEX's device (round 5: the calling thread's current device)
    uint64_t reads = 0, n_all = 0;
    const bool timing = cfg_get("GTARS_HOST_TIMING") != nullptr;  // stderr: seconds per stage (tools/fragsplit_bench.py)
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_begin = now();
    double t_append = 0, t_tok = 0;
    auto over_clusters = [&](auto &&body) {  // body(c) for every cluster, on up to 64 threads
        std::atomic<size_t> next{0};
        auto work = [&] {
            for (size_t c = next.fetch_add(1); c < nc; c = next.fetch_add(1)) body(c);
        };
        const unsigned nt = (unsigned)std::max<size_t>(1, std::min<size_t>(host_thread_budget(64), nc));
        std::vector<std::thread> th;
        for (unsigned k = 1; k < nt; ++k) th.emplace_back(work);
        work();
        for (auto &tt : th) tt.join();
    };
    // The tokenizer calls (one per wave: copy in, kernel, copy out) run on ONE helper thread, so that a wave is tokenized while
    // the next one is being inflated and parsed (folders of more files than one wave holds: the config's 10,000).
    std::mutex mu;
    std::condition_variable cv;
    std::deque<Wave *> jobs;
    bool closing = false;
    bool device_overflow = false;  // a device wave with more than 4e9 ids
    size_t in_flight = 0;       // waves queued or on the device
    bool wave_failed = false;   // a wave ended with an error (the producer stops)
    auto tok_body = [&] {
        (void)gtars::frag_select_device(pipeline_device);  // (HIP's current device belongs to the thread and starts at 0)
        for (;;) {
            Wave *w = nullptr;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return closing || !jobs.empty(); });
                if (jobs.empty()) return;
                w = jobs.front();
                jobs.pop_front();
            }
            const double t0 = now();
            try {  // (the body allocates: an exception on this thread must end the wave with a status, not the process)
            if (w->device) {
                std::vector<gtars::FragFileIn> in;
                for (TextFile &f : w->tf) {
                    gtars::FragFileIn x;
                    x.text = f.data.data();
                    x.n = f.data.size();
                    x.slots = f.slots.data();
                    x.n_slots = (uint32_t)f.slots.size();
                    x.keys = f.keys.data();
                    x.n_key_bytes = (uint32_t)f.keys.size();
                    x.members = f.members.data();
                    x.n_members = (uint32_t)f.members.size();
                    in.push_back(x);
                }
                gtars::FragWaveOut o;
                w->st = gtars::frag_wave_device(t->index, d_chroms, in, (uint32_t)nc, t->unk_id, o);
                if (w->st == GTARS_ERR_CAPACITY) {  // more ids than the device path's 32-bit positions: the whole call on the host parser
                    std::lock_guard<std::mutex> lk(mu);
                    device_overflow = true;
                }
                if (w->st) {
                    w->err = gtars_last_error();
                } else if (o.first_error_file >= 0) {
                    // a line the reference fails on: its message from the host parser, on that one file
                    SplitFile again;
                    split_one_file(files[w->first_file + (size_t)o.first_error_file], *m, false, again, &t->chroms);
                    w->st = again.st ? again.st : GTARS_ERR_INTERNAL;
                    w->err = again.st ? again.err : "fragment pipeline: the device parser rejected a line that the host parser accepts";
                } else {
                    w->n = o.n;
                    for (uint64_t r : o.n_reads) w->reads += r;
                    w->td[0] = o.t_h2d, w->td[1] = o.t_parse, w->td[2] = o.t_group, w->td[3] = o.t_tok, w->td[4] = o.t_d2h;
                    w->dev = std::move(o);
                    // the wave's runs per cluster (they lie in slot order: one ends where the next present one starts), by line
                    w->runs.assign(nc, {});
                    const gtars::FragWaveOut &d = w->dev;
                    if (d.n) {
                        size_t pc = nc, pi = 0;
                        for (uint32_t f = 0; f + 1 < (uint32_t)d.slot_off.size(); ++f)
                            for (uint32_t g = d.slot_off[f]; g < d.slot_off[f + 1]; ++g) {
                                const uint32_t st0 = d.run_start[g];
                                if (st0 == 0xFFFFFFFFu) continue;
                                if (pc < nc) w->runs[pc][pi].len = st0 - w->runs[pc][pi].start;
                                const uint32_t c = w->tf[f].slots[g - d.slot_off[f]].value;
                                w->runs[c].push_back(Run{d.run_line[g], f, g - d.slot_off[f], st0, 0});
                                pc = c, pi = w->runs[c].size() - 1;
                            }
                        if (pc < nc) w->runs[pc][pi].len = (uint32_t)d.n_ids - w->runs[pc][pi].start;
                        for (auto &rc : w->runs) std::sort(rc.begin(), rc.end(), [](const Run &a, const Run &b) { return a.line < b.line; });
                    }
                }
                for (TextFile &f : w->tf) f.data.release();  // the text is on the device (or no longer needed): the block back to the pool
            } else {
                uint64_t h = 0;
                w->st = gtars_tokenize(t->index, w->c.get(), w->s.get(), w->e.get(), w->n, w->off.get(), &w->ids, &h);
                if (w->st) w->err = gtars_last_error();
            }
            } catch (const std::bad_alloc &) {
                w->st = GTARS_ERR_INTERNAL;
                w->err = "out of host memory";
            } catch (const std::exception &ex) {
                w->st = GTARS_ERR_INTERNAL;
                try {
                    w->err = std::string("internal error: ") + ex.what();
                } catch (...) {
                }
            }
            {
                std::lock_guard<std::mutex> lk(mu);
                t_tok += now() - t0;
                --in_flight;
                if (w->st) wave_failed = true;
            }
            cv.notify_all();
        }
    };
    // The device path runs TWO of these threads, each with a stream of its own (fragparse.hip): a batch's text goes to the device
    // while the previous batch's kernels run and its results come back -- on one thread a batch is copy in, kernels, copy out, one
    // after the other, 10.5 ms of device time per 48 files of which 5 are the text's way over PCIe.  (They sleep in their waits.)
    const unsigned n_tok_threads = device_path ? (unsigned)std::max(1, std::min(4, atoi(cfg_get("GTARS_FRAG_DEVICE_THREADS") ? cfg_get("GTARS_FRAG_DEVICE_THREADS") : "2"))) : 1u;
    std::vector<std::thread> tok_threads;
    for (unsigned q = 0; q < n_tok_threads; ++q) tok_threads.emplace_back(tok_body);
    auto finish_tokenizer = [&] {
        {
            std::lock_guard<std::mutex> lk(mu);
            closing = true;
        }
        cv.notify_all();
        for (auto &th : tok_threads)
            if (th.joinable()) th.join();
    };
    struct AtExit {  // (an exception on the way -- out of memory -- must not meet a joinable thread)
        std::function<void()> f;
        ~AtExit() { f(); }
    } join_at_exit{finish_tokenizer};
    auto free_waves = [&] {
        for (Wave &w : waves)
            if (!w.device) gtars_free(w.ids);  // (a device wave's ids lie in a block of the pinned pool: Wave::dids)
    };
    auto host_sink = [&](size_t, std::vector<SplitFile> &res) -> gtars_status {
        const double t_a = now();
        waves.emplace_back();
        Wave &w = waves.back();
        w.coff.assign(nc + 1, 0);
        for (size_t c = 0; c < nc; ++c) {
            uint64_t add = 0;
            for (SplitFile &f : res) add += f.cols[c].c.size();
            w.coff[c + 1] = w.coff[c] + add;
        }
        const uint64_t n = w.n = w.coff[nc];
        w.c.reset(new uint32_t[n + 1]);
        w.s.reset(new uint32_t[n + 1]);
        w.e.reset(new uint32_t[n + 1]);
        w.off.reset(new uint64_t[n + 1]);
        w.off[0] = 0;  // (a wave without routed fragments is not sent to the tokenizer)
        // the wave's files, in file order, behind every cluster: column copies; the file-local barcode ids mapped through the
        // cluster's dictionary (one lookup per distinct barcode of a file, in its first-seen order, so the cluster's order is
        // what one pass over the cluster file would see)
        over_clusters([&](size_t c) {
            Cluster &k = cl[c];
            k.b.reserve(k.b.size() + (size_t)(w.coff[c + 1] - w.coff[c]));
            uint64_t at = w.coff[c];
            std::vector<uint32_t> map;
            for (SplitFile &f : res) {
                SplitFile::Cols &q = f.cols[c];
                map.resize(q.barcodes.names.size());
                for (size_t i = 0; i < map.size(); ++i) map[i] = k.barcodes.get_or_add(q.barcodes.names[i].data(), q.barcodes.names[i].size());
                std::copy(q.c.begin(), q.c.end(), w.c.get() + at);
                std::copy(q.s.begin(), q.s.end(), w.s.get() + at);
                std::copy(q.e.begin(), q.e.end(), w.e.get() + at);
                at += q.c.size();
                for (uint32_t lb : q.b) k.b.push_back(map[lb]);
            }
        });
        for (SplitFile &f : res) reads += f.n_reads;
        n_all += n;
        if (n) {
            std::lock_guard<std::mutex> lk(mu);
            jobs.push_back(&w);
            ++in_flight;
        }
        cv.notify_all();
        t_append += now() - t_a;
        return GTARS_OK;
    };
    if (!device_path) {
        st = for_each_split_wave(files, *m, false, &t->chroms, host_sink);
    } else {
        const char *cap_mb = cfg_get("GTARS_FRAG_DEVICE_WAVE_MB");  // (test hook: a small limit)
        const uint64_t byte_limit = (cap_mb ? (uint64_t)atoll(cap_mb) : 3500ull) << 20;  // the device path's 32-bit text positions
        bool too_large = false;
        st = stream_text_files(
            files, *m, byte_limit, pipeline_device,
            [&](size_t base, std::vector<TextFile> &tf) -> gtars_status {
                waves.emplace_back();
                Wave &w = waves.back();
                w.device = true;
                w.first_file = base;
                w.tf = std::move(tf);
                w.coff.assign(nc + 1, 0);
                {
                    std::lock_guard<std::mutex> lk(mu);
                    jobs.push_back(&w);
                    ++in_flight;
                }
                cv.notify_all();
                return GTARS_OK;
            },
            [&] {  // a device thread has nothing queued and nothing running
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return in_flight < n_tok_threads; });
            },
            [&] {
                std::lock_guard<std::mutex> lk(mu);
                return wave_failed;
            },
            &too_large);
        if (too_large) *redo_on_host = true;
    }
    const double t_split_done = now();
    finish_tokenizer();
    if (device_overflow) {
        free_waves();
        *redo_on_host = true;
        return fail(GTARS_ERR_INTERNAL, "fragment pipeline: a device wave with more than 4e9 token ids");
    }
    for (Wave &w : waves)  // (in file order: a wave's error lies in front of whatever stopped the producer)
        if (w.st) {
            const gtars_status e = w.st;
            const std::string msg = w.err;
            free_waves();
            return fail(e, msg);
        }
    if (st) {
        const std::string msg = gtars_last_error();
        free_waves();
        return fail(st, msg);
    }
    const double t_tok_done = now();
    if (device_path)
        for (Wave &w : waves)
            if (w.device) {
                reads += w.reads;
                n_all += w.n;
            }
    // every cluster regrouped by barcode: its fragments' ids wave after wave (= the cluster file's line order)
    auto **arr = (gtars_fragment_tokens_t **)calloc(nc ? nc : 1, sizeof(gtars_fragment_tokens_t *));
    // Device waves arrive regrouped by (file, barcode) (FragWaveOut): what is left is, per cluster, the runs in the order their first
    // lines appear -- wave after wave, and inside a wave by line number: the first-seen barcode order of one pass over the cluster's
    // file -- a dictionary lookup per run (the same barcode string of several files is ONE barcode of the cluster) and a copy per run.
    std::vector<Wave *> wave_at;
    for (Wave &w : waves) wave_at.push_back(&w);
    over_clusters([&](size_t c) {
        Cluster &k = cl[c];
        std::vector<uint64_t> cnt;
        uint64_t i = 0;
        std::vector<uint32_t> run_id;
        if (device_path) {
            // the cluster's runs wave after wave, inside a wave by line: the first-seen barcode order of one pass over its file
            size_t n_runs = 0;
            for (Wave *w : wave_at)
                if (w->device && c < w->runs.size()) n_runs += w->runs[c].size();
            run_id.reserve(n_runs);
            std::vector<uint64_t> per;
            for (Wave *w : wave_at) {
                if (!w->device || c >= w->runs.size()) continue;
                for (const Run &r : w->runs[c]) {
                    const TextFile &f = w->tf[r.file];
                    const gtars::FragSlot &sl = f.slots[r.slot];
                    const uint32_t id = k.barcodes.get_or_add(f.keys.data() + sl.off, sl.len);
                    if (id >= per.size()) per.resize((size_t)id + 1, 0);
                    per[id] += r.len;
                    run_id.push_back(id);
                }
            }
            cnt.assign(per.size() + 1, 0);
            for (size_t b = 0; b < per.size(); ++b) cnt[b + 1] = per[b];
        }
        const uint64_t nb = k.barcodes.names.size();
        if (!device_path) {
            cnt.assign(nb + 1, 0);
            for (Wave &w : waves)
                for (uint64_t r = w.coff[c]; r < w.coff[c + 1]; ++r, ++i) {
                    const uint64_t hits = w.off[r + 1] - w.off[r];
                    cnt[k.b[i] + 1] += hits ? hits : 1;  // a fragment without hits contributes one unk id
                }
        }
        for (uint64_t b = 0; b < nb; ++b) cnt[b + 1] += cnt[b];
        auto *ft = (gtars_fragment_tokens_t *)calloc(1, sizeof(gtars_fragment_tokens_t));
        ft->n_barcodes = nb;
        ft->barcodes = (char **)calloc(nb ? nb : 1, sizeof(char *));
        ft->offsets = (uint64_t *)malloc((nb + 1) * sizeof(uint64_t));
        ft->ids = (uint32_t *)malloc((cnt[nb] ? cnt[nb] : 1) * sizeof(uint32_t));
        memcpy(ft->offsets, cnt.data(), (nb + 1) * sizeof(uint64_t));
        for (uint64_t b = 0; b < nb; ++b) ft->barcodes[b] = dup_cstr(k.barcodes.names[b]);
        std::vector<uint64_t> fill(cnt.begin(), cnt.end() - 1);
        if (device_path) {
            // (a barcode's runs in (wave, line) order = file order: the fragments of a cluster lie file after file)
            size_t x = 0;
            for (Wave *w : wave_at) {
                if (!w->device || c >= w->runs.size()) continue;
                const uint32_t *src = w->dev.ids.get();
                for (const Run &r : w->runs[c]) {
                    uint64_t &at = fill[run_id[x++]];
                    memcpy(ft->ids + at, src + r.start, (size_t)r.len * sizeof(uint32_t));
                    at += r.len;
                }
            }
            arr[c] = ft;
            return;
        }
        i = 0;
        for (Wave &w : waves) {
            for (uint64_t r = w.coff[c]; r < w.coff[c + 1]; ++r, ++i) {
                uint64_t &at = fill[k.b[i]];
                if (w.off[r + 1] == w.off[r]) {
                    ft->ids[at++] = t->unk_id;
                } else {
                    for (uint64_t y = w.off[r]; y < w.off[r + 1]; ++y) ft->ids[at++] = w.ids[y];
                }
            }
        }
        arr[c] = ft;
    });
    free_waves();
    *out = arr;
    if (n_reads) *n_reads = reads;
    {
        // the stages of this call, for gtars_fragsplit_last_stages (bench.py's stage report)
        double *g = g_frag_stages;
        for (int k = 0; k < 12; ++k) g[k] = 0;
        g[0] = device_path ? 1 : 0;
        g[1] = (double)waves.size();
        g[2] = t_split_done - t_begin - t_append;  // read + inflate (host path: + parse + route)
        g[3] = t_append;
        g[4] = t_tok;
        g[5] = t_tok_done - t_split_done;  // ... of which behind the last wave's files
        g[6] = now() - t_tok_done;         // regroup by barcode
        for (Wave &w : waves)
            for (int k = 0; k < 5; ++k) g[7 + k] += w.td[k];
    }
    if (timing) {
        fprintf(stderr, "[gtars host timing] fragsplit_tokenize: %zu files in %zu wave(s), %s %.3f s, per-cluster append %.3f s, %s %.3f s (of which %.3f s after the last wave was read; %llu fragments), regroup of %zu clusters %.3f s\n",
                files.size(), waves.size(), device_path ? "gunzip (host threads)" : "gunzip + parse + route", t_split_done - t_begin - t_append,
                t_append, device_path ? "device waves (text in, parse, group, tokenize, results out)" : "tokenizer calls", t_tok,
                t_tok_done - t_split_done, (unsigned long long)n_all, nc, now() - t_tok_done);
        fprintf(stderr, "[gtars host timing]   since the call was entered: %.3f s (set-up before the first wave %.3f s)\n", now() - t_enter, t_begin - t_enter);
        if (device_path) {
            double td[5] = {0, 0, 0, 0, 0};
            for (Wave &w : waves)
                for (int k = 0; k < 5; ++k) td[k] += w.td[k];
            fprintf(stderr, "[gtars host timing]   device waves: text to the device %.3f s, line split + parse + sort by cluster %.3f s, gather %.3f s, tokenize %.3f s, results to the host %.3f s\n",
                    td[0], td[1], td[2], td[3], td[4]);
        }
    }
    return GTARS_OK;
}

static gtars_status gtars_fragsplit_tokenize_impl(const gtars_tokenizer_t *t, const char *files_dir, const gtars_barcode_map_t *m,
                                      gtars_fragment_tokens_t ***out, uint64_t *n_reads) {
    if (!t || !files_dir || !m || !out) return fail(GTARS_ERR_INVALID_ARG, "NULL argument");
    *out = nullptr;
    std::vector<std::string> files;
    gtars_status st = list_regular_files(files_dir, files);
    if (st) return st;
    return fragsplit_tokenize_core(t, files, m, out, n_reads);
}

static gtars_status gtars_fragsplit_tokenize_files_impl(const gtars_tokenizer_t *t, const char *const *paths, uint64_t n_paths,
                                                        const gtars_barcode_map_t *m, gtars_fragment_tokens_t ***out, uint64_t *n_reads) {
    if (!t || !m || !out || (n_paths && !paths)) return fail(GTARS_ERR_INVALID_ARG, "NULL argument");
    *out = nullptr;
    std::vector<std::string> files;
    for (uint64_t i = 0; i < n_paths; ++i) {
        if (!paths[i]) return fail(GTARS_ERR_INVALID_ARG, "NULL path");
        files.emplace_back(paths[i]);
    }
    return fragsplit_tokenize_core(t, files, m, out, n_reads);
}

}  // extern "C"

// ---- the C ABI never lets a C++ exception cross it (std::bad_alloc of a huge build, std::length_error ...)
extern "C" {

gtars_status gtars_fragsplit_tokenize_files(const gtars_tokenizer_t *t, const char *const *paths, uint64_t n_paths,
                                            const gtars_barcode_map_t *m, gtars_fragment_tokens_t ***out, uint64_t *n_reads) {
    return gtars::guarded([&]() -> gtars_status { return gtars_fragsplit_tokenize_files_impl(t, paths, n_paths, m, out, n_reads); });
}

gtars_status gtars_regionset_from_bed(const char *path, gtars_regionset_t **out) {
    return gtars::guarded([&]() -> gtars_status { return gtars_regionset_from_bed_impl(path, out); });
}

gtars_status gtars_igddb_from_bed_files(const char *const *paths, uint64_t n_paths, gtars_igddb_t **out) {
    return gtars::guarded([&]() -> gtars_status { return gtars_igddb_from_bed_files_impl(paths, n_paths, out); });
}

gtars_status gtars_fragsplit(const char *files_dir, const gtars_barcode_map_t *m, const char *out_dir, uint64_t *n_reads,
                             uint64_t *n_written) {
    return gtars::guarded([&]() -> gtars_status { return gtars_fragsplit_impl(files_dir, m, out_dir, n_reads, n_written); });
}

gtars_status gtars_fragsplit_tokenize(const gtars_tokenizer_t *t, const char *files_dir, const gtars_barcode_map_t *m,
                                      gtars_fragment_tokens_t ***out, uint64_t *n_reads) {
    return gtars::guarded([&]() -> gtars_status { return gtars_fragsplit_tokenize_impl(t, files_dir, m, out, n_reads); });
}

gtars_status gtars_igddb_load(const char *path, gtars_igddb_t **out, int32_t *nbp_out) {
    return gtars::guarded([&]() -> gtars_status { return gtars_igddb_load_impl(path, out, nbp_out); });
}

gtars_status gtars_igddb_save(const gtars_igddb_t *db, const char *path, int32_t nbp) {
    return gtars::guarded([&]() -> gtars_status { return gtars_igddb_save_impl(db, path, nbp); });
}

gtars_status gtars_igddb_from_arrays(const char *const *chrom_names, uint32_t n_chrom, const uint32_t *chrom,
                                     const int32_t *start, const int32_t *end, const int32_t *value,
                                     const uint32_t *file_idx, uint64_t n, const char *const *file_names,
                                     const uint32_t *num_regions, const double *avg_width, uint32_t n_files,
                                     gtars_igddb_t **out) {
    return gtars::guarded([&]() -> gtars_status { return gtars_igddb_from_arrays_impl(chrom_names, n_chrom, chrom, start, end, value, file_idx, n, file_names, num_regions, avg_width, n_files, out); });
}

gtars_status gtars_tokenizer_tokenize_fragment_file(const gtars_tokenizer_t *t, const char *path,
                                                    gtars_fragment_tokens_t **out) {
    return gtars::guarded([&]() -> gtars_status { return gtars_tokenizer_tokenize_fragment_file_impl(t, path, out); });
}

gtars_status gtars_fragments_read(const char *path, gtars_fragments_t **out) {
    return gtars::guarded([&]() -> gtars_status { return gtars_fragments_read_impl(path, out); });
}

gtars_status gtars_bed3_lines_read(const char *path, gtars_fragments_t **out) {
    return gtars::guarded([&]() -> gtars_status {
        if (!path || !out) return fail(GTARS_ERR_INVALID_ARG, "NULL argument");
        *out = nullptr;
        auto f = std::make_unique<gtars_fragments>();
        gtars_status st = read_bed3_lines(path, f->t);
        if (st) return st;
        for (const std::string &n : f->t.chroms) f->chrom_ptrs.push_back(n.c_str());
        *out = f.release();
        return GTARS_OK;
    });
}

// one line  chr<TAB>start<TAB>end  per hit (handlers.rs:141-150), `n` hits, into a malloc'ed buffer (gtars_free)
gtars_status gtars_format_hit_lines(const char *const *chrom_names, const uint32_t *hit_chrom, const uint32_t *hit_start,
                                    const uint32_t *hit_end, uint64_t n, char **out_text, uint64_t *out_len) {
    return gtars::guarded([&]() -> gtars_status {
        if (!out_text || !out_len || (n && (!chrom_names || !hit_chrom || !hit_start || !hit_end)))
            return fail(GTARS_ERR_INVALID_ARG, "NULL argument");
        std::string text;
        text.reserve((size_t)n * 24);
        char buf[16];
        auto put_u32 = [&](uint32_t v) {
            int k = 0;
            do { buf[k++] = (char)('0' + v % 10); v /= 10; } while (v);
            while (k) text.push_back(buf[--k]);
        };
        for (uint64_t i = 0; i < n; ++i) {
            text += chrom_names[hit_chrom[i]];
            text.push_back('\t');
            put_u32(hit_start[i]);
            text.push_back('\t');
            put_u32(hit_end[i]);
            text.push_back('\n');
        }
        char *p = (char *)malloc(text.size() + 1);
        if (!p) return fail(GTARS_ERR_INTERNAL, "out of host memory");
        memcpy(p, text.data(), text.size());
        p[text.size()] = 0;
        *out_text = p;
        *out_len = text.size();
        return GTARS_OK;
    });
}

}  // extern "C"
