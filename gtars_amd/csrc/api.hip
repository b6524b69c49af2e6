// api.hip -- C ABI of the integer-level engine: index / IGD handles, host and
// device entry points (declared in include/gtars_amd.h).
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <condition_variable>
#include <deque>
#include <functional>
#include <numeric>
#include <thread>
#include <tuple>

#include <memory>
#include <unordered_map>

#include "common.h"
#include "../../include/gtars_amd_debug.h"

namespace gtars {

// ------------------------------------------------------------------- errors
static thread_local std::string g_last_error;

void set_error(const std::string &msg) { g_last_error = msg; }

// ---- environment snapshot (common.h: cfg_get) ----------------------------------
extern "C" char **environ;
namespace {
using EnvMap = std::unordered_map<std::string, std::string>;
std::mutex g_env_mu;
std::shared_ptr<const EnvMap> g_env;  // never freed while a reader may hold a value: old snapshots are kept (g_env_old)
std::vector<std::shared_ptr<const EnvMap>> g_env_old;
std::shared_ptr<const EnvMap> take_env_snapshot() {
    auto m = std::make_shared<EnvMap>();
    for (char **e = environ; e && *e; ++e) {
        if (strncmp(*e, "GTARS_", 6) != 0) continue;
        const char *eq = strchr(*e, '=');
        if (!eq) continue;
        (*m)[std::string(*e, (size_t)(eq - *e))] = std::string(eq + 1);
    }
    return m;
}
}  // namespace
const char *cfg_get(const char *name) {
    std::shared_ptr<const EnvMap> m;
    {
        std::lock_guard<std::mutex> lk(g_env_mu);
        if (!g_env) g_env = take_env_snapshot();
        m = g_env;
    }
    auto it = m->find(name);
    return it == m->end() ? nullptr : it->second.c_str();  // (the snapshot outlives the call: see g_env_old)
}
long cfg_int(const char *name, long dflt) {
    const char *v = cfg_get(name);
    return v && *v ? atol(v) : dflt;
}

gtars_status fail(gtars_status st, const std::string &msg) {
    g_last_error = msg;
    return st;
}

gtars_status hip_fail(hipError_t e, const char *what, const char *file, int line) {
    char buf[512];
    snprintf(buf, sizeof buf, "HIP error %d (%s) at %s:%d: %s", (int)e, hipGetErrorString(e), file, line, what);
    g_last_error = buf;
    (void)hipGetLastError();
    return GTARS_ERR_HIP;
}

gtars_status require_device() {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return fail(GTARS_ERR_NO_DEVICE,
                    "no HIP device available: libgtars_amd has no CPU fallback (the oracle under oracle/ is "
                    "test infrastructure, not a product path)");
    }
    return GTARS_OK;
}

// ---------------------------------------------------------------- workspace
gtars_status Workspace::reserve(size_t need) {
    int dev = 0;
    GT_HIP(hipGetDevice(&dev));
    if (ptr && device == dev && bytes >= need) return GTARS_OK;
    if (ptr) {
        (void)hipDeviceSynchronize();  // a launch may still be using the old buffer
        (void)hipFree(ptr);
        ptr = nullptr;
        bytes = 0;
    }
    ep = ScanEpoch();
    size_t want = std::max<size_t>(need, 1 << 20);
    want = (want + (1 << 20) - 1) & ~(size_t)((1 << 20) - 1);
    GT_HIP(hipMalloc(&ptr, want));
    bytes = want;
    device = dev;
    // the head of the buffer is state that outlives a call (launch_igd_sweep's order flags): zero when the buffer is new
    igd_calls = 0;
    GT_HIP(hipMemset(ptr, 0, 64));
    GT_HIP(hipDeviceSynchronize());
    return GTARS_OK;
}

Workspace::~Workspace() {
    // Process teardown order vs. the HIP runtime is not defined; leaking at
    // exit is harmless, freeing after runtime shutdown is not.
}

// One grow-only workspace per (host thread, device, stream, purpose): calls issued by one thread on DIFFERENT
// streams may overlap on the device, so they must not share scratch memory; calls on the same stream are
// ordered by the stream.  The workspaces of a thread that exits go back to a process-wide pool (device memory
// is never returned to the runtime from a thread-exit handler), from which new threads take theirs: a program
// that keeps creating short-lived threads holds as many workspaces as it has threads alive, not as it ever had.
namespace {
struct WorkspacePool {
    std::mutex mu;
    std::vector<Workspace> idle;
};
WorkspacePool &workspace_pool() {
    static WorkspacePool *p = new WorkspacePool();  // never destroyed: thread-exit handlers may run late
    return *p;
}
struct TlsWorkspaces {
    std::map<std::tuple<int, hipStream_t, int>, Workspace> ws;
    ~TlsWorkspaces() {
        WorkspacePool &pool = workspace_pool();
        std::lock_guard<std::mutex> g(pool.mu);
        for (auto &kv : ws)
            if (kv.second.ptr) {
                pool.idle.push_back(kv.second);
                kv.second.ptr = nullptr;
            }
    }
};
}  // namespace

Workspace &tls_workspace(int slot, hipStream_t stream) {
    // the default stream (NULL) exists on every device: the current device is part of the key
    static thread_local TlsWorkspaces tls;
    int dev = 0;
    (void)hipGetDevice(&dev);
    const auto key = std::make_tuple(dev, stream, slot);
    auto it = tls.ws.find(key);
    if (it != tls.ws.end()) return it->second;
    Workspace w;
    {
        WorkspacePool &pool = workspace_pool();
        std::lock_guard<std::mutex> g(pool.mu);
        for (size_t i = 0; i < pool.idle.size(); ++i)
            if (pool.idle[i].device == dev) {
                w = pool.idle[i];
                w.ep = ScanEpoch();  // whoever used it last may have left anything in it
                // ... also in the 64 bytes at its head that outlive a call (launch_igd_sweep's order flags: "zero when the buffer
                // is new" must hold for an adopted buffer as well)
                w.igd_calls = 0;
                // (on the stream this workspace will serve: a memset on the null stream is not ordered in front of work on a
                // non-blocking stream, and may land in the middle of the first kernel that writes there)
                if (w.ptr) (void)hipMemsetAsync(w.ptr, 0, 64, stream);
                pool.idle[i] = pool.idle.back();
                pool.idle.pop_back();
                break;
            }
    }
    return tls.ws.emplace(key, w).first->second;
}

// ---------------------------------------------------------------- profiling
struct ProfEntry {
    std::string name;
    double total_ms = 0;
    u64 launches = 0;
};
struct PendingEvent {
    int entry;
    hipEvent_t e0, e1;
};
static thread_local bool g_prof_on = false;
static thread_local std::vector<ProfEntry> g_prof_entries;
static thread_local std::vector<PendingEvent> g_prof_pending;

static int prof_entry(const char *name) {
    for (size_t i = 0; i < g_prof_entries.size(); ++i)
        if (g_prof_entries[i].name == name) return (int)i;
    g_prof_entries.push_back(ProfEntry{name, 0, 0});
    return (int)g_prof_entries.size() - 1;
}

ProfScope::ProfScope(const char *n, hipStream_t s) : name(n), st(s) {
    on = g_prof_on;
    if (!on) return;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
        on = false;
        return;
    }
    (void)hipEventRecord(e0, st);
}

ProfScope::~ProfScope() {
    if (!on) return;
    (void)hipEventRecord(e1, st);
    g_prof_pending.push_back(PendingEvent{prof_entry(name), e0, e1});
}

static void prof_drain() {
    for (auto &p : g_prof_pending) {
        float ms = 0;
        if (hipEventSynchronize(p.e1) == hipSuccess && hipEventElapsedTime(&ms, p.e0, p.e1) == hipSuccess) {
            g_prof_entries[p.entry].total_ms += ms;
            g_prof_entries[p.entry].launches += 1;
        }
        (void)hipEventDestroy(p.e0);
        (void)hipEventDestroy(p.e1);
    }
    g_prof_pending.clear();
}

// Deterministic observation of a choice a kernel made on the device (tests assert on this, never on kernel times):
// with profiling enabled, reads the device word behind the work queued so far and counts one "launch" of the entry
// named after its state.  Synchronises the stream -- profiling mode only.
void prof_note_device_flag(const char *if_set, const char *if_clear, const u32 *d_flag, hipStream_t st) {
    if (!g_prof_on) return;
    u32 h = 0;
    if (hipMemcpyAsync(&h, d_flag, sizeof h, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
        (void)hipGetLastError();
        return;
    }
    g_prof_entries[prof_entry(h ? if_set : if_clear)].launches += 1;
}

// a choice the host made (profiling mode: counts one "launch" of the entry; a fact for the tests, not a time)
void prof_note_fact(const char *name) {
    if (g_prof_on) g_prof_entries[prof_entry(name)].launches += 1;
}

// -------------------------------------------------------------- device bufs
template <class T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    gtars_status upload(const std::vector<T> &h) {
        n = h.size();
        GT_HIP(hipMalloc((void **)&p, std::max<size_t>(n, 1) * sizeof(T) + 32));  // + slack: the IGD sweep stages whole 16-byte vectors
        if (n) GT_HIP(hipMemcpy(p, h.data(), n * sizeof(T), hipMemcpyHostToDevice));
        return GTARS_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
};

struct ScopedDev {
    void *p = nullptr;
    ~ScopedDev() {
        if (p) (void)hipFree(p);
    }
    gtars_status alloc(size_t bytes) {
        GT_HIP(hipMalloc(&p, std::max<size_t>(bytes, 16)));
        return GTARS_OK;
    }
    template <class T>
    T *as() {
        return (T *)p;
    }
};

// the calling thread's current device for the scope: `want`, and what it was before again afterwards
struct DeviceScope {
    int before = -1;
    bool switched = false;
    gtars_status st = GTARS_OK;
    explicit DeviceScope(int want) {
        if (want < 0) return;
        hipError_t e = hipGetDevice(&before);
        if (e == hipSuccess && before != want) {
            e = hipSetDevice(want);
            switched = e == hipSuccess;
        }
        if (e != hipSuccess) st = hip_fail(e, "select the handle's device", __FILE__, __LINE__);
    }
    ~DeviceScope() {
        if (switched) (void)hipSetDevice(before);
    }
};

// Device affinity of a handle (round 6: every entry point that takes one).  A handle's device memory -- and everything built
// lazily on it later -- lives on the device that was current when it was built.
//  * host-buffer entry points run ON that device whatever the calling thread's current device is, and put the caller's device
//    back afterwards:   GT_ON_DEVICE_OF(h);
//  * `*_device` entry points take the CALLER's device pointers and stream, which belong to the caller's current device: a handle
//    that lives elsewhere is an argument error, not a memory fault:   GT_SAME_DEVICE_AS(h);
static gtars_status require_handle_device(int handle_device) {
    int cur = -1;
    hipError_t e = hipGetDevice(&cur);
    if (e != hipSuccess) return hip_fail(e, "query the current device", __FILE__, __LINE__);
    if (cur != handle_device)
        return fail(GTARS_ERR_INVALID_ARG, "handle lives on device " + std::to_string(handle_device) + ", current device is " +
                                               std::to_string(cur) + ": device pointers and stream must belong to the handle's device");
    return GTARS_OK;
}
#define GT_ON_DEVICE_OF(h)                                \
    DeviceScope gt_device_scope_((h) ? (h)->device : -1); \
    if (gt_device_scope_.st) return gt_device_scope_.st
#define GT_SAME_DEVICE_AS(h)                                                 \
    do {                                                                     \
        if (h) {                                                             \
            const gtars_status gt_dev_st_ = require_handle_device((h)->device); \
            if (gt_dev_st_) return gt_dev_st_;                               \
        }                                                                    \
    } while (0)

}  // namespace gtars

using namespace gtars;

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

// K1 policy: large builds are ordered by the device radix sort, small ones by std::stable_sort
// (identical results; GTARS_DEVICE_SORT=0/1 forces one path, used by the tests)
static bool use_device_sort(u64 n) {
    if (const char *e = cfg_get("GTARS_DEVICE_SORT")) return atoi(e) != 0;
    return n >= (1u << 16);
}

// perm[p] = input row of sorted position p, ordered by (chrom, k1, [k2], input order) on the device
static gtars_status sorted_perm_device(const std::vector<u32> &chrom, const std::vector<u32> &k1,
                                       const std::vector<u32> *k2, u32 n_chrom, std::vector<u32> &perm) {
    const u32 n = (u32)chrom.size();
    perm.resize(n);
    if (!n) return GTARS_OK;
    ScopedDev buf;
    gtars_status st = buf.alloc((size_t)n * 4 * 4);
    if (st) return st;
    u32 *dc = buf.as<u32>(), *d1 = dc + n, *d2 = d1 + n, *dp = d2 + n;
    GT_HIP(hipMemcpy(dc, chrom.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    GT_HIP(hipMemcpy(d1, k1.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    if (k2) GT_HIP(hipMemcpy(d2, k2->data(), (size_t)n * 4, hipMemcpyHostToDevice));
    st = device_sort_perm(dc, d1, k2 ? d2 : nullptr, n, n_chrom, dp, nullptr);
    if (st) return st;
    GT_HIP(hipMemcpy(perm.data(), dp, (size_t)n * 4, hipMemcpyDeviceToHost));
    return GTARS_OK;
}

// ================================================================== handles

struct gtars_index {
    int kind = 0;
    u32 n_chrom = 0;
    u64 n = 0;
    int device = 0;
    // host mirrors (small: metadata only) + full host copy of the stored order
    std::vector<u32> h_starts, h_ends, h_vals, h_max_ends;
    std::vector<u32> h_chrom_off, h_chrom_aux, h_chrom_sub, h_sub_off;
    DevBuf<u32> starts, ends, vals, max_ends, chrom_off, chrom_aux, chrom_sub, sub_off;
    // blocked acceleration structure (Bits kind), see AccelView in common.h
    DevBuf<u32> acc_rec2, acc_rec4, acc_rec8, acc_blk_first, acc_lut, acc_qkeys, acc_chrom_tab, acc_idc, acc_idc_pos, acc_chrom_iv_end;
    bool acc_ids_affine = false, acc_ends_mono = false, acc_runs_ok = false;
    u32 acc_n_blocks = 0, acc_n_units = 0, acc_n_buckets = 0, acc_lut_words = 0, acc_q_words = 0;
    u32 acc_lut_shift = 0, acc_q_shift = 0, acc_search_top = 0, acc_top_shift = 0;
    bool has_accel = false;
    // stored position -> input row (both kinds)
    std::vector<u32> h_rows;
    // FLAT COMPANION of an AIList-kind index with nested sub-lists (null otherwise): the same intervals and values as a Bits-kind
    // index with its blocked structure.  The reference's default IndexedRegionSet index IS AIList (indexed_region_set.rs:111-113),
    // and everything it answers through it -- count / any / find_overlaps (sorted unique source rows) / subset_by_overlaps /
    // intersect_all -- does not depend on the enumeration order, so those calls run on the companion's LDS kernels; only
    // enumeration in AIList::find order (ailist.rs:153-178, 238-263) stays on this index.  flat_pos[p'] = this index's stored
    // position of the companion's stored position p' (for the device bitmap of gtars_mark_overlapped_device).
    gtars_index *flat = nullptr;
    DevBuf<u32> flat_pos;
    // ... and, round 5, enumeration in AIList::find order as well: the companion's LDS tokenizer writes the hits as its own stored
    // positions, and k_ailist_reorder sorts every query's hits by ail_key[position] -- the hit's mirrored position in THIS index's
    // stored order (sub-list start + sub-list length - 1 - index inside the sub-list: ascending = sub-lists in order, each from
    // its last interval down, ailist.rs:153-178, 238-263) -- and replaces them by ail_val[key]
    DevBuf<u32> ail_key, ail_val;
    // ... unless the universe is so deep (mean number of intervals over a covered position) that most queries have long tails:
    // there the LDS tokenizer walks and the one-thread-per-query kernel wins (tools/ailist_bench.py: depth 0.5 / 4.6 / 7.7 ->
    // 0.43 vs 1.52, 2.30 vs 3.07, 3.34 vs 2.71 ms per 16M queries); GTARS_AILIST_REORDER_MAX_DEPTH moves the line (default 6)
    double ail_depth = 0;
    // per-chromosome sorted copy of the ends, built on the first Bits::count call (bits.rs:118-121)
    mutable std::mutex ends_mu;
    mutable DevBuf<u32> ends_sorted;
    mutable bool ends_ready = false;
    AccelView accel() const {
        AccelView a;
        a.rec2 = reinterpret_cast<const uint4 *>(acc_rec2.p);
        a.rec4 = reinterpret_cast<const uint4 *>(acc_rec4.p);
        a.rec8 = cfg_flag("GTARS_TOK_NO_UNIT_RECORDS") ? nullptr : reinterpret_cast<const uint4 *>(acc_rec8.p);  // (the switch: A/B, tests)
        a.idc = acc_idc.p;
        a.ids_affine = acc_ids_affine ? 1u : 0u;
        a.blk_first = acc_blk_first.p;
        a.lut = acc_lut.p;
        a.qkeys = acc_qkeys.p;
        a.chrom_tab = reinterpret_cast<const uint4 *>(acc_chrom_tab.p);
        a.n_blocks = acc_n_blocks;
        a.n_units = acc_n_units;
        a.n_buckets = acc_n_buckets;
        a.lut_words = acc_lut_words;
        a.q_words = acc_q_words;
        a.lut_shift = acc_lut_shift;
        a.q_shift = acc_q_shift;
        a.search_top = acc_search_top;
        a.top_shift = acc_top_shift;
        a.n_chrom = n_chrom;
        a.max_chrom_n = max_chrom_n();
        a.runs_ok = acc_runs_ok && !cfg_flag("GTARS_TOK_NO_RUNS") ? 1u : 0u;  // (the switch: tests and A/B runs of the tail walk)
        a.ends_mono = a.runs_ok && acc_ends_mono ? 1u : 0u;
        a.chrom_iv_end = acc_chrom_iv_end.p;
        return a;
    }
    // The same structure with ids that are the STORED POSITIONS of the hits (position of (block b, slot k) =
    // ACC_OWN * b + k + idc_pos[chrom]): what the region-returning calls gather starts / ends / values by.
    AccelView accel_pos() const {
        AccelView a = accel();
        a.idc = acc_idc_pos.p;
        a.ids_affine = 1u;
        a.rec4 = nullptr;
        return a;
    }
    // most intervals on one chromosome = the most hits one query can have
    u32 max_chrom_n() const {
        u32 m = 0;
        for (size_t c = 0; c + 1 < h_chrom_off.size(); ++c) m = std::max(m, h_chrom_off[c + 1] - h_chrom_off[c]);
        return m;
    }
    IndexView view() const {
        IndexView v;
        v.starts = starts.p;
        v.ends = ends.p;
        v.vals = vals.p;
        v.max_ends = max_ends.p;
        v.chrom_off = chrom_off.p;
        v.chrom_aux = chrom_aux.p;
        v.chrom_sub = chrom_sub.p;
        v.sub_off = sub_off.p;
        v.n_chrom = n_chrom;
        v.n = (u32)n;
        return v;
    }
};

struct gtars_igd {
    u32 n_chrom = 0, n_files = 0;
    u64 n = 0;
    int device = 0;  // the device the database was built on (and everything built lazily later lives on)
    // Databases with records longer than the piece length keep a second index of PIECES for the min_overlap == 1 counts (see
    // build_pieces_view below); null otherwise.  Owned.
    gtars_igd *pieces = nullptr;
    // host copy of the stored starts / ends: filled at build by the host-sort path, on first use
    // (total_records / export) after a device build
    mutable std::vector<i32> h_starts, h_ends;
    mutable std::mutex mirror_mu;
    mutable bool mirror_ready = false;
    gtars_status ensure_mirror() const {
        std::lock_guard<std::mutex> lk(mirror_mu);
        if (mirror_ready) return GTARS_OK;
        h_starts.resize(n);
        h_ends.resize(n);
        if (n) {
            GT_HIP(hipMemcpy(h_starts.data(), starts.p, n * sizeof(i32), hipMemcpyDeviceToHost));
            GT_HIP(hipMemcpy(h_ends.data(), ends.p, n * sizeof(i32), hipMemcpyDeviceToHost));
        }
        mirror_ready = true;
        return GTARS_OK;
    }
    DevBuf<i32> starts, ends, values, chrom_maxlen;
    DevBuf<u32> files, chrom_off;
    // tiles of IGD_TILE_RECORDS consecutive records of one chromosome (batch sweep, igd_sweep.hip)
    DevBuf<u32> tile_first, tile_cnt, tile_chrom;
    // per tile: the largest end among the chromosome's records BEFORE the tile (0: none) -- the carry-in of the
    // prefix maximum the sweep builds in LDS
    DevBuf<i32> tile_carry;
    // ownership bound of every tile and the tiles of every chromosome (IgdTiles, common.h)
    DevBuf<u32> tile_bnd, chrom_tile_off;
    // static tables of the sweep (IgdTiles::pm / files16 / tab): built once in finish_tiles
    DevBuf<i32> tile_pm;
    DevBuf<unsigned short> tile_files16;
    DevBuf<u32> tile_tab;
    // ... and of its rank-histogram form (IgdTiles::ends_sorted / erank / tab_r; optional: a failed allocation leaves the walk)
    DevBuf<i32> tile_ends_sorted;
    DevBuf<unsigned short> tile_erank;
    DevBuf<u32> tile_tab_r;
    // per chromosome, the reference contig's tile count at nbp = 16384 (min_overlap <= 0 counts), built on first use
    mutable std::mutex ntiles_mu;
    mutable DevBuf<i32> chrom_ntiles;
    mutable bool ntiles_ready = false;
    gtars_status ensure_ntiles() const {
        std::lock_guard<std::mutex> lk(ntiles_mu);
        if (ntiles_ready) return GTARS_OK;
        gtars_status st = ensure_mirror();
        if (st) return st;
        std::vector<u32> off(n_chrom + 1, 0);
        if (n_chrom) GT_HIP(hipMemcpy(off.data(), chrom_off.p, ((size_t)n_chrom + 1) * 4, hipMemcpyDeviceToHost));
        std::vector<i32> nt(std::max<u32>(n_chrom, 1), 0);
        for (u32 c = 0; c < n_chrom; ++c) {
            i32 mx = 0;
            for (u32 p = off[c]; p < off[c + 1]; ++p) mx = std::max(mx, h_ends[p]);
            nt[c] = off[c + 1] > off[c] ? (mx - 1) / 16384 + 1 : 0;
        }
        if ((st = chrom_ntiles.upload(nt))) return st;
        ntiles_ready = true;
        return GTARS_OK;
    }
    // static routing table (IgdTiles::route_*)
    DevBuf<u32> route_lut, route_base, route_len;
    u32 route_n = 0, route_shift = 0;
    DevBuf<u32> route_flut, route_kq, route_fbase;  // IgdTiles::route_f*
    u32 route_fn = 0, route_fshift = 0;
    // IgdTiles::pme_file, built on the first binary count with min_overlap == 1
    mutable std::mutex pme_mu;
    mutable DevBuf<i32> pme_file;
    mutable bool pme_ready = false;
    gtars_status ensure_pme() const {
        std::lock_guard<std::mutex> lk(pme_mu);
        if (pme_ready || n == 0) return GTARS_OK;
        ScopedDev ws;
        const size_t wsb = igd_pme_ws_bytes((u32)n);
        gtars_status st = ws.alloc(wsb);
        if (st) return st;
        i32 *p = nullptr;
        GT_HIP(hipMalloc((void **)&p, (size_t)n * 4 + 32));
        st = igd_build_pme_file(view(), p, ws.p, wsb, nullptr);
        hipError_t e = hipDeviceSynchronize();
        if (st || e != hipSuccess) {
            (void)hipFree(p);
            return st ? st : fail(GTARS_ERR_HIP, "pme_file build failed");
        }
        pme_file.p = p;
        pme_file.n = n;
        pme_ready = true;
        return GTARS_OK;
    }
    // are the records' values all distinct?  (then the per-query "first occurrence of a value" rule keeps every hit);
    // decided on the first per-query call by sorting a copy of the values
    mutable std::mutex uniq_mu;
    mutable int values_unique = -1;
    gtars_status ensure_values_unique(bool *out) const {
        std::lock_guard<std::mutex> lk(uniq_mu);
        if (values_unique < 0) {
            if (n < 2) {
                values_unique = 1;
            } else {
                const u32 n32 = (u32)n;
                ScopedDev buf;
                const size_t wsb = radix_sort_ws_bytes(n32);
                gtars_status st = buf.alloc((size_t)n32 * 16 + wsb + 64);
                if (st) return st;
                u32 *k0 = buf.as<u32>(), *v0 = k0 + n32, *k1 = v0 + n32, *v1 = k1 + n32;
                void *ws = (void *)(v1 + n32);
                u32 *d_dup = (u32 *)((char *)ws + wsb);
                GT_HIP(hipMemcpy(k0, values.p, (size_t)n32 * 4, hipMemcpyDeviceToDevice));
                GT_HIP(hipMemset(d_dup, 0, 4));
                int res = 0;
                st = radix_sort_pairs(k0, v0, k1, v1, n32, 0, 32, ws, wsb, &res, nullptr);
                if (st) return st;
                st = launch_has_adjacent_equal(res ? k1 : k0, n32, d_dup, nullptr);
                if (st) return st;
                u32 h = 1;
                GT_HIP(hipMemcpy(&h, d_dup, 4, hipMemcpyDeviceToHost));
                values_unique = h ? 0 : 1;
            }
        }
        *out = values_unique == 1;
        return GTARS_OK;
    }
    u32 n_tiles = 0;
    IgdTiles tiles() const {
        IgdTiles t;
        t.first = tile_first.p;
        t.cnt = tile_cnt.p;
        t.chrom = tile_chrom.p;
        t.carry = tile_carry.p;
        t.bnd = tile_bnd.p;
        t.chrom_tile_off = chrom_tile_off.p;
        t.pme_file = pme_ready && !cfg_get("GTARS_IGD_NO_PME") ? pme_file.p : nullptr;
        t.pm = tile_pm.p;
        t.files16 = tile_files16.p;
        t.tab = tile_tab.p;
        t.ends_sorted = tile_ends_sorted.p;
        t.erank = tile_erank.p;
        t.tab_r = tile_tab_r.p;
        t.route_lut = route_n ? route_lut.p : nullptr;
        t.route_base = route_base.p;
        t.route_len = route_len.p;
        t.route_n = route_n;
        t.route_shift = route_shift;
        t.route_flut = route_fn ? route_flut.p : nullptr;
        t.route_kq = route_kq.p;
        t.route_fbase = route_fbase.p;
        t.route_fn = route_fn;
        t.route_fshift = route_fshift;
        t.n_tiles = n_tiles;
        return t;
    }
    // after tile_first / tile_cnt / tile_chrom, starts and chrom_maxlen are on the device
    gtars_status finish_tiles(const std::vector<u32> &tch) {
        std::vector<u32> cto(n_chrom + 1, 0);
        for (u32 c : tch) cto[c + 1]++;
        for (u32 c = 0; c < n_chrom; ++c) cto[c + 1] += cto[c];
        gtars_status st = chrom_tile_off.upload(cto);
        if (st) return st;
        std::vector<u32> zero(std::max<u32>(n_tiles, 1), 0);
        if ((st = tile_bnd.upload(zero))) return st;
        if ((st = launch_igd_tile_bounds(view(), tile_first.p, tile_cnt.p, tile_chrom.p, n_tiles, tile_bnd.p, nullptr))) return st;
        if (n_tiles && n_files <= 65535) {
            // one pass over the database: prefix maxima, u16 file ids, search tables (what the sweep streams with the records)
            GT_HIP(hipMalloc((void **)&tile_pm.p, (size_t)n * 4 + 32));  // + slack: staged as whole 16-byte vectors
            tile_pm.n = n;
            GT_HIP(hipMalloc((void **)&tile_files16.p, ((size_t)n + 1) / 2 * 4 + 32));
            tile_files16.n = n;
            GT_HIP(hipMalloc((void **)&tile_tab.p, (size_t)n_tiles * IGD_TILE_TAB_WORDS * 4));
            tile_tab.n = (size_t)n_tiles * IGD_TILE_TAB_WORDS;
            if ((st = launch_igd_tile_tables(view(), tile_first.p, tile_cnt.p, tile_chrom.p, tile_carry.p, n_tiles, tile_pm.p, tile_files16.p,
                                             tile_tab.p, nullptr)))
                return st;
            // the rank-histogram form's blocks (sorted ends + eranks: 6 bytes per staged record, ~7 per record).  Not for a pieces
            // view (its continuation rule is not a rank difference) and not under GTARS_IGD_NO_RANK_TABLES (tests / A-B runs)
            if (!piece_flags && n_files <= 16384 && !cfg_flag("GTARS_IGD_NO_RANK_TABLES")) {
                const size_t slots = (size_t)n_tiles * IGD_TILE_BLOCK;
                hipError_t e = hipMalloc((void **)&tile_ends_sorted.p, slots * 4 + 32);
                if (e == hipSuccess) e = hipMalloc((void **)&tile_erank.p, slots * 2 + 32);
                if (e == hipSuccess) e = hipMalloc((void **)&tile_tab_r.p, (size_t)n_tiles * IGD_TILE_TABR_WORDS * 4);
                if (e == hipSuccess) {
                    tile_ends_sorted.n = tile_erank.n = slots;
                    tile_tab_r.n = (size_t)n_tiles * IGD_TILE_TABR_WORDS;
                    if ((st = launch_igd_tile_tables_rank(view(), tile_first.p, tile_cnt.p, tile_chrom.p, n_tiles, tile_ends_sorted.p,
                                                          tile_erank.p, tile_tab_r.p, nullptr)))
                        return st;
                } else {
                    (void)hipGetLastError();  // out of device memory: the database keeps the walked form
                    tile_ends_sorted.release();
                    tile_erank.release();
                    tile_tab_r.release();
                }
            }
        }
        GT_HIP(hipDeviceSynchronize());
        if (n_tiles && n_tiles < 65535) {
            // routing table over (chromosome, start >> shift), from the tile bounds the device has just computed
            std::vector<u32> hb(n_tiles);
            GT_HIP(hipMemcpy(hb.data(), tile_bnd.p, (size_t)n_tiles * 4, hipMemcpyDeviceToHost));
            std::vector<u32> len(n_chrom, 0), base(n_chrom + 1, 0);
            for (u32 c = 0; c < n_chrom; ++c)
                if (cto[c + 1] > cto[c]) len[c] = hb[cto[c + 1] - 1];
            constexpr u64 ROUTE_MAX = 4096;
            u32 sh = 0;
            for (;; ++sh) {
                u64 tot = 0;
                for (u32 c = 0; c < n_chrom; ++c) tot += ((u64)len[c] >> sh) + 2;
                if (tot <= ROUTE_MAX || sh == 31) break;
            }
            for (u32 c = 0; c < n_chrom; ++c) base[c + 1] = base[c] + (len[c] >> sh) + 2;
            if (base[n_chrom] <= ROUTE_MAX) {
                std::vector<unsigned short> lut(((size_t)base[n_chrom] + 2) & ~(size_t)1, 0);
                for (u32 c = 0; c < n_chrom; ++c) {
                    u32 t = cto[c];
                    const u32 t1 = cto[c + 1], nj = (len[c] >> sh) + 2;
                    for (u32 j = 0; j < nj; ++j) {
                        const u64 x = (u64)j << sh;
                        while (t < t1 && (u64)hb[t] <= x) ++t;  // first tile with bound > x
                        lut[base[c] + j] = (unsigned short)t;
                    }
                }
                std::vector<u32> packed(lut.size() / 2);
                for (size_t w = 0; w < packed.size(); ++w) packed[w] = (u32)lut[2 * w] | ((u32)lut[2 * w + 1] << 16);
                if ((st = route_lut.upload(packed))) return st;
                if ((st = route_base.upload(base))) return st;
                if ((st = route_len.upload(len))) return st;
                route_n = base[n_chrom];
                route_shift = sh;
            }
            // the fine tables (IgdTiles::route_f*), when they fit the routing kernel's LDS next to its 16-bit counters
            u32 fsh0 = 16;
            if (const char *e = cfg_get("GTARS_IGD_ROUTE_FSHIFT_MIN")) fsh0 = (u32)std::min(31, std::max(16, atoi(e)));  // tests: coarse buckets
            for (u32 fsh = fsh0; fsh < 32; ++fsh) {
                u64 nf = 0;
                std::vector<u32> fbase(n_chrom + 1, 0);
                for (u32 c = 0; c < n_chrom; ++c) {
                    const u64 nb = len[c] ? (((u64)len[c] - 1) >> fsh) + 2 : 1;
                    nf += nb;
                    fbase[c + 1] = (u32)std::min<u64>(nf, 0xFFFFFFFFu);
                }
                if (igd_route_fine_lds_bytes(n_tiles, n_chrom, nf) > 160 * 1024 - 64) continue;
                std::vector<unsigned short> fl(((size_t)nf + 2) & ~(size_t)1, 0), kq(((size_t)n_tiles + 2) & ~(size_t)1, 0);
                const u32 mask = (u32)((1ull << fsh) - 1);
                for (u32 c = 0; c < n_chrom; ++c) {
                    u32 t = cto[c];
                    const u32 t1 = cto[c + 1], nb = fbase[c + 1] - fbase[c];
                    for (u32 j = 0; j < nb; ++j) {
                        const u64 x = (u64)j << fsh;
                        while (t < t1 && (u64)hb[t] - 1 < x) ++t;  // first tile with key >= x
                        fl[fbase[c] + j] = (unsigned short)t;
                    }
                    for (u32 k = cto[c]; k < t1; ++k) kq[k] = (unsigned short)(((hb[k] - 1u) & mask) >> (fsh - 16));
                }
                auto pack = [](const std::vector<unsigned short> &v) {
                    std::vector<u32> w(v.size() / 2);
                    for (size_t i = 0; i < w.size(); ++i) w[i] = (u32)v[2 * i] | ((u32)v[2 * i + 1] << 16);
                    return w;
                };
                if ((st = route_flut.upload(pack(fl)))) return st;
                if ((st = route_kq.upload(pack(kq)))) return st;
                if ((st = route_fbase.upload(fbase))) return st;
                route_fn = (u32)nf;
                route_fshift = fsh;
                break;
            }
        }
        return GTARS_OK;
    }
    IgdView view() const {
        IgdView v;
        v.starts = starts.p;
        v.ends = ends.p;
        v.files = files.p;
        v.values = values.p;
        v.chrom_off = chrom_off.p;
        v.chrom_maxlen = chrom_maxlen.p;
        v.chrom_ntiles = ntiles_ready ? chrom_ntiles.p : nullptr;
        v.pm = cfg_get("GTARS_IGD_NO_PM_START") ? nullptr : tile_pm.p;
        v.n_chrom = n_chrom;
        v.n = (u32)n;
        v.n_files = n_files;
        v.pieces = piece_flags ? 1u : 0u;
        return v;
    }
    bool piece_flags = false;  // this IS a pieces view: bit 31 of a file id marks a continuation piece
};

extern "C" {

const char *gtars_last_error(void) { return g_last_error.c_str(); }
const char *gtars_version(void) { return "gtars_amd 0.1.0 (gfx950)"; }

int gtars_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

void gtars_free(void *p) { free(p); }

void gtars_prof_enable(int on) { g_prof_on = on != 0; }
// test hook: take a new snapshot of the GTARS_* environment (common.h: cfg_get); the caller makes sure that no call is in flight
void gtars_debug_reload_env(void) {
    auto m = take_env_snapshot();
    std::lock_guard<std::mutex> lk(g_env_mu);
    if (g_env) g_env_old.push_back(g_env);  // (a value handed out by cfg_get stays valid)
    g_env = m;
}
void gtars_prof_reset(void) {
    prof_drain();
    g_prof_entries.clear();
}
int gtars_prof_read(const char **names, double *total_ms, uint64_t *launches, int cap) {
    prof_drain();
    int n = (int)g_prof_entries.size();
    for (int i = 0; i < n && i < cap; ++i) {
        if (names) names[i] = g_prof_entries[i].name.c_str();
        if (total_ms) total_ms[i] = g_prof_entries[i].total_ms;
        if (launches) launches[i] = g_prof_entries[i].launches;
    }
    return n;
}

// ------------------------------------------------------------- index build

static gtars_status gtars_index_build_impl(const uint32_t *chrom, const uint32_t *start, const uint32_t *end,
                               const uint32_t *val, uint64_t n, uint32_t n_chrom, int kind,
                               gtars_index_t **out) {
    if (!out) return fail(GTARS_ERR_INVALID_ARG, "out is NULL");
    *out = nullptr;
    if (kind != GTARS_KIND_BITS && kind != GTARS_KIND_AILIST)
        return fail(GTARS_ERR_INVALID_ARG, "kind must be GTARS_KIND_BITS or GTARS_KIND_AILIST");
    if (n && (!chrom || !start || !end)) return fail(GTARS_ERR_INVALID_ARG, "NULL interval arrays");
    if (n >= 0xFFFFFFF0ull) return fail(GTARS_ERR_INVALID_ARG, "too many intervals for a u32-indexed index");
    gtars_status st = require_device();
    if (st) return st;

    // bucket by chromosome, keeping input order (utils/mod.rs:57-87)
    std::vector<u32> cnt(n_chrom + 1, 0);
    for (u64 i = 0; i < n; ++i) {
        if (chrom[i] >= n_chrom) return fail(GTARS_ERR_INVALID_ARG, "interval chromosome id >= n_chrom");
        cnt[chrom[i] + 1]++;
    }
    std::vector<u32> off(n_chrom + 1, 0);
    for (u32 c = 0; c < n_chrom; ++c) off[c + 1] = off[c] + cnt[c + 1];
    std::vector<u32> perm(n);
    {
        std::vector<u32> fillp(off.begin(), off.end() - 1);
        for (u64 i = 0; i < n; ++i) perm[fillp[chrom[i]]++] = (u32)i;
    }

    auto *ix = new gtars_index();
    ix->kind = kind;
    ix->n_chrom = n_chrom;
    ix->n = n;
    ix->h_chrom_off = off;
    ix->h_starts.resize(n);
    ix->h_ends.resize(n);
    ix->h_vals.resize(n);
    ix->h_chrom_aux.assign(n_chrom, 0);

    if (kind == GTARS_KIND_BITS) {
        const bool dev_sort = use_device_sort(n);
        if (dev_sort) {
            // K1: (chrom, start, end, input order) by stable radix passes on the device
            std::vector<u32> hc(chrom, chrom + n), hs(start, start + n), he(end, end + n);
            gtars_status s1 = sorted_perm_device(hc, hs, &he, n_chrom, perm);
            if (s1) {
                delete ix;
                return s1;
            }
        }
        for (u32 c = 0; c < n_chrom; ++c) {
            // Bits::build: stable sort by (start, end) (bits.rs:105, interval.rs:18-31)
            if (!dev_sort)
                std::stable_sort(perm.begin() + off[c], perm.begin() + off[c + 1], [&](u32 a, u32 b) {
                    if (start[a] != start[b]) return start[a] < start[b];
                    return end[a] < end[b];
                });
            u32 max_len = 0;
            for (u32 p = off[c]; p < off[c + 1]; ++p) {
                const u32 i = perm[p];
                const u32 len = end[i] >= start[i] ? end[i] - start[i] : 0;  // checked_sub -> 0
                max_len = std::max(max_len, len);
            }
            ix->h_chrom_aux[c] = max_len;
        }
        for (u64 p = 0; p < n; ++p) {
            const u32 i = perm[p];
            ix->h_starts[p] = start[i];
            ix->h_ends[p] = end[i];
            ix->h_vals[p] = val ? val[i] : i;
        }
        ix->h_rows = perm;
    } else {
        // AIList::build (ailist.rs:105-151) + decompose (ailist.rs:198-236)
        ix->h_max_ends.resize(n);
        ix->h_rows.resize(n);
        ix->h_chrom_sub.assign(n_chrom + 1, 0);
        const size_t min_cov = 10;
        u64 filled = 0;
        for (u32 c = 0; c < n_chrom; ++c) {
            ix->h_chrom_sub[c] = (u32)ix->h_sub_off.size();
            std::stable_sort(perm.begin() + off[c], perm.begin() + off[c + 1],
                             [&](u32 a, u32 b) { return start[a] < start[b]; });
            std::vector<u32> cur(perm.begin() + off[c], perm.begin() + off[c + 1]), l2;
            if (cur.empty()) {
                ix->h_sub_off.push_back((u32)filled);  // terminator only
                continue;
            }
            ix->h_sub_off.push_back((u32)filled);
            for (;;) {
                l2.clear();
                const u64 first = filled;
                for (size_t idx = 0; idx < cur.size(); ++idx) {
                    const u32 i = cur[idx];
                    size_t count = 0;
                    for (size_t k = 1; k < min_cov * 2; ++k) {
                        if (idx + k >= cur.size()) break;
                        if (end[i] > end[cur[idx + k]]) count++;
                    }
                    if (count >= min_cov) {
                        l2.push_back(i);
                    } else {
                        ix->h_starts[filled] = start[i];
                        ix->h_ends[filled] = end[i];
                        ix->h_vals[filled] = val ? val[i] : i;
                        ix->h_rows[filled] = i;
                        filled++;
                    }
                }
                u32 mx = 0;
                for (u64 p = first; p < filled; ++p) {
                    mx = std::max(mx, ix->h_ends[p]);
                    ix->h_max_ends[p] = mx;
                }
                cur.swap(l2);
                if (cur.empty()) break;
                ix->h_sub_off.push_back((u32)filled);
            }
            ix->h_sub_off.push_back((u32)filled);  // terminator for this chromosome
        }
        // chrom_sub[c]..chrom_sub[c+1] spans the boundaries incl. terminator
        ix->h_chrom_sub[n_chrom] = (u32)ix->h_sub_off.size();
    }

    std::vector<u32> h_rec2, h_rec4, h_blk_first, h_lut, h_q, h_cblk, h_ctab, h_idc, h_iv_end;
    // The blocked structure is skipped (generic kernels serve the index) when it would not fit the LDS
    // kernels anyway: more units than the LDS budget holds even at the coarsest unit size (thousands of
    // non-empty contigs: every one needs at least one unit), or more blocks than a query's state word
    // addresses.  Decided BEFORE anything is allocated: padded block counts are computed in 64 bits.
    // AIList kind: a chromosome with ONE sub-list is its intervals in start order with max_ends as prefix maximum --
    // exactly what the blocked structure is built over; AIList::find then yields the same hits in DESCENDING stored
    // order (ailist.rs:238-263), which the LDS kernel emits by reversing each query's hits.  Indexes with a second
    // sub-list anywhere (intervals that contain >= 10 of their next 20 neighbours, ailist.rs:198-223) keep the generic kernel.
    bool single_sublist = kind == GTARS_KIND_AILIST;
    if (kind == GTARS_KIND_AILIST)
        for (u32 c = 0; c < n_chrom; ++c)
            if (ix->h_chrom_sub[c + 1] - ix->h_chrom_sub[c] > 2) single_sublist = false;  // boundaries incl. terminator
    bool build_accel = (kind == GTARS_KIND_BITS || single_sublist) && n > 0;
    u32 shift = 0;
    if (build_accel) {
        // LDS budget of the tokenizer kernels: one 1024-thread workgroup per CU with its copy of the unit keys
        // (2 B per unit), the bucket table (2 B per bucket, at most 4096 + 1) and the chromosome tables
        constexpr u32 kBucketMax = 4096;
        u32 unit_max = 1024;
        {
            const long budget = 148 * 1024 - 20l * (long)n_chrom - 16 - 2l * (kBucketMax + 8);
            if (budget > 4096) unit_max = (u32)std::min<long>(budget / 2 / 8 * 8, 65528);  // lut entries are u16
        }
        if (const char *e = cfg_get("GTARS_TOP_MAX")) {
            const long v = atol(e);
            if (v >= 64) unit_max = (u32)std::min<long>(v, 65528);
        }
        u64 nb64 = 0;
        for (;;) {
            // padded block count at this shift
            nb64 = 0;
            const u64 g = 1ull << shift;
            for (u32 c = 0; c < n_chrom; ++c) {
                const u64 b = ((u64)off[c + 1] - off[c] + ACC_OWN - 1) / ACC_OWN;
                nb64 += (b + g - 1) / g * g;
            }
            if ((nb64 >> shift) <= unit_max || shift >= 16) break;
            ++shift;
        }
        if ((nb64 >> shift) > unit_max || nb64 > (u64)((1u << 22) - 1u)) build_accel = false;
    }
    if (build_accel) {
        const u64 g = 1ull << shift;
        h_cblk.assign(n_chrom + 1, 0);
        for (u32 c = 0; c < n_chrom; ++c) {
            const u64 b = ((u64)off[c + 1] - off[c] + ACC_OWN - 1) / ACC_OWN;
            h_cblk[c + 1] = h_cblk[c] + (u32)((b + g - 1) / g * g);
        }
        const u32 nb = h_cblk[n_chrom];
        // ids that follow from the position: within every chromosome the stored values ascend by one
        bool affine = true;
        h_idc.assign(n_chrom, 0);
        for (u32 c = 0; c < n_chrom && affine; ++c) {
            if (off[c + 1] == off[c]) continue;
            const u32 v0 = ix->h_vals[off[c]];
            for (u32 p = off[c]; p < off[c + 1]; ++p)
                if (ix->h_vals[p] != v0 + (p - off[c])) {
                    affine = false;
                    break;
                }
            h_idc[c] = v0 - (u32)ACC_OWN * h_cblk[c];  // mod 2^32
        }
        if (cfg_get("GTARS_NO_AFFINE_IDS")) affine = false;  // tests: force the id records
        ix->acc_ids_affine = affine;
        h_rec2.assign((size_t)nb * 8, 0);
        if (!affine) h_rec4.assign((size_t)nb * 16, 0);
        h_blk_first.assign(nb, 0xFFFFFFFFu);
        std::vector<u64> chrom_span(n_chrom, 0);  // max end + 1 (0: no intervals)
        // AccelView::runs_ok / ends_mono / chrom_iv_end: is no interval inverted; do the ends ascend with the starts everywhere?
        bool mono = true, upright = true;
        h_iv_end.assign(n_chrom, 0);
        for (u32 c = 0; c < n_chrom; ++c) {
            u32 real = 0;
            for (u32 p = off[c]; p < off[c + 1]; ++p) {
                if (ix->h_starts[p] > ix->h_ends[p]) upright = false;
                if (p > off[c] && ix->h_ends[p] < ix->h_ends[p - 1]) mono = false;
                if (ix->h_starts[p] != 0xFFFFFFFFu) real = p - off[c] + 1;
            }
            h_iv_end[c] = (u32)ACC_OWN * h_cblk[c] + real;
        }
        ix->acc_runs_ok = upright;
        ix->acc_ends_mono = mono && upright;
        for (u32 c = 0; c < n_chrom; ++c) {
            u32 pm = 0;  // prefix max of the ends, in (start, end) order
            for (u32 b = h_cblk[c]; b < h_cblk[c + 1]; ++b) {
                // slots 0, 1: own intervals, slots 2, 3: look-ahead = the next block's intervals
                u32 *r2 = &h_rec2[(size_t)b * 8];
                u32 *r4 = affine ? nullptr : &h_rec4[(size_t)b * 16];
                for (int k = 0; k < ACC_SLOTS; ++k) {
                    const u64 p = (u64)off[c] + (u64)(b - h_cblk[c]) * ACC_OWN + k;
                    const bool ok = p < off[c + 1];
                    const u32 st_ = ok ? ix->h_starts[p] : 0xFFFFFFFFu;  // sentinel: never < q_end, stops the scan
                    const u32 en_ = ok ? ix->h_ends[p] : 0u;
                    r2[k] = st_;
                    r2[4 + k] = en_;
                    if (r4) {
                        r4[k] = st_;
                        r4[4 + k] = en_;
                        r4[8 + k] = ok ? ix->h_vals[p] : 0u;
                    }
                }
                // Search key of the block: the largest end among ALL intervals up to and including its own
                // (a prefix maximum, so it ascends).  An interval overlaps a query only if its end is
                // > q_start, hence the first block whose key is > q_start holds the first possible hit --
                // a tighter start than Bits::find's lower_bound(q_start - max_len) (bits.rs:144-147), with
                // the same hit set and order, and immune to a few very wide intervals inflating max_len.
                // Padding blocks (no own interval) keep the sentinel key.
                const u64 p0 = (u64)off[c] + (u64)(b - h_cblk[c]) * ACC_OWN;
                for (int k = 0; k < ACC_OWN; ++k)
                    if (p0 + k < off[c + 1]) pm = std::max(pm, ix->h_ends[p0 + k]);
                if (p0 < off[c + 1]) h_blk_first[b] = pm;
            }
            chrom_span[c] = off[c + 1] > off[c] ? (u64)pm + 1 : 0;
        }
        // Unit keys (key of the last block of every 2^shift blocks) in one ascending key space: chromosome
        // c's keys live in [gbase[c], gbase[c] + span[c]], span = max end + 1 being the sentinel key.
        // Consecutive chromosomes are 2^q_shift apart, so that the floor-quantised in-bucket search can
        // never stop on a block of an EARLIER chromosome (its keys quantise strictly below the target).
        const u32 n_units = nb >> shift;
        std::vector<u64> uk(n_units);
        u64 total = 0;
        for (u32 c = 0; c < n_chrom; ++c) total += chrom_span[c] + 1;
        // The bucket table is as fine as the LDS budget allows: what the unit keys (2 B each), the chromosome tables and
        // 16 KB of id staging (256 words per wave) leave of the kernels' 158 KB, up to 16384 buckets -- the in-bucket search
        // reads the bucket's first 8 keys at once, so the fewer units a bucket holds the better (100k-region universe:
        // 50k units, 16384 buckets, 3 units per bucket on average).  GTARS_TOK_BUCKETS overrides (A/B runs).
        u32 kBucketMax = 4096;
        {
            const long left = 158l * 1024 - 16l * 1024 - 2l * (long)(n_units + 16) - 20l * (long)n_chrom - 64;
            while (kBucketMax < 16384 && 2l * (2 * (long)kBucketMax + 16) <= left) kBucketMax *= 2;
            if (const char *e = cfg_get("GTARS_TOK_BUCKETS")) {
                const long v = atol(e);
                if (v >= 64 && v <= 65536) kBucketMax = (u32)v;
            }
        }
        // bucket width 2^lsh: the smallest that needs <= kBucketMax buckets (padding between chromosomes included)
        u32 lsh = 4, qsh = 0;
        for (;; ++lsh) {
            qsh = lsh > 16 ? lsh - 16 : 0;
            const u64 padded = total + (u64)n_chrom * (1ull << qsh);
            if (((padded >> lsh) + 1) <= kBucketMax || lsh >= 40) break;
        }
        u64 gbase = 0;
        h_ctab.assign((size_t)n_chrom * 4, 0);
        for (u32 c = 0; c < n_chrom; ++c) {
            const u64 span = chrom_span[c];
            h_ctab[4 * (size_t)c + 0] = (u32)(gbase & 0xFFFFFFFFu);
            h_ctab[4 * (size_t)c + 1] = (u32)std::min<u64>(span, 0xFFFFFFFFu);
            h_ctab[4 * (size_t)c + 2] = (u32)(gbase >> 32);  // the key space is 64 bits wide
            h_ctab[4 * (size_t)c + 3] = h_cblk[c + 1];
            for (u32 t = h_cblk[c] >> shift; t < (h_cblk[c + 1] >> shift); ++t)
                uk[t] = gbase + std::min<u64>(h_blk_first[(((size_t)t + 1) << shift) - 1], span);
            gbase += span + (1ull << qsh);
        }
        const u32 n_buckets = (u32)std::min<u64>((gbase >> lsh) + 1, 1u << 20);
        std::vector<uint16_t> lut16(((size_t)n_buckets + 1 + 7) & ~(size_t)7, 0), q16(((size_t)n_units + 8 + 7) & ~(size_t)7, 0xFFFFu);  // + 8: the in-bucket search reads 8 keys from any unit on
        u32 max_occ = 0;
        {
            u32 u = 0;
            for (u32 b = 0; b <= n_buckets; ++b) {
                while (u < n_units && uk[u] < ((u64)b << lsh)) ++u;  // lut[b] = units with key < b << lsh
                lut16[b] = (uint16_t)u;
                if (b) max_occ = std::max<u32>(max_occ, (u32)lut16[b] - (u32)lut16[b - 1]);
            }
            for (size_t b = n_buckets + 1; b < lut16.size(); ++b) lut16[b] = (uint16_t)n_units;
        }
        for (u32 u = 0; u < n_units; ++u) q16[u] = (uint16_t)((uk[u] & ((1ull << lsh) - 1)) >> qsh);
        u32 search_top = 0;  // in-bucket search covers ranges up to 2 * search_top - 1 units
        while (2 * search_top < max_occ + 1) search_top = search_top ? search_top * 2 : 1;
        h_lut.assign(lut16.size() / 2, 0);
        memcpy(h_lut.data(), lut16.data(), lut16.size() * sizeof(uint16_t));
        h_q.assign(q16.size() / 2, 0);
        memcpy(h_q.data(), q16.data(), q16.size() * sizeof(uint16_t));
        ix->acc_n_blocks = nb;
        ix->acc_n_units = n_units;
        ix->acc_n_buckets = n_buckets;
        ix->acc_lut_words = (u32)h_lut.size();
        ix->acc_q_words = (u32)h_q.size();
        ix->acc_lut_shift = lsh;
        ix->acc_q_shift = qsh;
        ix->acc_search_top = search_top;
        ix->acc_top_shift = shift;
        // the key space is 64 bits wide (hg38 needs 3.1e9; larger genomes just get wider buckets)
        ix->has_accel = nb > 0 && (gbase >> lsh) < (1ull << 31);
    }

    GT_HIP(hipGetDevice(&ix->device));
    st = ix->starts.upload(ix->h_starts);
    if (!st && ix->has_accel) st = ix->acc_rec2.upload(h_rec2);
    if (!st && ix->has_accel && !ix->acc_ids_affine) st = ix->acc_rec4.upload(h_rec4);
    if (!st && ix->has_accel && ix->acc_ids_affine && ix->acc_top_shift == 1) {
        // unit records (AccelView::rec8): unit u = blocks 2u, 2u + 1; slots 0-3 = their own intervals (= the four slots of block 2u's
        // 32-byte record), slots 4-7 = the own intervals of blocks 2u + 2, 2u + 3 (= the four slots of block 2u + 2's record), or
        // sentinels behind the chromosome's last unit
        const u32 nbk = ix->acc_n_blocks, nun = nbk >> 1;
        std::vector<u32> h_rec8((size_t)nun * 16);
        std::vector<u32> blk_chrom_end(nbk, 0);
        for (u32 c = 0; c < n_chrom; ++c)
            for (u32 b = h_cblk[c]; b < h_cblk[c + 1]; ++b) blk_chrom_end[b] = h_cblk[c + 1];
        for (u32 u = 0; u < nun; ++u) {
            const u32 b = 2u * u;
            u32 *r = &h_rec8[(size_t)u * 16];
            const u32 *lo = &h_rec2[(size_t)b * 8];
            const bool has_next = b + 2u < blk_chrom_end[b];
            const u32 *hi = has_next ? &h_rec2[(size_t)(b + 2u) * 8] : nullptr;
            for (int k = 0; k < 4; ++k) {
                r[k] = lo[k];
                r[8 + k] = lo[4 + k];
                r[4 + k] = hi ? hi[k] : 0xFFFFFFFFu;
                r[12 + k] = hi ? hi[4 + k] : 0u;
            }
        }
        st = ix->acc_rec8.upload(h_rec8);
    }
    if (!st && ix->has_accel) st = ix->acc_idc.upload(h_idc);
    if (!st && ix->has_accel) {
        std::vector<u32> h_idc_pos(n_chrom, 0);
        for (u32 c = 0; c < n_chrom; ++c) h_idc_pos[c] = off[c] - (u32)ACC_OWN * h_cblk[c];  // mod 2^32
        st = ix->acc_idc_pos.upload(h_idc_pos);
    }
    if (!st && ix->has_accel) st = ix->acc_blk_first.upload(h_blk_first);
    if (!st && ix->has_accel) st = ix->acc_chrom_iv_end.upload(h_iv_end);
    if (!st && ix->has_accel) st = ix->acc_lut.upload(h_lut);
    if (!st && ix->has_accel) st = ix->acc_qkeys.upload(h_q);
    if (!st && ix->has_accel) st = ix->acc_chrom_tab.upload(h_ctab);
    if (!st) st = ix->ends.upload(ix->h_ends);
    if (!st) st = ix->vals.upload(ix->h_vals);
    if (!st) st = ix->max_ends.upload(ix->h_max_ends);
    if (!st) st = ix->chrom_off.upload(ix->h_chrom_off);
    if (!st) st = ix->chrom_aux.upload(ix->h_chrom_aux);
    if (!st) st = ix->chrom_sub.upload(ix->h_chrom_sub);
    if (!st) st = ix->sub_off.upload(ix->h_sub_off);
    if (st) {
        gtars_index_free(ix);
        return st;
    }
    if (kind == GTARS_KIND_AILIST && !single_sublist && n > 0) {
        // nested sub-lists: the flat companion (see gtars_index::flat) -- optional: an index without it answers every call on
        // the generic kernels, as before
        gtars_index *fl = nullptr;
        if (gtars_index_build_impl(chrom, start, end, val, n, n_chrom, GTARS_KIND_BITS, &fl) == GTARS_OK && fl && fl->has_accel &&
            tokenize_lds_supported(fl->accel())) {
            std::vector<u32> inv(n), map(n);
            for (u64 p = 0; p < n; ++p) inv[ix->h_rows[p]] = (u32)p;
            for (u64 p = 0; p < n; ++p) map[p] = inv[fl->h_rows[p]];
            if (ix->flat_pos.upload(map) == GTARS_OK) {
                ix->flat = fl;
                fl = nullptr;
                // mirrored positions: for stored position a of sub-list [s0, s1): key = s0 + (s1 - 1 - a)
                std::vector<u32> mirror(n), key(n), val(n);
                for (u32 c = 0; c < n_chrom; ++c)
                    for (u32 h = ix->h_chrom_sub[c]; h + 1 < ix->h_chrom_sub[c + 1]; ++h) {
                        const u32 s0 = ix->h_sub_off[h], s1 = ix->h_sub_off[h + 1];
                        for (u32 a = s0; a < s1; ++a) mirror[a] = s0 + (s1 - 1u - a);
                    }
                for (u64 p = 0; p < n; ++p) {
                    key[p] = mirror[map[p]];
                    val[mirror[map[p]]] = ix->h_vals[map[p]];
                }
                {
                    double bp = 0, span = 0;
                    for (u32 c = 0; c < n_chrom; ++c) {
                        const u32 a0 = ix->h_chrom_off[c], a1 = ix->h_chrom_off[c + 1];
                        if (a0 == a1) continue;
                        u32 lo = 0xFFFFFFFFu, hi = 0;
                        for (u32 a = a0; a < a1; ++a) {
                            lo = std::min(lo, ix->h_starts[a]);
                            hi = std::max(hi, ix->h_ends[a]);
                            if (ix->h_ends[a] > ix->h_starts[a]) bp += (double)(ix->h_ends[a] - ix->h_starts[a]);
                        }
                        if (hi > lo) span += (double)(hi - lo);
                    }
                    ix->ail_depth = span > 0 ? bp / span : 0;
                }
                if (ix->ail_key.upload(key) != GTARS_OK || ix->ail_val.upload(val) != GTARS_OK) {
                    ix->ail_key.release();  // (out of device memory: enumeration stays on the generic kernel)
                    ix->ail_val.release();
                }
            }
        }
        if (fl) gtars_index_free(fl);
        set_error("");
    }
    *out = ix;
    return GTARS_OK;
}

void gtars_index_free(gtars_index_t *ix) {
    if (!ix) return;
    gtars_index_free(ix->flat);
    ix->flat_pos.release();
    ix->ail_key.release();
    ix->ail_val.release();
    ix->starts.release();
    ix->ends.release();
    ix->vals.release();
    ix->max_ends.release();
    ix->chrom_off.release();
    ix->chrom_aux.release();
    ix->chrom_sub.release();
    ix->sub_off.release();
    ix->acc_rec2.release();
    ix->acc_rec4.release();
    ix->acc_rec8.release();
    ix->acc_idc.release();
    ix->acc_idc_pos.release();
    ix->acc_blk_first.release();
    ix->acc_chrom_iv_end.release();
    ix->acc_lut.release();
    ix->ends_sorted.release();
    ix->acc_qkeys.release();
    ix->acc_chrom_tab.release();
    delete ix;
}

uint64_t gtars_index_len(const gtars_index_t *ix) { return ix ? ix->n : 0; }
uint32_t gtars_index_n_chrom(const gtars_index_t *ix) { return ix ? ix->n_chrom : 0; }
int gtars_index_kind(const gtars_index_t *ix) { return ix ? ix->kind : -1; }
int gtars_index_device(const gtars_index_t *ix) { return ix ? ix->device : -1; }

uint64_t gtars_index_chrom_len(const gtars_index_t *ix, uint32_t c) {
    if (!ix || c >= ix->n_chrom) return 0;
    return ix->h_chrom_off[c + 1] - ix->h_chrom_off[c];
}

gtars_status gtars_index_stored(const gtars_index_t *ix, uint32_t c, uint32_t *start, uint32_t *end,
                                uint32_t *val) {
    if (!ix) return fail(GTARS_ERR_INVALID_ARG, "NULL index");
    if (c >= ix->n_chrom) return GTARS_OK;
    const u32 lo = ix->h_chrom_off[c], hi = ix->h_chrom_off[c + 1];
    // read back from the device so that tests see what the kernels see
    if (hi > lo) {
        if (start) GT_HIP(hipMemcpy(start, ix->starts.p + lo, (hi - lo) * 4, hipMemcpyDeviceToHost));
        if (end) GT_HIP(hipMemcpy(end, ix->ends.p + lo, (hi - lo) * 4, hipMemcpyDeviceToHost));
        if (val) GT_HIP(hipMemcpy(val, ix->vals.p + lo, (hi - lo) * 4, hipMemcpyDeviceToHost));
    }
    return GTARS_OK;
}

uint32_t gtars_index_max_len(const gtars_index_t *ix, uint32_t c) {
    if (!ix || c >= ix->n_chrom || ix->kind != GTARS_KIND_BITS) return 0;
    return ix->h_chrom_aux[c];
}

uint64_t gtars_index_n_sublists(const gtars_index_t *ix, uint32_t c) {
    if (!ix || c >= ix->n_chrom || ix->kind != GTARS_KIND_AILIST) return 0;
    const u32 nb = ix->h_chrom_sub[c + 1] - ix->h_chrom_sub[c];
    return nb ? nb - 1 : 0;
}

gtars_status gtars_index_sublist_offsets(const gtars_index_t *ix, uint32_t c, uint64_t *out) {
    if (!ix || !out) return fail(GTARS_ERR_INVALID_ARG, "NULL argument");
    if (c >= ix->n_chrom || ix->kind != GTARS_KIND_AILIST) return GTARS_OK;
    const u32 b = ix->h_chrom_sub[c], e = ix->h_chrom_sub[c + 1];
    for (u32 h = b; h + 1 < e; ++h) out[h - b] = ix->h_sub_off[h] - ix->h_chrom_off[c];
    return GTARS_OK;
}

// ----------------------------------------------------------------- queries

// fused enumerate dispatcher: LDS-tiled fast path for Bits, generic kernel otherwise
static size_t fused_ws_bytes(const gtars_index *ix, u64 nq) {
    return std::max(enumerate_fused_ws_bytes(nq), tokenize_lds_ws_bytes(nq));
}
static bool use_lds_path(const gtars_index *ix) {
    const bool disabled = cfg_flag("GTARS_NO_LDS_PATH") || cfg_flag("GTARS_NO_LDS_PATH_FOR_TEST");  // A/B runs, tests
    // has_accel: a Bits-kind index, or an AIList-kind one whose chromosomes all have a single sub-list
    return !disabled && ix->has_accel && tokenize_lds_supported(ix->accel());
}

static bool sweep_wanted(const gtars_index *ix, const EnumOut &out) {
    if (cfg_flag("GTARS_NO_LDS_PATH") || cfg_flag("GTARS_NO_LDS_PATH_FOR_TEST") || cfg_flag("GTARS_TOK_NO_SWEEP")) return false;
    return (out.sorted || cfg_flag("GTARS_TOK_SWEEP")) && ix->has_accel && tokenize_sweep_supported(ix->accel());
}

// K2 dispatch: Bits-kind indexes with the blocked structure count through k_count_lds, everything else
// (AIList order is irrelevant for counts, but its index has no blocked structure) through k_count
// the index whose blocked structure answers an ORDER-INDEPENDENT call on `ix`: ix itself, the flat companion of a nested AIList
// index (gtars_index::flat), or null (generic kernels)
static const gtars_index *lds_target(const gtars_index *ix) {
    if (use_lds_path(ix)) return ix;
    if (ix->flat && use_lds_path(ix->flat)) return ix->flat;
    return nullptr;
}
static gtars_status count_dispatch(const gtars_index *ix, const u32 *qc, const u32 *qs, const u32 *qe, u64 nq, int has_min,
                                   i32 min_overlap, u32 *counts, u8 *any, hipStream_t st) {
    if (const gtars_index *t = lds_target(ix)) return launch_count_lds(t->accel(), qc, qs, qe, nq, has_min, min_overlap, counts, any, st);
    return launch_count(ix->view(), ix->kind, qc, qs, qe, nq, has_min, min_overlap, counts, any, st);
}
static gtars_status run_fused(const gtars_index *ix, const u32 *qc, const u32 *qs, const u32 *qe, u64 nq,
                              int has_min, i32 min_overlap, const EnumOut &out, void *ws, size_t ws_bytes,
                              ScanEpoch &ep, hipStream_t s) {
    // hit counts are summed in 32 bits per tile: refuse a batch whose smallest tile could overflow them (every query
    // of the tile overlapping every interval of the largest chromosome) instead of returning wrapped offsets
    if ((u64)ix->max_chrom_n() * (u64)enumerate_fused_tile_queries() > 0xFFFFFFFFull)
        return fail(GTARS_ERR_INVALID_ARG, "index too dense: a tile of queries could have more than 2^32 - 1 hits");
    // a batch in (chromosome, start) order (the caller's hint, or GTARS_TOK_SWEEP=1: tests and A/B runs): the sweep form -- needs the
    // blocked records only, not the LDS search image, so it also serves universes beyond k_tok_lds' key budget
    if (sweep_wanted(ix, out) && !out.starts && !out.ends)
        return launch_tokenize_sweep(ix->accel(), qc, qs, qe, nq, has_min, min_overlap, out, ws, ws_bytes, ep, s, nullptr, nullptr,
                                     ix->kind == GTARS_KIND_AILIST);
    if (use_lds_path(ix) && !out.starts && !out.ends)
        return launch_tokenize_lds(ix->accel(), qc, qs, qe, nq, has_min, min_overlap, out, ws, ws_bytes, ep, s, nullptr, nullptr,
                                   ix->kind == GTARS_KIND_AILIST);
    if (ix->kind == GTARS_KIND_AILIST && ix->flat && ix->ail_key.p && ix->ail_val.p && use_lds_path(ix->flat) && !out.starts && !out.ends &&
        !cfg_flag("GTARS_AILIST_NO_REORDER") && ix->ail_depth < (double)cfg_int("GTARS_AILIST_REORDER_MAX_DEPTH", 6)) {
        // a nested AIList index: the hit SET from the flat companion's LDS tokenizer (its stored positions, Bits order), the ORDER
        // by k_ailist_reorder -- one more pass over the offsets and ids instead of the one-thread-per-query generic kernel
        gtars_status st = launch_tokenize_lds(ix->flat->accel_pos(), qc, qs, qe, nq, has_min, min_overlap, out, ws, ws_bytes, ep, s, nullptr,
                                              nullptr, false);
        if (st || !out.vals) return st;  // (offsets only: the counts do not depend on the order)
        prof_note_fact("ailist_nested_on_lds");
        return launch_ailist_reorder(out.vals, out.offsets, nq, out.capacity, ix->ail_key.p, ix->ail_val.p, s);
    }
    ep = ScanEpoch();  // the generic kernel clears the workspace itself
    return launch_enumerate_fused(ix->view(), ix->kind, qc, qs, qe, nq, has_min, min_overlap, out, ws,
                                  ws_bytes, s);
}
static gtars_status read_scan_head(const void *ws, hipStream_t s, u64 *total) {
    ScanHead h;
    GT_HIP(hipMemcpyAsync(&h, ws, sizeof h, hipMemcpyDeviceToHost, s));
    GT_HIP(hipStreamSynchronize(s));
    if (total) *total = h.total;
    if (h.err) {
        (void)hipMemsetAsync(const_cast<void *>(ws), 0, sizeof(ScanHead), s);
        return fail(GTARS_ERR_INTERNAL, "chained scan timed out (look-back spin limit)");
    }
    return GTARS_OK;
}

// run_fused + read back {total, err}.  The LDS kernel assumes that its whole grid is resident; if other
// work holds CUs and a look-back spin runs into its limit, the batch is redone by the generic kernel, which
// draws a ticket for every tile and therefore only ever waits on tiles that are running.
static gtars_status run_fused_sync(const gtars_index *ix, const u32 *qc, const u32 *qs, const u32 *qe, u64 nq, int has_min,
                                   i32 min_overlap, const EnumOut &out, void *ws, size_t ws_bytes, ScanEpoch &ep,
                                   hipStream_t s, u64 *total) {
    gtars_status st = run_fused(ix, qc, qs, qe, nq, has_min, min_overlap, out, ws, ws_bytes, ep, s);
    if (st) return st;
    st = read_scan_head(ws, s, total);
    if (st == GTARS_OK && cfg_get("GTARS_TEST_FORCE_LOOKBACK_TIMEOUT")) st = GTARS_ERR_INTERNAL;  // test hook
    if (st != GTARS_ERR_INTERNAL || !use_lds_path(ix) || out.starts || out.ends) return st;
    ep = ScanEpoch();
    st = launch_enumerate_fused(ix->view(), ix->kind, qc, qs, qe, nq, has_min, min_overlap, out, ws, ws_bytes, s);
    if (st) return st;
    return read_scan_head(ws, s, total);
}

static gtars_status check_query_args(const void *ix, const void *a, const void *b, const void *c, u64 nq) {
    if (!ix) return fail(GTARS_ERR_INVALID_ARG, "NULL handle");
    if (nq && (!a || !b || !c)) return fail(GTARS_ERR_INVALID_ARG, "NULL query arrays");
    return GTARS_OK;
}

gtars_status gtars_tokenize_device(const gtars_index_t *ix, const uint32_t *d_qc, const uint32_t *d_qs,
                                   const uint32_t *d_qe, uint64_t nq, uint64_t *d_offsets, uint32_t *d_ids,
                                   uint64_t ids_capacity, uint64_t *total_hits, void *stream) {
    return gtars_tokenize_device_ex(ix, d_qc, d_qs, d_qe, nq, d_offsets, d_ids, ids_capacity, total_hits, stream, GTARS_TOK_AUTO);
}

gtars_status gtars_tokenize_device_ex(const gtars_index_t *ix, const uint32_t *d_qc, const uint32_t *d_qs,
                                      const uint32_t *d_qe, uint64_t nq, uint64_t *d_offsets, uint32_t *d_ids,
                                      uint64_t ids_capacity, uint64_t *total_hits, void *stream, int hint) {
    GT_SAME_DEVICE_AS(ix);
    gtars_status st = check_query_args(ix, d_qc, d_qs, d_qe, nq);
    if (st) return st;
    if (!d_offsets) return fail(GTARS_ERR_INVALID_ARG, "d_offsets is NULL");
    if ((hint & ~(3 | GTARS_TOK_SORTED)) || (hint & 3) == 3) return fail(GTARS_ERR_INVALID_ARG, "unknown tokenizer hint");
    hipStream_t s = (hipStream_t)stream;
    Workspace &ws = tls_workspace(0, s);
    const size_t wsb = fused_ws_bytes(ix, nq);
    st = ws.reserve(wsb);
    if (st) return st;
    EnumOut out{d_offsets, d_ids, nullptr, nullptr, d_ids ? ids_capacity : 0, hint & 3, (hint & GTARS_TOK_SORTED) != 0};
    if (!total_hits) return run_fused(ix, d_qc, d_qs, d_qe, nq, 0, 0, out, ws.ptr, ws.bytes, ws.ep, s);
    {
        st = run_fused_sync(ix, d_qc, d_qs, d_qe, nq, 0, 0, out, ws.ptr, ws.bytes, ws.ep, s, total_hits);
        if (st) {
            ws.ep = ScanEpoch();  // force a clean workspace next time
            return st;
        }
        if (d_ids && *total_hits > ids_capacity)
            return fail(GTARS_ERR_CAPACITY, "ids buffer too small: need " + std::to_string(*total_hits));
    }
    return GTARS_OK;
}

gtars_status gtars_debug_occupy_device(void *stream, uint32_t workgroups, uint32_t lds_bytes, uint32_t microseconds) {
    gtars_status st = require_device();
    if (st) return st;
    if (lds_bytes > 160 * 1024) return fail(GTARS_ERR_INVALID_ARG, "lds_bytes > 160 KB");
    return launch_occupy(workgroups, lds_bytes, microseconds, (hipStream_t)stream);
}

gtars_status gtars_histogram_u32_device(const uint32_t *d_ids, uint64_t n, uint32_t n_bins, uint32_t *d_bins, void *stream) {
    if ((n && !d_ids) || (n_bins && !d_bins)) return fail(GTARS_ERR_INVALID_ARG, "NULL argument");
    gtars_status st = require_device();
    if (st) return st;
    return launch_hist_u32(d_ids, n, n_bins, d_bins, (hipStream_t)stream);
}

gtars_status gtars_histogram_rows_device(const uint64_t *d_offsets, const uint32_t *d_ids, const uint32_t *d_row, uint64_t nq,
                                         uint32_t row0, uint32_t n_rows, uint32_t n_cols, uint32_t *d_mat, void *stream) {
    if (nq && (!d_offsets || !d_row)) return fail(GTARS_ERR_INVALID_ARG, "NULL argument");
    if ((uint64_t)n_rows * n_cols && !d_mat) return fail(GTARS_ERR_INVALID_ARG, "d_mat is NULL");
    gtars_status st = require_device();
    if (st) return st;
    return launch_hist_rows(d_offsets, d_ids, d_row, nq, row0, n_rows, n_cols, d_mat, (hipStream_t)stream);
}

static gtars_status fill_device(const gtars_index_t *ix, const uint32_t *d_qc, const uint32_t *d_qs, const uint32_t *d_qe, uint64_t nq,
                                const uint64_t *d_offsets, uint32_t *d_ids, u64 capacity, int hint, void *stream);

gtars_status gtars_fill_device(const gtars_index_t *ix, const uint32_t *d_qc, const uint32_t *d_qs,
                               const uint32_t *d_qe, uint64_t nq, const uint64_t *d_offsets,
                               uint32_t *d_ids, void *stream) {
    // (no total: the caller's buffer holds offsets[nq] ids by contract; the build with the run form, as for any unbounded buffer)
    return fill_device(ix, d_qc, d_qs, d_qe, nq, d_offsets, d_ids, 1ull << 62, GTARS_TOK_AUTO, stream);
}

gtars_status gtars_fill_device_n(const gtars_index_t *ix, const uint32_t *d_qc, const uint32_t *d_qs,
                                 const uint32_t *d_qe, uint64_t nq, const uint64_t *d_offsets,
                                 uint32_t *d_ids, uint64_t total_hits, void *stream) {
    // what the sizing pass measured decides the build: fewer than four ids per query is not a hit-heavy batch
    const int hint = total_hits / 4 >= nq ? GTARS_TOK_WIDE : GTARS_TOK_NARROW;
    return fill_device(ix, d_qc, d_qs, d_qe, nq, d_offsets, d_ids, std::max<u64>(total_hits, 1), hint, stream);
}

static gtars_status fill_device(const gtars_index_t *ix, const uint32_t *d_qc, const uint32_t *d_qs, const uint32_t *d_qe, uint64_t nq,
                                const uint64_t *d_offsets, uint32_t *d_ids, u64 capacity, int hint, void *stream) {
    GT_SAME_DEVICE_AS(ix);
    gtars_status st = check_query_args(ix, d_qc, d_qs, d_qe, nq);
    if (st) return st;
    if (!d_offsets || !d_ids) return fail(GTARS_ERR_INVALID_ARG, "NULL output");
    hipStream_t s = (hipStream_t)stream;
    if (use_lds_path(ix) && nq) {
        // the fused tokenizer once more, its offsets into scratch (they equal the caller's): 16.5 us per 1M C2 queries, and a
        // hit-heavy batch leaves by wave-wide stores, where the fill kernel walks every query's hits again lane by lane.  The
        // caller's buffer holds offsets[nq] ids by contract: no capacity to watch.
        Workspace &ws = tls_workspace(0, s);
        gtars_status st2 = ws.reserve(fused_ws_bytes(ix, nq));
        if (st2) return st2;
        Workspace &tmp = tls_workspace(4, s);
        st2 = tmp.reserve((size_t)(nq + 1) * sizeof(u64));
        if (st2) return st2;
        EnumOut out{(u64 *)tmp.ptr, d_ids, nullptr, nullptr, capacity, hint};
        return run_fused(ix, d_qc, d_qs, d_qe, nq, 0, 0, out, ws.ptr, ws.bytes, ws.ep, s);
    }
    return launch_fill(ix->view(), ix->kind, d_qc, d_qs, d_qe, nq, 0, 0, d_offsets, d_ids, nullptr, nullptr, s);
}

gtars_status gtars_count_overlaps_device(const gtars_index_t *ix, const uint32_t *d_qc,
                                         const uint32_t *d_qs, const uint32_t *d_qe, uint64_t nq,
                                         int has_min, int32_t min_overlap, uint32_t *d_counts,
                                         void *stream) {
    GT_SAME_DEVICE_AS(ix);
    gtars_status st = check_query_args(ix, d_qc, d_qs, d_qe, nq);
    if (st) return st;
    if (nq && !d_counts) return fail(GTARS_ERR_INVALID_ARG, "d_counts is NULL");
    return count_dispatch(ix, d_qc, d_qs, d_qe, nq, has_min, min_overlap, d_counts, nullptr,
                        (hipStream_t)stream);
}

}  // extern "C"

// host-pointer helpers -------------------------------------------------------

struct DevQueries {
    ScopedDev buf;
    u32 *c = nullptr, *s = nullptr, *e = nullptr;
    gtars_status upload(const u32 *qc, const u32 *qs, const u32 *qe, u64 nq) {
        const size_t pad = ((size_t)nq * 4 + 255) & ~(size_t)255;
        gtars_status st = buf.alloc(pad * 3);
        if (st) return st;
        c = (u32 *)buf.p;
        s = (u32 *)((char *)buf.p + pad);
        e = (u32 *)((char *)buf.p + 2 * pad);
        if (nq) {
            GT_HIP(hipMemcpy(c, qc, nq * 4, hipMemcpyHostToDevice));
            GT_HIP(hipMemcpy(s, qs, nq * 4, hipMemcpyHostToDevice));
            GT_HIP(hipMemcpy(e, qe, nq * 4, hipMemcpyHostToDevice));
        }
        return GTARS_OK;
    }
};

template <class T>
static T *host_alloc(u64 n) {
    return (T *)malloc(std::max<u64>(n, 1) * sizeof(T));
}

// fused enumerate into library-allocated host arrays
static gtars_status enumerate_to_host(const gtars_index_t *ix, const u32 *qc, const u32 *qs, const u32 *qe,
                                      u64 nq, int has_min, i32 min_overlap, u64 *offsets, u32 **out_val,
                                      u32 **out_start, u32 **out_end, u64 *out_n) {
    gtars_status st = require_device();
    if (st) return st;
    DevQueries q;
    st = q.upload(qc, qs, qe, nq);
    if (st) return st;
    ScopedDev d_off, d_ws;
    st = d_off.alloc((nq + 1) * 8);
    if (st) return st;
    const size_t wsb = fused_ws_bytes(ix, nq);
    st = d_ws.alloc(wsb);
    if (st) return st;
    // pass 1: offsets + total only (no payload buffers)
    EnumOut o1{d_off.as<u64>(), nullptr, nullptr, nullptr, 0};
    ScanEpoch ep;
    u64 h = 0;
    st = run_fused_sync(ix, q.c, q.s, q.e, nq, has_min, min_overlap, o1, d_ws.p, wsb, ep, nullptr, &h);
    if (st) return st;
    if (offsets) GT_HIP(hipMemcpy(offsets, d_off.p, (nq + 1) * 8, hipMemcpyDeviceToHost));
    if (out_n) *out_n = h;
    const int nout = (out_val ? 1 : 0) + (out_start ? 1 : 0) + (out_end ? 1 : 0);
    if (nout) {
        ScopedDev d_out;
        const size_t pad = ((size_t)h * 4 + 255) & ~(size_t)255;
        st = d_out.alloc(pad * 3);
        if (st) return st;
        u32 *dv = out_val ? (u32 *)d_out.p : nullptr;
        u32 *ds = out_start ? (u32 *)((char *)d_out.p + pad) : nullptr;
        u32 *de = out_end ? (u32 *)((char *)d_out.p + 2 * pad) : nullptr;
        if (use_lds_path(ix)) {
            // the fused tokenizer once more, this time emitting the hits' stored positions (reference order, Bits or
            // AIList), and one gather of the payload columns by position -- instead of the generic per-query fill pass
            ScopedDev d_pos;
            if ((st = d_pos.alloc(pad))) return st;
            EnumOut o2{d_off.as<u64>(), d_pos.as<u32>(), nullptr, nullptr, h};
            st = launch_tokenize_lds(ix->accel_pos(), q.c, q.s, q.e, nq, has_min, min_overlap, o2, d_ws.p, wsb, ep, nullptr, nullptr,
                                     nullptr, ix->kind == GTARS_KIND_AILIST);
            if (st) return st;
            st = launch_gather_hits(ix->view(), d_pos.as<u32>(), h, dv, ds, de, nullptr);
            if (st) return st;
            GT_HIP(hipDeviceSynchronize());
        } else {
            st = launch_fill(ix->view(), ix->kind, q.c, q.s, q.e, nq, has_min, min_overlap, d_off.as<u64>(), dv, ds,
                             de, nullptr);
            if (st) return st;
            GT_HIP(hipDeviceSynchronize());
        }
        auto fetch = [&](u32 **dst, u32 *src) -> gtars_status {
            if (!dst) return GTARS_OK;
            *dst = host_alloc<u32>(h);
            if (!*dst) return fail(GTARS_ERR_INTERNAL, "out of host memory");
            if (h) GT_HIP(hipMemcpy(*dst, src, h * 4, hipMemcpyDeviceToHost));
            return GTARS_OK;
        };
        if ((st = fetch(out_val, dv))) return st;
        if ((st = fetch(out_start, ds))) return st;
        if ((st = fetch(out_end, de))) return st;
    }
    return GTARS_OK;
}

// ---------------------------------------------------------------- host pipeline
// The host-pointer tokenizer as a stream: per calling thread (and device) one set of device buffers, two streams and a
// helper thread, all kept across calls.  A batch is cut into chunks; the calling thread copies chunk k+1 to the device
// while the kernel of chunk k runs and the helper thread copies the results of chunk k-1 back (copies from and to
// pageable memory block the thread that issues them, hence two threads for the two directions of the link).
// Chunks chain on the device: chunk k starts its offsets at the running total chunk k-1 left in d_chain[k].
namespace {
class HelperThread {
public:
    ~HelperThread() { stop(); }
    void post(std::function<void()> job) {
        {
            std::lock_guard<std::mutex> g(mu_);
            if (!started_) {
                started_ = true;
                th_ = std::thread([this] { run(); });
            }
            jobs_.push_back(std::move(job));
            ++pending_;
        }
        cv_.notify_one();
    }
    void wait_idle() {
        std::unique_lock<std::mutex> g(mu_);
        idle_.wait(g, [this] { return pending_ == 0; });
    }
    void stop() {
        {
            std::lock_guard<std::mutex> g(mu_);
            if (!started_ || quit_) return;
            quit_ = true;
        }
        cv_.notify_one();
        if (th_.joinable()) th_.join();
    }

private:
    void run() {
        for (;;) {
            std::function<void()> job;
            {
                std::unique_lock<std::mutex> g(mu_);
                cv_.wait(g, [this] { return quit_ || !jobs_.empty(); });
                if (jobs_.empty()) return;
                job = std::move(jobs_.front());
                jobs_.pop_front();
            }
            job();
            {
                std::lock_guard<std::mutex> g(mu_);
                --pending_;
            }
            idle_.notify_all();
        }
    }
    std::mutex mu_;
    std::condition_variable cv_, idle_;
    std::deque<std::function<void()>> jobs_;
    std::thread th_;
    size_t pending_ = 0;
    bool started_ = false, quit_ = false;
};

constexpr int PIPE_MAX_CHUNKS = 16;
struct HostPipe {
    int device = -1;
    void *d_buf = nullptr;
    size_t d_bytes = 0;
    hipStream_t s_in = nullptr, s_comp = nullptr, s_out = nullptr;
    hipEvent_t ev_in[PIPE_MAX_CHUNKS] = {}, ev_done[PIPE_MAX_CHUNKS] = {};
    u64 *d_chain = nullptr;  // [PIPE_MAX_CHUNKS + 1] running totals
    u64 *h_chain = nullptr;  // pinned mirror
    HelperThread helper;
    gtars_status init() {
        int dev = 0;
        GT_HIP(hipGetDevice(&dev));
        if (device == dev) return GTARS_OK;
        GT_HIP(hipStreamCreateWithFlags(&s_in, hipStreamNonBlocking));
        GT_HIP(hipStreamCreateWithFlags(&s_comp, hipStreamNonBlocking));
        GT_HIP(hipStreamCreateWithFlags(&s_out, hipStreamNonBlocking));
        for (int k = 0; k < PIPE_MAX_CHUNKS; ++k) {
            GT_HIP(hipEventCreateWithFlags(&ev_in[k], hipEventDisableTiming));
            GT_HIP(hipEventCreateWithFlags(&ev_done[k], hipEventDisableTiming));
        }
        GT_HIP(hipMalloc((void **)&d_chain, sizeof(u64) * (PIPE_MAX_CHUNKS + 1)));
        GT_HIP(hipHostMalloc((void **)&h_chain, sizeof(u64) * (PIPE_MAX_CHUNKS + 1), hipHostMallocDefault));
        device = dev;
        return GTARS_OK;
    }
    gtars_status reserve(size_t need) {
        if (d_bytes >= need) return GTARS_OK;
        if (d_buf) {
            (void)hipDeviceSynchronize();
            (void)hipFree(d_buf);
            d_buf = nullptr;
            d_bytes = 0;
        }
        const size_t want = (need + (need >> 2) + (1 << 20)) & ~(size_t)((1 << 20) - 1);
        GT_HIP(hipMalloc(&d_buf, want));
        d_bytes = want;
        return GTARS_OK;
    }
};
// one pipe per (calling thread, device): a thread that switches devices between calls gets that device's pipe.  The pipes of a
// thread that exits go back to a process-wide pool -- device buffer, pinned words, streams, events and the (idle) helper thread
// included -- from which new threads take theirs, like the workspaces above: a caller that runs every batch on a short-lived
// thread of its own (fragsplit_tokenize_core's wave tokenizer did) used to leave all of that behind on every call.
struct HostPipePool {
    std::mutex mu;
    std::vector<HostPipe *> idle;
};
HostPipePool &host_pipe_pool() {
    static HostPipePool *p = new HostPipePool();  // never destroyed: thread-exit handlers may run late
    return *p;
}
struct TlsHostPipes {
    std::map<int, HostPipe *> pipes;
    ~TlsHostPipes() {
        HostPipePool &pool = host_pipe_pool();
        std::lock_guard<std::mutex> g(pool.mu);
        for (auto &kv : pipes)
            if (kv.second) pool.idle.push_back(kv.second);
    }
};
HostPipe &tls_host_pipe() {
    static thread_local TlsHostPipes tls;
    int dev = 0;
    (void)hipGetDevice(&dev);
    HostPipe *&slot = tls.pipes[dev];
    if (!slot) {
        HostPipePool &pool = host_pipe_pool();
        std::lock_guard<std::mutex> g(pool.mu);
        for (size_t i = 0; i < pool.idle.size(); ++i)
            if (pool.idle[i]->device == dev) {
                slot = pool.idle[i];
                pool.idle[i] = pool.idle.back();
                pool.idle.pop_back();
                break;
            }
    }
    if (!slot) slot = new HostPipe();
    return *slot;
}
// A batch that needed more than this keeps only this much device memory afterwards (a 1e9-query batch is 22 GB of
// staging; the next 1M-query call should not find a thread holding on to it).  GTARS_PIPE_KEEP_MB overrides.
size_t pipe_keep_bytes() {
    static const size_t keep = [] {
        const char *e = cfg_get("GTARS_PIPE_KEEP_MB");
        return (size_t)(e && *e ? atoll(e) : 1024) << 20;
    }();
    return keep;
}
}  // namespace

static gtars_status hip_try(hipError_t e, const char *what) {
    return e == hipSuccess ? GTARS_OK : hip_fail(e, what, __FILE__, __LINE__);
}

// offsets[nq + 1] and up to ids_capacity ids into caller memory; *out_n = hits.  ids may be null (offsets only).
static gtars_status tokenize_pipeline(const gtars_index_t *ix, const u32 *qc, const u32 *qs, const u32 *qe, u64 nq, u64 *offsets,
                                      u32 *ids, u64 ids_capacity, u64 *out_n) {
    // HIP's current device belongs to the host THREAD and starts at 0: a helper thread of the caller (the fragment pipeline's
    // wave tokenizer) would otherwise stage and launch on device 0 with an index that lives on device k
    DeviceScope on_index_device(ix->device);
    if (on_index_device.st) return on_index_device.st;
    HostPipe &hp = tls_host_pipe();
    gtars_status st = hp.init();
    if (st) return st;
    if (nq == 0) {
        if (offsets) offsets[0] = 0;
        *out_n = 0;
        return GTARS_OK;
    }
    // chunks: multiples of the kernel's tile, at least 256k queries each; chained launches need the LDS kernel
    const bool chain = use_lds_path(ix);
    int n_chunks = 1;
    // (a chunk costs ~25 us of fixed copy / launch / hand-over overhead per direction: 1M queries in one chunk 512 us,
    // in 2 / 4 / 8 chunks 564 / 745 / 1231 us -- chunks pay from ~4M queries each)
    if (chain) n_chunks = (int)std::min<u64>(PIPE_MAX_CHUNKS, std::max<u64>(1, nq / (4u << 20)));
    if (const char *e = cfg_get("GTARS_PIPE_CHUNKS")) n_chunks = std::max(1, std::min(chain ? PIPE_MAX_CHUNKS : 1, atoi(e)));
    const u64 chunk = ((nq + n_chunks - 1) / n_chunks + 4095) / 4096 * 4096;
    n_chunks = (int)((nq + chunk - 1) / chunk);
    // device layout: qc | qs | qe | offsets | ids (capacity: what the caller can take, at least a typical batch's hits)
    const u64 cap_dev = ids ? std::max<u64>(ids_capacity, 1) : 0;
    const size_t qpad = ((size_t)nq * 4 + 255) & ~(size_t)255;
    const size_t opad = ((size_t)(nq + 1) * 8 + 255) & ~(size_t)255;
    st = hp.reserve(3 * qpad + opad + (size_t)cap_dev * 4 + 256);
    if (st) return st;
    struct Trim {  // on every way out: an unusually large batch does not leave its staging memory with the thread
        HostPipe &hp;
        ~Trim() {
            if (hp.d_bytes <= pipe_keep_bytes()) return;
            (void)hipStreamSynchronize(hp.s_in);
            (void)hipStreamSynchronize(hp.s_comp);
            (void)hipStreamSynchronize(hp.s_out);
            (void)hipFree(hp.d_buf);
            hp.d_buf = nullptr;
            hp.d_bytes = 0;
        }
    } trim{hp};
    u32 *d_qc = (u32 *)hp.d_buf, *d_qs = (u32 *)((char *)hp.d_buf + qpad), *d_qe = (u32 *)((char *)hp.d_buf + 2 * qpad);
    u64 *d_off = (u64 *)((char *)hp.d_buf + 3 * qpad);
    u32 *d_ids = ids ? (u32 *)((char *)hp.d_buf + 3 * qpad + opad) : nullptr;
    Workspace &ws = tls_workspace(0, hp.s_comp);
    st = ws.reserve(fused_ws_bytes(ix, chunk));
    if (st) return st;
    if (n_chunks == 1) {
        // one chunk: everything in order on one stream from the calling thread; the batch's total is the last offset
        GT_HIP(hipMemcpyAsync(d_qc, qc, nq * 4, hipMemcpyHostToDevice, hp.s_comp));
        GT_HIP(hipMemcpyAsync(d_qs, qs, nq * 4, hipMemcpyHostToDevice, hp.s_comp));
        GT_HIP(hipMemcpyAsync(d_qe, qe, nq * 4, hipMemcpyHostToDevice, hp.s_comp));
        EnumOut out{d_off, d_ids, nullptr, nullptr, cap_dev};
        // The LDS tokenizer cannot time out (its look-back counts a missing tile itself).  The generic kernel can: its
        // ScanHead.err is read back before offsets[nq] is trusted, and a timed-out batch is redone (run_fused_sync).
        if (chain) {
            st = run_fused(ix, d_qc, d_qs, d_qe, nq, 0, 0, out, ws.ptr, ws.bytes, ws.ep, hp.s_comp);
        } else {
            u64 scan_total = 0;
            st = run_fused_sync(ix, d_qc, d_qs, d_qe, nq, 0, 0, out, ws.ptr, ws.bytes, ws.ep, hp.s_comp, &scan_total);
        }
        if (st) {
            ws.ep = ScanEpoch();
            return st;
        }
        u64 total = 0;
        if (offsets) {
            GT_HIP(hipMemcpyAsync(offsets, d_off, (nq + 1) * 8, hipMemcpyDeviceToHost, hp.s_comp));
            GT_HIP(hipStreamSynchronize(hp.s_comp));
            total = offsets[nq];
        } else {
            GT_HIP(hipMemcpyAsync(&total, d_off + nq, 8, hipMemcpyDeviceToHost, hp.s_comp));
            GT_HIP(hipStreamSynchronize(hp.s_comp));
        }
        *out_n = total;
        if (ids && total) {
            GT_HIP(hipMemcpyAsync(ids, d_ids, std::min<u64>(total, ids_capacity) * 4, hipMemcpyDeviceToHost, hp.s_comp));
            GT_HIP(hipStreamSynchronize(hp.s_comp));
        }
        if (ids && total > ids_capacity) return fail(GTARS_ERR_CAPACITY, "ids buffer too small: need " + std::to_string(total));
        return GTARS_OK;
    }
    GT_HIP(hipMemsetAsync(hp.d_chain, 0, sizeof(u64), hp.s_comp));
    hp.h_chain[0] = 0;
    gtars_status job_status[PIPE_MAX_CHUNKS];
    for (int k = 0; k < n_chunks; ++k) job_status[k] = GTARS_OK;
    for (int k = 0; k < n_chunks; ++k) {
        const u64 q0 = (u64)k * chunk, n_k = std::min<u64>(chunk, nq - q0);
        // no early return inside this loop: the helper's jobs hold pointers into this frame (job_status) and into the
        // caller's arrays, so every exit goes through wait_idle() below
        if ((st = hip_try(hipMemcpyAsync(d_qc + q0, qc + q0, n_k * 4, hipMemcpyHostToDevice, hp.s_in), "H2D chrom")) ||
            (st = hip_try(hipMemcpyAsync(d_qs + q0, qs + q0, n_k * 4, hipMemcpyHostToDevice, hp.s_in), "H2D start")) ||
            (st = hip_try(hipMemcpyAsync(d_qe + q0, qe + q0, n_k * 4, hipMemcpyHostToDevice, hp.s_in), "H2D end")) ||
            (st = hip_try(hipEventRecord(hp.ev_in[k], hp.s_in), "record H2D event")) ||
            (st = hip_try(hipStreamWaitEvent(hp.s_comp, hp.ev_in[k], 0), "wait H2D event")))
            break;
        EnumOut out{d_off + q0, d_ids, nullptr, nullptr, cap_dev};
        if (chain)
            st = launch_tokenize_lds(ix->accel(), d_qc + q0, d_qs + q0, d_qe + q0, n_k, 0, 0, out, ws.ptr, ws.bytes, ws.ep, hp.s_comp,
                                     hp.d_chain + k, hp.d_chain + k + 1, ix->kind == GTARS_KIND_AILIST);
        else
            st = run_fused(ix, d_qc, d_qs, d_qe, nq, 0, 0, out, ws.ptr, ws.bytes, ws.ep, hp.s_comp);
        if (st) break;
        if (chain)
            st = hip_try(hipMemcpyAsync(hp.h_chain + k + 1, hp.d_chain + k + 1, sizeof(u64), hipMemcpyDeviceToHost, hp.s_comp), "D2H total");
        else
            st = hip_try(hipMemcpyAsync(hp.h_chain + 1, &((ScanHead *)ws.ptr)->total, sizeof(u64), hipMemcpyDeviceToHost, hp.s_comp),
                         "D2H total");
        if (st || (st = hip_try(hipEventRecord(hp.ev_done[k], hp.s_comp), "record done event"))) break;
        const bool last = k + 1 == n_chunks;
        const int dev = hp.device;
        HostPipe *php = &hp;
        gtars_status *js = &job_status[k];
        hp.helper.post([=]() {
            // results of chunk k: its offsets (the batch's last offset comes with the last chunk) and its ids
            if (hipSetDevice(dev) != hipSuccess || hipEventSynchronize(php->ev_done[k]) != hipSuccess) {
                *js = GTARS_ERR_HIP;
                return;
            }
            const u64 lo = php->h_chain[k], hi = php->h_chain[k + 1];
            hipError_t e = hipSuccess;
            if (offsets) e = hipMemcpyAsync(offsets + q0, d_off + q0, (n_k + (last ? 1 : 0)) * 8, hipMemcpyDeviceToHost, php->s_out);
            if (e == hipSuccess && ids && hi > lo && lo < ids_capacity)
                e = hipMemcpyAsync(ids + lo, d_ids + lo, (std::min<u64>(hi, ids_capacity) - lo) * 4, hipMemcpyDeviceToHost, php->s_out);
            if (e == hipSuccess) e = hipStreamSynchronize(php->s_out);
            if (e != hipSuccess) *js = GTARS_ERR_HIP;
        });
    }
    hp.helper.wait_idle();
    if (st) {
        (void)hipStreamSynchronize(hp.s_comp);
        ws.ep = ScanEpoch();
        return st;
    }
    for (int k = 0; k < n_chunks; ++k)
        if (job_status[k]) return fail(GTARS_ERR_HIP, "host pipeline: device-to-host copy failed");
    const u64 total = hp.h_chain[chain ? n_chunks : 1];
    *out_n = total;
    if (ids && total > ids_capacity) return fail(GTARS_ERR_CAPACITY, "ids buffer too small: need " + std::to_string(total));
    return GTARS_OK;
}

extern "C" {

static gtars_status gtars_tokenize_into_impl(const gtars_index_t *ix, const uint32_t *qc, const uint32_t *qs, const uint32_t *qe,
                                 uint64_t nq, uint64_t *offsets, uint32_t *ids, uint64_t ids_capacity, uint64_t *out_n) {
    gtars_status st = check_query_args(ix, qc, qs, qe, nq);
    if (st) return st;
    if (!offsets || !out_n || (ids_capacity && !ids)) return fail(GTARS_ERR_INVALID_ARG, "NULL output");
    *out_n = 0;
    st = require_device();
    if (st) return st;
    return tokenize_pipeline(ix, qc, qs, qe, nq, offsets, ids_capacity ? ids : nullptr, ids_capacity, out_n);
}

static gtars_status gtars_tokenize_impl(const gtars_index_t *ix, const uint32_t *qc, const uint32_t *qs,
                            const uint32_t *qe, uint64_t nq, uint64_t *offsets, uint32_t **out_ids,
                            uint64_t *out_n) {
    gtars_status st = check_query_args(ix, qc, qs, qe, nq);
    if (st) return st;
    if (!out_ids || !out_n) return fail(GTARS_ERR_INVALID_ARG, "NULL output");
    *out_ids = nullptr;
    *out_n = 0;
    st = require_device();
    if (st) return st;
    // the streaming pipeline with a guessed capacity; a second (ids only) pass on overflow
    u64 cap = nq * 2 + 1024, total = 0;
    u32 *ids = host_alloc<u32>(cap);
    if (!ids) return fail(GTARS_ERR_INTERNAL, "out of host memory");
    std::vector<u64> tmp_off;
    u64 *off = offsets;
    if (!off) {
        tmp_off.resize(nq + 1);
        off = tmp_off.data();
    }
    st = tokenize_pipeline(ix, qc, qs, qe, nq, off, ids, cap, &total);
    if (st == GTARS_ERR_CAPACITY) {
        free(ids);
        cap = total;
        ids = host_alloc<u32>(cap);
        if (!ids) return fail(GTARS_ERR_INTERNAL, "out of host memory");
        st = tokenize_pipeline(ix, qc, qs, qe, nq, off, ids, cap, &total);
    }
    if (st) {
        free(ids);
        return st;
    }
    *out_ids = ids;
    *out_n = total;
    return GTARS_OK;
}

gtars_status gtars_count_overlaps(const gtars_index_t *ix, const uint32_t *qc, const uint32_t *qs,
                                  const uint32_t *qe, uint64_t nq, int has_min, int32_t min_overlap,
                                  uint32_t *counts) {
    GT_ON_DEVICE_OF(ix);
    gtars_status st = check_query_args(ix, qc, qs, qe, nq);
    if (st) return st;
    if (nq && !counts) return fail(GTARS_ERR_INVALID_ARG, "counts is NULL");
    st = require_device();
    if (st) return st;
    if (!nq) return GTARS_OK;
    DevQueries q;
    st = q.upload(qc, qs, qe, nq);
    if (st) return st;
    ScopedDev d;
    st = d.alloc(nq * 4);
    if (st) return st;
    st = count_dispatch(ix, q.c, q.s, q.e, nq, has_min, min_overlap, d.as<u32>(), nullptr, nullptr);
    if (st) return st;
    GT_HIP(hipMemcpy(counts, d.p, nq * 4, hipMemcpyDeviceToHost));
    return GTARS_OK;
}

static gtars_status bits_count_prepare(const gtars_index_t *ix) {
    if (ix->kind != GTARS_KIND_BITS) return fail(GTARS_ERR_INVALID_ARG, "Bits::count needs a Bits-kind index");
    std::lock_guard<std::mutex> lk(ix->ends_mu);
    if (ix->ends_ready) return GTARS_OK;
    std::vector<u32> e(ix->h_ends);
    for (u32 c = 0; c < ix->n_chrom; ++c) std::sort(e.begin() + ix->h_chrom_off[c], e.begin() + ix->h_chrom_off[c + 1]);
    gtars_status st = ix->ends_sorted.upload(e);
    if (st) return st;
    ix->ends_ready = true;
    return GTARS_OK;
}

gtars_status gtars_bits_count_device(const gtars_index_t *ix, const uint32_t *d_qc, const uint32_t *d_qs,
                                     const uint32_t *d_qe, uint64_t nq, uint64_t *d_counts, void *stream) {
    GT_SAME_DEVICE_AS(ix);
    if (!ix) return fail(GTARS_ERR_INVALID_ARG, "index is NULL");
    if (nq && (!d_qc || !d_qs || !d_qe || !d_counts)) return fail(GTARS_ERR_INVALID_ARG, "NULL device pointer");
    gtars_status st = require_device();
    if (st) return st;
    if ((st = bits_count_prepare(ix))) return st;
    return launch_bits_count(ix->view(), ix->ends_sorted.p, d_qc, d_qs, d_qe, nq, (u64 *)d_counts, (hipStream_t)stream);
}

gtars_status gtars_bits_count(const gtars_index_t *ix, const uint32_t *qc, const uint32_t *qs, const uint32_t *qe,
                              uint64_t nq, uint64_t *counts) {
    GT_ON_DEVICE_OF(ix);
    gtars_status st = check_query_args(ix, qc, qs, qe, nq);
    if (st) return st;
    if (nq && !counts) return fail(GTARS_ERR_INVALID_ARG, "counts is NULL");
    st = require_device();
    if (st) return st;
    if ((st = bits_count_prepare(ix))) return st;
    if (!nq) return GTARS_OK;
    DevQueries q;
    st = q.upload(qc, qs, qe, nq);
    if (st) return st;
    ScopedDev d;
    st = d.alloc(nq * 8);
    if (st) return st;
    st = launch_bits_count(ix->view(), ix->ends_sorted.p, q.c, q.s, q.e, nq, d.as<u64>(), nullptr);
    if (st) return st;
    GT_HIP(hipMemcpy(counts, d.p, nq * 8, hipMemcpyDeviceToHost));
    return GTARS_OK;
}

gtars_status gtars_any_overlaps(const gtars_index_t *ix, const uint32_t *qc, const uint32_t *qs,
                                const uint32_t *qe, uint64_t nq, int has_min, int32_t min_overlap,
                                uint8_t *out) {
    GT_ON_DEVICE_OF(ix);
    gtars_status st = check_query_args(ix, qc, qs, qe, nq);
    if (st) return st;
    if (nq && !out) return fail(GTARS_ERR_INVALID_ARG, "out is NULL");
    st = require_device();
    if (st) return st;
    if (!nq) return GTARS_OK;
    DevQueries q;
    st = q.upload(qc, qs, qe, nq);
    if (st) return st;
    ScopedDev d;
    st = d.alloc(nq);
    if (st) return st;
    st = count_dispatch(ix, q.c, q.s, q.e, nq, has_min, min_overlap, nullptr, d.as<u8>(), nullptr);
    if (st) return st;
    GT_HIP(hipMemcpy(out, d.p, nq, hipMemcpyDeviceToHost));
    return GTARS_OK;
}

gtars_status gtars_find_overlaps(const gtars_index_t *ix, const uint32_t *qc, const uint32_t *qs,
                                 const uint32_t *qe, uint64_t nq, int has_min, int32_t min_overlap,
                                 uint64_t *offsets, uint32_t **out_start, uint32_t **out_end,
                                 uint32_t **out_val, uint64_t *out_n) {
    GT_ON_DEVICE_OF(ix);
    gtars_status st = check_query_args(ix, qc, qs, qe, nq);
    if (st) return st;
    if (out_start) *out_start = nullptr;
    if (out_end) *out_end = nullptr;
    if (out_val) *out_val = nullptr;
    return enumerate_to_host(ix, qc, qs, qe, nq, has_min, min_overlap, offsets, out_val, out_start, out_end,
                             out_n);
}

gtars_status gtars_find_overlap_indices(const gtars_index_t *ix, const uint32_t *qc, const uint32_t *qs,
                                        const uint32_t *qe, uint64_t nq, int has_min, int32_t min_overlap,
                                        uint64_t *offsets, uint32_t **out_idx, uint64_t *out_n) {
    GT_ON_DEVICE_OF(ix);
    gtars_status st = check_query_args(ix, qc, qs, qe, nq);
    if (st) return st;
    if (!out_idx || !out_n || !offsets) return fail(GTARS_ERR_INVALID_ARG, "NULL output");
    *out_idx = nullptr;
    *out_n = 0;
    st = require_device();
    if (st) return st;
    if (lds_target(ix) == ix->flat && ix->flat) ix = ix->flat;  // (sorted unique source rows: the same from either order)
    // Every source row that shares a hit's coordinates is itself a hit (same
    // overlap, same filter), so "all rows sharing coordinates, sorted, dedup"
    // (indexed_region_set.rs:246-263) == the hit source indices, sorted.
    DevQueries q;
    st = q.upload(qc, qs, qe, nq);
    if (st) return st;
    ScopedDev d_off, d_ws, d_cnt, d_off2;
    if ((st = d_off.alloc((nq + 1) * 8))) return st;
    if ((st = d_off2.alloc((nq + 1) * 8))) return st;
    if ((st = d_cnt.alloc(nq * 4))) return st;
    const size_t wsb = std::max(fused_ws_bytes(ix, nq), scan_ws_bytes(nq));
    if ((st = d_ws.alloc(wsb))) return st;
    EnumOut o1{d_off.as<u64>(), nullptr, nullptr, nullptr, 0};
    ScanEpoch ep;
    u64 h = 0;
    st = run_fused_sync(ix, q.c, q.s, q.e, nq, has_min, min_overlap, o1, d_ws.p, wsb, ep, nullptr, &h);
    if (st) return st;
    ScopedDev d_val;
    if ((st = d_val.alloc(h * 4))) return st;
    if (use_lds_path(ix)) {  // the fused tokenizer writes the source indices itself (the order inside a query does not matter here)
        EnumOut o2{d_off.as<u64>(), d_val.as<u32>(), nullptr, nullptr, h};
        st = launch_tokenize_lds(ix->accel(), q.c, q.s, q.e, nq, has_min, min_overlap, o2, d_ws.p, wsb, ep, nullptr, nullptr, nullptr,
                                 ix->kind == GTARS_KIND_AILIST);
    } else {
        st = launch_fill(ix->view(), ix->kind, q.c, q.s, q.e, nq, has_min, min_overlap, d_off.as<u64>(),
                         d_val.as<u32>(), nullptr, nullptr, nullptr);
    }
    if (st) return st;
    st = launch_sort_unique_segments(d_val.as<u32>(), d_off.as<u64>(), nq, d_cnt.as<u32>(), nullptr);
    if (st) return st;
    st = launch_scan_u32_to_u64(d_cnt.as<u32>(), nq, d_off2.as<u64>(), d_ws.p, wsb, nullptr);
    if (st) return st;
    // compact on the host side of the copy: segments are already contiguous
    // when nothing was de-duplicated (the overwhelmingly common case)
    std::vector<u64> off1(nq + 1), off2(nq + 1);
    GT_HIP(hipMemcpy(off1.data(), d_off.p, (nq + 1) * 8, hipMemcpyDeviceToHost));
    GT_HIP(hipMemcpy(off2.data(), d_off2.p, (nq + 1) * 8, hipMemcpyDeviceToHost));
    std::vector<u32> vals(h);
    if (h) GT_HIP(hipMemcpy(vals.data(), d_val.p, h * 4, hipMemcpyDeviceToHost));
    const u64 h2 = off2[nq];
    u32 *res = host_alloc<u32>(h2);
    if (!res) return fail(GTARS_ERR_INTERNAL, "out of host memory");
    for (u64 qi = 0; qi < nq; ++qi) {
        const u64 len = off2[qi + 1] - off2[qi];
        if (len) memcpy(res + off2[qi], vals.data() + off1[qi], len * 4);
    }
    memcpy(offsets, off2.data(), (nq + 1) * 8);
    *out_idx = res;
    *out_n = h2;
    return GTARS_OK;
}

// ---- index-side subset (multi_chrom_overlapper.rs:449-478, indexed_region_set.rs:201-230) ----------------------

gtars_status gtars_mark_overlapped_device(const gtars_index_t *ix, const uint32_t *d_qc, const uint32_t *d_qs,
                                          const uint32_t *d_qe, uint64_t nq, int has_min, int32_t min_overlap,
                                          uint32_t *d_mark, void *stream) {
    GT_SAME_DEVICE_AS(ix);
    gtars_status st = check_query_args(ix, d_qc, d_qs, d_qe, nq);
    if (st) return st;
    if (!d_mark && ix->n) return fail(GTARS_ERR_INVALID_ARG, "d_mark is NULL");
    if ((st = require_device())) return st;
    const gtars_index *t = lds_target(ix);
    if (!t) return fail(GTARS_ERR_INVALID_ARG, "index has no blocked structure: use gtars_subset_by_overlaps");
    hipStream_t s = (hipStream_t)stream;
    const size_t mark_bytes = ((size_t)ix->n + 31) / 32 * 4;
    GT_HIP(hipMemsetAsync(d_mark, 0, mark_bytes, s));
    if (t == ix) return launch_mark_lds(ix->accel_pos(), d_qc, d_qs, d_qe, nq, has_min, min_overlap, d_mark, s);
    // nested AIList: marked by the flat companion's positions, then carried over to this index's stored positions
    Workspace &ws = tls_workspace(3, s);
    if ((st = ws.reserve(mark_bytes + 64))) return st;
    u32 *tmp = (u32 *)((char *)ws.ptr + 64);
    GT_HIP(hipMemsetAsync(tmp, 0, mark_bytes, s));
    if ((st = launch_mark_lds(t->accel_pos(), d_qc, d_qs, d_qe, nq, has_min, min_overlap, tmp, s))) return st;
    return launch_permute_marks(tmp, ix->flat_pos.p, ix->n, d_mark, s);
}

}  // extern "C"

// ascending stored positions hit by any query (blocked structure), or -- generic kernels -- the hits themselves
static gtars_status subset_positions(const gtars_index_t *ix, const u32 *qc, const u32 *qs, const u32 *qe, u64 nq, int has_min,
                                     i32 min_overlap, std::vector<u32> &pos) {
    pos.clear();
    if (!nq || !ix->n) return GTARS_OK;
    DevQueries q;
    gtars_status st = q.upload(qc, qs, qe, nq);
    if (st) return st;
    const size_t words = ((size_t)ix->n + 31) / 32;
    ScopedDev d_mark;
    if ((st = d_mark.alloc(words * 4))) return st;
    st = gtars_mark_overlapped_device(ix, q.c, q.s, q.e, nq, has_min, min_overlap, d_mark.as<u32>(), nullptr);
    if (st) return st;
    std::vector<u32> mark(words);
    GT_HIP(hipMemcpy(mark.data(), d_mark.p, words * 4, hipMemcpyDeviceToHost));
    for (size_t w = 0; w < words; ++w) {
        u32 m = mark[w];
        while (m) {
            pos.push_back((u32)(w * 32) + (u32)__builtin_ctz(m));
            m &= m - 1;
        }
    }
    return GTARS_OK;
}

extern "C" {

static gtars_status gtars_subset_by_overlaps_impl(const gtars_index_t *ix, const uint32_t *qc, const uint32_t *qs, const uint32_t *qe,
                                                  uint64_t nq, int has_min, int32_t min_overlap, uint32_t **out_chrom,
                                                  uint32_t **out_start, uint32_t **out_end, uint64_t *out_n) {
    GT_ON_DEVICE_OF(ix);
    gtars_status st = check_query_args(ix, qc, qs, qe, nq);
    if (st) return st;
    if (!out_chrom || !out_start || !out_end || !out_n) return fail(GTARS_ERR_INVALID_ARG, "NULL output");
    *out_chrom = *out_start = *out_end = nullptr;
    *out_n = 0;
    if ((st = require_device())) return st;
    struct Trip {
        u32 c, s, e;
        bool operator<(const Trip &o) const { return std::tie(c, s, e) < std::tie(o.c, o.s, o.e); }
        bool operator==(const Trip &o) const { return c == o.c && s == o.s && e == o.e; }
    };
    std::vector<Trip> hits;
    if (lds_target(ix) == ix->flat && ix->flat) ix = ix->flat;  // (a set of coordinates: the same from either stored order)
    if (use_lds_path(ix)) {
        std::vector<u32> pos;
        if ((st = subset_positions(ix, qc, qs, qe, nq, has_min, min_overlap, pos))) return st;
        hits.reserve(pos.size());
        u32 c = 0;
        for (u32 p : pos) {  // positions ascend: so do their chromosomes
            while (c + 1 < ix->h_chrom_off.size() && p >= ix->h_chrom_off[c + 1]) ++c;
            hits.push_back(Trip{c, ix->h_starts[p], ix->h_ends[p]});
        }
    } else {
        // generic kernels: the hits' coordinates per query (find_overlaps_regions), the set is formed here
        std::vector<u64> off(nq + 1);
        u32 *hs = nullptr, *he = nullptr;
        u64 h = 0;
        st = enumerate_to_host(ix, qc, qs, qe, nq, has_min, min_overlap, off.data(), nullptr, &hs, &he, &h);
        if (st) {
            free(hs);
            free(he);
            return st;
        }
        hits.reserve(h);
        for (u64 qi = 0; qi < nq; ++qi)
            for (u64 k = off[qi]; k < off[qi + 1]; ++k) hits.push_back(Trip{qc[qi], hs[k], he[k]});
        free(hs);
        free(he);
    }
    // BTreeSet<(chr, start, end)>: sorted, de-duplicated (Bits positions are already in this order; AIList's are not)
    if (!std::is_sorted(hits.begin(), hits.end())) std::sort(hits.begin(), hits.end());
    hits.erase(std::unique(hits.begin(), hits.end()), hits.end());
    const u64 n = hits.size();
    u32 *oc = host_alloc<u32>(n), *os = host_alloc<u32>(n), *oe = host_alloc<u32>(n);
    if (!oc || !os || !oe) {
        free(oc);
        free(os);
        free(oe);
        return fail(GTARS_ERR_INTERNAL, "out of host memory");
    }
    for (u64 i = 0; i < n; ++i) {
        oc[i] = hits[i].c;
        os[i] = hits[i].s;
        oe[i] = hits[i].e;
    }
    *out_chrom = oc;
    *out_start = os;
    *out_end = oe;
    *out_n = n;
    return GTARS_OK;
}

gtars_status gtars_subset_by_overlaps(const gtars_index_t *ix, const uint32_t *qc, const uint32_t *qs, const uint32_t *qe, uint64_t nq,
                                      int has_min, int32_t min_overlap, uint32_t **out_chrom, uint32_t **out_start,
                                      uint32_t **out_end, uint64_t *out_n) {
    return guarded([&] { return gtars_subset_by_overlaps_impl(ix, qc, qs, qe, nq, has_min, min_overlap, out_chrom, out_start, out_end, out_n); });
}

static gtars_status gtars_subset_source_indices_impl(const gtars_index_t *ix, const uint32_t *qc, const uint32_t *qs,
                                                     const uint32_t *qe, uint64_t nq, int has_min, int32_t min_overlap,
                                                     uint32_t **out_idx, uint64_t *out_n) {
    GT_ON_DEVICE_OF(ix);
    gtars_status st = check_query_args(ix, qc, qs, qe, nq);
    if (st) return st;
    if (!out_idx || !out_n) return fail(GTARS_ERR_INVALID_ARG, "NULL output");
    *out_idx = nullptr;
    *out_n = 0;
    if ((st = require_device())) return st;
    std::vector<u32> vals;
    if (lds_target(ix) == ix->flat && ix->flat) ix = ix->flat;  // (a set of source rows: the same from either stored order)
    if (use_lds_path(ix)) {
        std::vector<u32> pos;
        if ((st = subset_positions(ix, qc, qs, qe, nq, has_min, min_overlap, pos))) return st;
        vals.reserve(pos.size());
        for (u32 p : pos) vals.push_back(ix->h_vals[p]);
    } else {
        std::vector<u64> off(nq + 1);
        u32 *hv = nullptr;
        u64 h = 0;
        st = enumerate_to_host(ix, qc, qs, qe, nq, has_min, min_overlap, off.data(), &hv, nullptr, nullptr, &h);
        if (st) {
            free(hv);
            return st;
        }
        vals.assign(hv, hv + h);
        free(hv);
    }
    // BTreeSet<usize>: ascending, unique
    std::sort(vals.begin(), vals.end());
    vals.erase(std::unique(vals.begin(), vals.end()), vals.end());
    u32 *res = host_alloc<u32>(vals.size());
    if (!res) return fail(GTARS_ERR_INTERNAL, "out of host memory");
    if (!vals.empty()) memcpy(res, vals.data(), vals.size() * 4);
    *out_idx = res;
    *out_n = vals.size();
    return GTARS_OK;
}

gtars_status gtars_subset_source_indices(const gtars_index_t *ix, const uint32_t *qc, const uint32_t *qs, const uint32_t *qe,
                                         uint64_t nq, int has_min, int32_t min_overlap, uint32_t **out_idx, uint64_t *out_n) {
    return guarded([&] { return gtars_subset_source_indices_impl(ix, qc, qs, qe, nq, has_min, min_overlap, out_idx, out_n); });
}

// ---------------------------------------------------------------------- IGD

// PIECE_FLAG in a record's file id (pieces view only, see build_pieces_view): the record is a continuation piece
constexpr u32 IGD_PIECE_FLAG = 0x80000000u;
static gtars_status gtars_igd_build_core(const uint32_t *chrom, const int32_t *start, const int32_t *end,
                             const int32_t *value, const uint32_t *file_idx, uint64_t n, uint32_t n_chrom,
                             uint32_t n_files, bool piece_flags, gtars_igd_t **out) {
    const u32 fmask = piece_flags ? ~IGD_PIECE_FLAG : 0xFFFFFFFFu;
    if (!out) return fail(GTARS_ERR_INVALID_ARG, "out is NULL");
    *out = nullptr;
    if (n && (!chrom || !start || !end || !file_idx)) return fail(GTARS_ERR_INVALID_ARG, "NULL record arrays");
    gtars_status st = require_device();
    if (st) return st;
    if (n >= 0xFFFFFFF0ull) return fail(GTARS_ERR_INVALID_ARG, "too many records");
    if (use_device_sort(n)) {
        // ---- large builds: one sequential host pass (drop rule, bounds, per-chromosome counts and max
        // length), then upload, radix sort and gather on the device.  Dropped records get the key
        // n_chrom, so they sort behind every chromosome and are cut off.
        std::vector<u32> kc(n), hoff(n_chrom + 1, 0);
        std::vector<i32> hml(n_chrom, 0);
        u64 kept = 0;
        for (u64 i = 0; i < n; ++i) {
            // Igd::add drop rule (igd.rs:114-116)
            if (start[i] < 0 || end[i] < 0 || start[i] >= end[i]) {
                kc[i] = n_chrom;
                continue;
            }
            if (chrom[i] >= n_chrom) return fail(GTARS_ERR_INVALID_ARG, "record chromosome id >= n_chrom");
            if ((file_idx[i] & fmask) >= n_files) return fail(GTARS_ERR_INVALID_ARG, "record file_idx >= n_files");
            kc[i] = chrom[i];
            hoff[chrom[i] + 1]++;
            hml[chrom[i]] = std::max(hml[chrom[i]], end[i] - start[i]);
            ++kept;
        }
        for (u32 c = 0; c < n_chrom; ++c) hoff[c + 1] += hoff[c];
        auto *g = new gtars_igd();
        (void)hipGetDevice(&g->device);
        g->piece_flags = piece_flags;
        g->n_chrom = n_chrom;
        g->n_files = n_files;
        g->n = kept;
        auto bail = [&](gtars_status e) {
            gtars_igd_free(g);
            return e;
        };
        ScopedDev in, ws;
        const u32 n32 = (u32)n;
        // (a pieces view serves counts only: no values column, here or on the host path below)
        const bool with_values = !piece_flags;
        if ((st = in.alloc((size_t)n * 4 * (with_values ? 6 : 5)))) return bail(st);
        u32 *d_kc = in.as<u32>(), *d_s = d_kc + n, *d_e = d_s + n, *d_f = d_e + n, *d_v = d_f + n, *d_perm = with_values ? d_v + n : d_v;
        if (hipMemcpy(d_kc, kc.data(), n * 4, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(d_s, start, n * 4, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(d_e, end, n * 4, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(d_f, file_idx, n * 4, hipMemcpyHostToDevice) != hipSuccess ||
            (!with_values ? hipSuccess
                          : value ? hipMemcpy(d_v, value, n * 4, hipMemcpyHostToDevice) : hipMemset(d_v, 0, n * 4)) != hipSuccess)
            return bail(fail(GTARS_ERR_HIP, "IGD build: upload failed"));
        // chromosome-major, then start, ties in insertion order (finalize: stable sort by start, igd.rs:157-167);
        // kept starts are >= 0, so u32 order == i32 order
        const size_t ws_bytes = device_sort_perm_ws_bytes(n32);
        if ((st = ws.alloc(ws_bytes))) return bail(st);
        if ((st = device_sort_perm_ws(d_kc, d_s, nullptr, n32, n_chrom + 1, d_perm, ws.p, ws_bytes, nullptr))) return bail(st);
        const size_t kb = std::max<u64>(kept, 1) * 4 + 32;  // + slack: the sweep stages whole 16-byte vectors
        if (hipMalloc((void **)&g->starts.p, kb) != hipSuccess || hipMalloc((void **)&g->ends.p, kb) != hipSuccess ||
            hipMalloc((void **)&g->files.p, kb) != hipSuccess || (with_values && hipMalloc((void **)&g->values.p, kb) != hipSuccess))
            return bail(fail(GTARS_ERR_HIP, "IGD build: device allocation failed"));
        g->starts.n = g->ends.n = g->files.n = kept;
        g->values.n = with_values ? kept : 0;
        const u32 k32 = (u32)kept;
        if ((st = device_gather_u32(d_s, d_perm, k32, (u32 *)g->starts.p, nullptr))) return bail(st);
        if ((st = device_gather_u32(d_e, d_perm, k32, (u32 *)g->ends.p, nullptr))) return bail(st);
        if ((st = device_gather_u32(d_f, d_perm, k32, g->files.p, nullptr))) return bail(st);
        if (with_values && (st = device_gather_u32(d_v, d_perm, k32, (u32 *)g->values.p, nullptr))) return bail(st);
        if (hipDeviceSynchronize() != hipSuccess) return bail(fail(GTARS_ERR_HIP, "IGD build: device sort failed"));
        std::vector<u32> tf, tc, tch;
        for (u32 c = 0; c < n_chrom; ++c)
            for (u32 p = hoff[c]; p < hoff[c + 1]; p += IGD_TILE_RECORDS) {
                tf.push_back(p);
                tc.push_back(std::min<u32>(IGD_TILE_RECORDS, hoff[c + 1] - p));
                tch.push_back(c);
            }
        g->n_tiles = (u32)tf.size();
        st = g->tile_first.upload(tf);
        if (!st) st = g->tile_cnt.upload(tc);
        if (!st) st = g->tile_chrom.upload(tch);
        if (!st) st = g->chrom_off.upload(hoff);
        if (!st) st = g->chrom_maxlen.upload(hml);
        if (st) return bail(st);
        {
            // tile_carry: per-tile maximum end on the device, exclusive running maximum per chromosome on the host
            std::vector<i32> tmax(g->n_tiles, 0), carry(g->n_tiles, 0);
            ScopedDev d_tmax;
            if ((st = d_tmax.alloc((size_t)g->n_tiles * 4))) return bail(st);
            if ((st = launch_igd_tile_max_end(g->ends.p, g->tile_first.p, g->tile_cnt.p, g->n_tiles, d_tmax.as<i32>(), nullptr)))
                return bail(st);
            if (g->n_tiles && hipMemcpy(tmax.data(), d_tmax.p, (size_t)g->n_tiles * 4, hipMemcpyDeviceToHost) != hipSuccess)
                return bail(fail(GTARS_ERR_HIP, "IGD build: tile maxima readback failed"));
            i32 run = 0;
            for (u32 t = 0; t < g->n_tiles; ++t) {
                if (t == 0 || tch[t] != tch[t - 1]) run = 0;
                carry[t] = run;
                run = std::max(run, tmax[t]);
            }
            if ((st = g->tile_carry.upload(carry))) return bail(st);
        }
        if ((st = g->finish_tiles(tch))) return bail(st);
        *out = g;
        return GTARS_OK;
    }
    // ---- small builds: host stable sort
    // Igd::add drop rule (igd.rs:114-116)
    std::vector<u32> keep;
    keep.reserve(n);
    for (u64 i = 0; i < n; ++i) {
        if (start[i] < 0 || end[i] < 0 || start[i] >= end[i]) continue;
        if (chrom[i] >= n_chrom) return fail(GTARS_ERR_INVALID_ARG, "record chromosome id >= n_chrom");
        if ((file_idx[i] & fmask) >= n_files) return fail(GTARS_ERR_INVALID_ARG, "record file_idx >= n_files");
        keep.push_back((u32)i);
    }
    // chromosome-major, then start, ties in insertion order (finalize: stable sort by start, igd.rs:157-167)
    std::stable_sort(keep.begin(), keep.end(), [&](u32 a, u32 b) {
        if (chrom[a] != chrom[b]) return chrom[a] < chrom[b];
        return start[a] < start[b];
    });
    auto *g = new gtars_igd();
    (void)hipGetDevice(&g->device);
    g->piece_flags = piece_flags;
    g->n_chrom = n_chrom;
    g->n_files = n_files;
    g->n = keep.size();
    std::vector<i32> hv(g->n), hml(n_chrom, 0);
    std::vector<u32> hf(g->n), hoff(n_chrom + 1, 0);
    g->h_starts.resize(g->n);
    g->h_ends.resize(g->n);
    g->mirror_ready = true;
    for (u64 p = 0; p < g->n; ++p) {
        const u32 i = keep[p];
        g->h_starts[p] = start[i];
        g->h_ends[p] = end[i];
        hv[p] = value ? value[i] : 0;
        hf[p] = file_idx[i];
        hoff[chrom[i] + 1]++;
        hml[chrom[i]] = std::max(hml[chrom[i]], end[i] - start[i]);
    }
    for (u32 c = 0; c < n_chrom; ++c) hoff[c + 1] += hoff[c];
    std::vector<u32> tf, tc, tch;
    for (u32 c = 0; c < n_chrom; ++c)
        for (u32 p = hoff[c]; p < hoff[c + 1]; p += IGD_TILE_RECORDS) {
            tf.push_back(p);
            tc.push_back(std::min<u32>(IGD_TILE_RECORDS, hoff[c + 1] - p));
            tch.push_back(c);
        }
    g->n_tiles = (u32)tf.size();
    std::vector<i32> carry(g->n_tiles, 0);
    {
        i32 run = 0;
        for (u32 t = 0; t < g->n_tiles; ++t) {
            if (t == 0 || tch[t] != tch[t - 1]) run = 0;
            carry[t] = run;
            for (u32 p = tf[t]; p < tf[t] + tc[t]; ++p) run = std::max(run, g->h_ends[p]);
        }
    }
    st = g->tile_carry.upload(carry);
    if (!st) st = g->starts.upload(g->h_starts);
    if (!st) st = g->tile_first.upload(tf);
    if (!st) st = g->tile_cnt.upload(tc);
    if (!st) st = g->tile_chrom.upload(tch);
    if (!st) st = g->ends.upload(g->h_ends);
    if (!st && !piece_flags) st = g->values.upload(hv);  // (a pieces view serves counts only)
    if (!st) st = g->files.upload(hf);
    if (!st) st = g->chrom_off.upload(hoff);
    if (!st) st = g->chrom_maxlen.upload(hml);
    if (!st) st = g->finish_tiles(tch);
    if (st) {
        gtars_igd_free(g);
        return st;
    }
    *out = g;
    return GTARS_OK;
}

// ---- the pieces view ----------------------------------------------------------------------------------------------------------
// The flat layout (one record per stored interval, sorted by start) finds a query's candidates through the prefix maximum of the
// ends and routes it by the chromosome's LONGEST record: one multi-kilobase record inflates both for everything behind it, and a
// database with 1 % of 5-100 kbp records (broad peaks) took 15.6 ms for config 3's batch instead of 0.5 (0.1 % of records up to
// 1 Mbp: 182 ms).  The reference bounds the damage by replicating a record into every 16384-bp tile it spans (igd.rs:109-153) and
// counting it in the first tile of the query only (:812-817).  Same idea, for the long records only: a record longer than
// the piece length P (igd_piece_bp below) is cut at the multiples of P into PIECES that are stored as records of their own (continuation pieces
// flagged in bit 31 of the file id, bit 15 of the u16 copy), and a piece counts for a query iff it overlaps it AND (it is its
// record's first piece OR it starts at or before the query's start) -- i.e. the piece that holds max(q.start, record.start): every
// overlapping record exactly once.  No record of the view is longer than P, so prefix maxima and ownership stay local.
// "No earlier piece of the same file ends after q.start" still identifies the file's first counted piece (the earliest piece that
// ends after q.start is a counted one: the pieces of a record before its counted piece end at or before q.start), so the binary
// count keeps its pme_file form.  Used for min_overlap == 1 (the LOLA / igd-search default); other values keep the flat layout.
// Piece length: a power of two, 4x the power of two below the 90th percentile of the record lengths, at least 1024 -- long enough
// that nine records in ten are never cut, short enough that a piece inflates a query's candidate range by no more than a few
// typical records (config 3's widths with 1 % of 5-100 kbp records: 2048; measured 1.49 / 1.01 / 0.87 / 0.79 ms at 16384 / 8192 /
// 4096 / 2048).  The reference's 16384 is the upper bound.  0: no pieces view.
static i32 igd_piece_bp(const int32_t *start, const int32_t *end, uint64_t n) {
    if (cfg_get("GTARS_IGD_NO_PIECES")) return 0;
    if (const char *e = cfg_get("GTARS_IGD_PIECE_BP")) {  // tests: tiny pieces on small databases
        const long v = atol(e);
        if (v >= 16 && v <= (1 << 30)) return (i32)v;
    }
    u64 hist[32] = {}, kept = 0;
    for (u64 i = 0; i < n; ++i) {
        if (start[i] < 0 || end[i] < 0 || start[i] >= end[i]) continue;
        hist[31 - __builtin_clz((u32)(end[i] - start[i]))]++;
        ++kept;
    }
    if (!kept) return 0;
    u64 cum = 0;
    int b = 0;
    for (; b < 31; ++b) {
        cum += hist[b];
        if (cum * 10 >= kept * 9) break;
    }
    const int sh = std::min(std::max(b + 2, 10), 14);  // [1024, 16384]
    return (i32)1 << sh;
}

static gtars_status build_pieces_view(gtars_igd *g, const uint32_t *chrom, const int32_t *start, const int32_t *end,
                                      const uint32_t *file_idx, uint64_t n) {
    if (!g || !g->n) return GTARS_OK;
    const i32 P = igd_piece_bp(start, end, n);
    if (!P) return GTARS_OK;
    if (cfg_flag("GTARS_IGD_TEST_PIECES_FAIL")) return fail(GTARS_ERR_HIP, "test hook: the pieces view could not be allocated");
    u64 n_long = 0, n_pieces = 0;
    for (u64 i = 0; i < n; ++i) {
        if (start[i] < 0 || end[i] < 0 || start[i] >= end[i]) continue;  // Igd::add drop rule
        ++n_pieces;
        if (end[i] - start[i] > P) {
            ++n_long;
            n_pieces += (u64)((end[i] - 1) / P - start[i] / P);
        }
    }
    if (!n_long) return GTARS_OK;
    if (n_pieces >= 0xFFFFFFF0ull) return GTARS_OK;  // (would not fit the 32-bit record index: the flat layout serves)
    std::vector<u32> pc(n_pieces), pf(n_pieces);
    std::vector<i32> ps(n_pieces), pe(n_pieces);
    u64 k = 0;
    for (u64 i = 0; i < n; ++i) {
        const i32 s0 = start[i], e0 = end[i];
        if (s0 < 0 || e0 < 0 || s0 >= e0) continue;
        if (e0 - s0 <= P) {
            pc[k] = chrom[i], ps[k] = s0, pe[k] = e0, pf[k] = file_idx[i];
            ++k;
            continue;
        }
        i32 a = s0;
        bool first = true;
        while (a < e0) {
            const i64 nb = ((i64)a / P + 1) * (i64)P;  // next multiple of P above a
            const i32 b = (i32)std::min<i64>(nb, e0);
            pc[k] = chrom[i], ps[k] = a, pe[k] = b, pf[k] = file_idx[i] | (first ? 0u : IGD_PIECE_FLAG);
            ++k;
            first = false;
            a = b;
        }
    }
    gtars_igd *pv = nullptr;
    gtars_status st = gtars_igd_build_core(pc.data(), ps.data(), pe.data(), nullptr, pf.data(), k, g->n_chrom, g->n_files, true, &pv);
    if (st) return st;
    g->pieces = pv;
    return GTARS_OK;
}

static gtars_status gtars_igd_build_impl(const uint32_t *chrom, const int32_t *start, const int32_t *end,
                                         const int32_t *value, const uint32_t *file_idx, uint64_t n, uint32_t n_chrom,
                                         uint32_t n_files, gtars_igd_t **out) {
    gtars_status st = gtars_igd_build_core(chrom, start, end, value, file_idx, n, n_chrom, n_files, false, out);
    if (st) return st;
    // The pieces view is an accelerator, not part of the database: when it cannot be built (device or host memory: it is a second
    // copy of the records) the flat layout alone serves every query correctly, only slower for databases with long records.
    try {
        const gtars_status pv = build_pieces_view(*out, chrom, start, end, file_idx, n);
        if (pv != GTARS_OK) {
            // (an accelerator that is missing is a slowdown nobody asked about: say so once per process, and why -- a failure that
            // is not a memory shortage would otherwise only ever show as that slowdown)
            static std::atomic<bool> said{false};
            if (!said.exchange(true))
                fprintf(stderr, "gtars_amd: the pieces view of an IGD database could not be built (status %d: %s); databases with long records count slower\n",
                        (int)pv, gtars_last_error());
            prof_note_fact("igd_pieces_view_dropped");
            (void)hipGetLastError();
            set_error("");
        }
    } catch (const std::bad_alloc &) {
        (*out)->pieces = nullptr;
    }
    return GTARS_OK;
}

// which index serves a count: the pieces view for min_overlap == 1 when the database has one (binary counts: only in their
// pme_file form)
static const gtars_igd *igd_count_target(const gtars_igd *g, int32_t min_overlap, int binary) {
    if (!g->pieces || min_overlap != 1) return g;
    if (binary && cfg_get("GTARS_IGD_NO_PME")) return g;
    return g->pieces;
}

void gtars_igd_free(gtars_igd_t *g) {
    if (!g) return;
    gtars_igd_free(g->pieces);
    g->starts.release();
    g->ends.release();
    g->values.release();
    g->files.release();
    g->chrom_off.release();
    g->chrom_maxlen.release();
    g->tile_first.release();
    g->tile_cnt.release();
    g->tile_chrom.release();
    g->tile_carry.release();
    g->tile_bnd.release();
    g->chrom_tile_off.release();
    g->tile_pm.release();
    g->tile_files16.release();
    g->tile_tab.release();
    g->tile_ends_sorted.release();
    g->tile_erank.release();
    g->tile_tab_r.release();
    g->chrom_ntiles.release();
    g->route_lut.release();
    g->route_base.release();
    g->route_len.release();
    g->route_flut.release();
    g->route_kq.release();
    g->route_fbase.release();
    g->pme_file.release();
    delete g;
}

uint64_t gtars_igd_len(const gtars_igd_t *g) { return g ? g->n : 0; }
uint32_t gtars_igd_n_files(const gtars_igd_t *g) { return g ? g->n_files : 0; }
int gtars_igd_device(const gtars_igd_t *g) { return g ? g->device : -1; }

// include/gtars_amd_debug.h: forge the device id a handle records (tests of the device-affinity checks on a one-GPU box)
int gtars_debug_set_handle_device(void *handle, int is_igd, int device) {
    if (!handle) return -1;
    int before;
    if (is_igd) {
        gtars_igd *g = (gtars_igd *)handle;
        before = g->device;
        g->device = device;
        if (g->pieces) g->pieces->device = device;
    } else {
        gtars_index *ix = (gtars_index *)handle;
        before = ix->device;
        ix->device = device;
        if (ix->flat) ix->flat->device = device;
    }
    return before;
}

uint64_t gtars_igd_total_records(const gtars_igd_t *g, int32_t nbp) {
    if (!g) return 0;
    if (nbp <= 0) nbp = 16384;
    DeviceScope on_handle_device(g->device);
    if (on_handle_device.st || g->ensure_mirror()) return 0;
    u64 t = 0;
    for (u64 i = 0; i < g->n; ++i) t += (u64)((g->h_ends[i] - 1) / nbp - g->h_starts[i] / nbp + 1);
    return t;
}

gtars_status gtars_igd_export(const gtars_igd_t *g, uint32_t *chrom, int32_t *start, int32_t *end, int32_t *value,
                              uint32_t *file_idx) {
    GT_ON_DEVICE_OF(g);
    if (!g) return fail(GTARS_ERR_INVALID_ARG, "NULL handle");
    const size_t n = g->n;
    if (!n) return GTARS_OK;
    if (start || end) {
        gtars_status ms = g->ensure_mirror();
        if (ms) return ms;
    }
    if (start) memcpy(start, g->h_starts.data(), n * 4);
    if (end) memcpy(end, g->h_ends.data(), n * 4);
    if (value) GT_HIP(hipMemcpy(value, g->values.p, n * 4, hipMemcpyDeviceToHost));
    if (file_idx) GT_HIP(hipMemcpy(file_idx, g->files.p, n * 4, hipMemcpyDeviceToHost));
    if (chrom) {
        std::vector<u32> off(g->n_chrom + 1);
        GT_HIP(hipMemcpy(off.data(), g->chrom_off.p, off.size() * 4, hipMemcpyDeviceToHost));
        for (u32 c = 0; c < g->n_chrom; ++c)
            for (u32 i = off[c]; i < off[c + 1]; ++i) chrom[i] = c;
    }
    return GTARS_OK;
}

gtars_status gtars_igd_count_device(const gtars_igd_t *g, const uint32_t *d_qc, const uint32_t *d_qs,
                                    const uint32_t *d_qe, uint64_t nq, int32_t min_overlap, int binary,
                                    uint64_t *d_hits, void *stream) {
    GT_SAME_DEVICE_AS(g);
    gtars_status st = check_query_args(g, d_qc, d_qs, d_qe, nq);
    if (st) return st;
    if (!d_hits) return fail(GTARS_ERR_INVALID_ARG, "d_hits is NULL");
    if (min_overlap < 1) {
        // The reference's tile walk also admits non-overlapping records then, depending on the 16384-bp tile they fall in
        // (igd.rs:772-846): reproduced by the per-query kernel with the walk's tile test (kernels.hip, IgdQual).
        if ((st = g->ensure_ntiles())) return st;
        return launch_igd_count(g->view(), nullptr, d_qc, d_qs, d_qe, nq, min_overlap, binary, d_hits, (hipStream_t)stream);
    }
    {
        const gtars_igd *t = igd_count_target(g, min_overlap, binary);  // (the pieces view of a database with long records)
        if (t != g && g_prof_on) g_prof_entries[prof_entry("igd_pieces_view")].launches += 1;  // (a fact for the tests)
        g = t;
    }
    if (igd_sweep_supported(g->view(), nq)) {
        // large batch: group the queries by owner tile once, stream the database once (igd_sweep.hip)
        const bool no_pme = cfg_get("GTARS_IGD_NO_PME") != nullptr;  // tests / A-B runs: the credited-file list instead
        if (binary && min_overlap == 1 && !no_pme && (st = g->ensure_pme())) return st;
        Workspace &ws = tls_workspace(2, (hipStream_t)stream);
        st = ws.reserve(igd_sweep_ws_bytes(nq, g->n_tiles, g->n_chrom));
        if (st) return st;
        return launch_igd_sweep(g->view(), g->tiles(), d_qc, d_qs, d_qe, nq, min_overlap, binary, d_hits, ws.ptr, ws.bytes,
                                (hipStream_t)stream, ws.igd_calls++);
    }
    // small batch: one thread per query (kernels.hip); binary counts with min_overlap == 1 through pme_file as well
    if (binary && min_overlap == 1 && !cfg_get("GTARS_IGD_NO_PME") && (st = g->ensure_pme())) return st;
    return launch_igd_count(g->view(), g->tiles().pme_file, d_qc, d_qs, d_qe, nq, min_overlap, binary, d_hits, (hipStream_t)stream,
                            g->tiles().pm);
}

// Several query sets against one database (LOLA: the universe and the user sets, enrichment.rs:198-221 -- the reference calls
// count_region_hits once per set).  Up to 4 sets share ONE pass over the database: the partition tags every (start, end) pair with
// its set and the sweep keeps one row of LDS counters per set; sets that cannot share a pass (more than 4, too many files for the
// counters, batches below the sweep's crossover, min_overlap < 1) are counted one call per set -- same vectors either way.
gtars_status gtars_igd_count_sets_device(const gtars_igd_t *g, const uint32_t *d_qc, const uint32_t *d_qs, const uint32_t *d_qe,
                                         const uint64_t *set_off, uint32_t n_sets, int32_t min_overlap, int binary,
                                         uint64_t *d_hits, void *stream) {
    GT_SAME_DEVICE_AS(g);
    if (!g) return fail(GTARS_ERR_INVALID_ARG, "NULL handle");
    if (!set_off || n_sets == 0) return fail(GTARS_ERR_INVALID_ARG, "set_off is NULL or n_sets is 0");
    if (set_off[0] != 0) return fail(GTARS_ERR_INVALID_ARG, "set_off[0] must be 0");
    for (u32 k = 0; k < n_sets; ++k)
        if (set_off[k + 1] < set_off[k]) return fail(GTARS_ERR_INVALID_ARG, "set_off must not decrease");
    const u64 nq = set_off[n_sets];
    gtars_status st = check_query_args(g, d_qc, d_qs, d_qe, nq);
    if (st) return st;
    if (!d_hits) return fail(GTARS_ERR_INVALID_ARG, "d_hits is NULL");
    const size_t F = g->n_files;
    const gtars_igd *gs = igd_count_target(g, min_overlap, binary);  // what a shared pass sweeps
    const bool no_shared = cfg_get("GTARS_IGD_NO_SHARED_PASS") != nullptr;  // tests / A-B runs
    for (u32 k0 = 0; k0 < n_sets;) {
        // the longest run of <= 4 consecutive sets that can share a pass
        u32 k1 = k0 + 1;
        if (min_overlap >= 1 && !no_shared) {
            u32 best = k0 + 1;
            for (u32 k = k0 + 2; k <= std::min<u32>(n_sets, k0 + 4); ++k)
                if (igd_sweep_sets_supported(gs->view(), gs->tiles(), set_off[k] - set_off[k0], k - k0)) best = k;
            k1 = best;
        }
        const u64 lo = set_off[k0], n = set_off[k1] - lo;
        if (k1 - k0 == 1) {
            st = gtars_igd_count_device(g, d_qc + lo, d_qs + lo, d_qe + lo, n, min_overlap, binary, d_hits + (size_t)k0 * F, stream);
            if (st) return st;
        } else {
            const bool no_pme = cfg_get("GTARS_IGD_NO_PME") != nullptr;
            if (binary && min_overlap == 1 && !no_pme && (st = gs->ensure_pme())) return st;
            Workspace &ws = tls_workspace(2, (hipStream_t)stream);
            st = ws.reserve(igd_sweep_ws_bytes(n, gs->n_tiles, gs->n_chrom));
            if (st) return st;
            u32 bounds[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
            for (u32 k = k0 + 1; k < k1; ++k) bounds[k - k0 - 1] = (u32)(set_off[k] - lo);
            st = launch_igd_sweep(gs->view(), gs->tiles(), d_qc + lo, d_qs + lo, d_qe + lo, n, min_overlap, binary, d_hits + (size_t)k0 * F,
                                  ws.ptr, ws.bytes, (hipStream_t)stream, ws.igd_calls++, k1 - k0, bounds);
            if (st) return st;
            if (g_prof_on) g_prof_entries[prof_entry("igd_sets_shared_pass")].launches += 1;  // (a fact for the tests, not a time)
        }
        k0 = k1;
    }
    return GTARS_OK;
}

gtars_status gtars_igd_count_sets(const gtars_igd_t *g, const uint32_t *qc, const uint32_t *qs, const uint32_t *qe,
                                  const uint64_t *set_off, uint32_t n_sets, int32_t min_overlap, int binary, uint64_t *hits) {
    GT_ON_DEVICE_OF(g);
    if (!set_off || n_sets == 0) return fail(GTARS_ERR_INVALID_ARG, "set_off is NULL or n_sets is 0");
    const u64 nq = set_off[n_sets];
    gtars_status st = check_query_args(g, qc, qs, qe, nq);
    if (st) return st;
    if (!hits && g->n_files) return fail(GTARS_ERR_INVALID_ARG, "hits is NULL");
    st = require_device();
    if (st) return st;
    DevQueries q;
    st = q.upload(qc, qs, qe, nq);
    if (st) return st;
    ScopedDev d;
    const size_t cells = (size_t)std::max<u32>(g->n_files, 1) * n_sets;
    st = d.alloc(cells * 8);
    if (st) return st;
    st = gtars_igd_count_sets_device(g, q.c, q.s, q.e, set_off, n_sets, min_overlap, binary, d.as<u64>(), nullptr);
    if (st) return st;
    if (g->n_files) GT_HIP(hipMemcpy(hits, d.p, (size_t)g->n_files * n_sets * 8, hipMemcpyDeviceToHost));
    return GTARS_OK;
}

gtars_status gtars_igd_count(const gtars_igd_t *g, const uint32_t *qc, const uint32_t *qs, const uint32_t *qe,
                             uint64_t nq, int32_t min_overlap, int binary, uint64_t *hits) {
    GT_ON_DEVICE_OF(g);
    gtars_status st = check_query_args(g, qc, qs, qe, nq);
    if (st) return st;
    if (!hits && g->n_files) return fail(GTARS_ERR_INVALID_ARG, "hits is NULL");
    st = require_device();
    if (st) return st;
    DevQueries q;
    st = q.upload(qc, qs, qe, nq);
    if (st) return st;
    ScopedDev d;
    st = d.alloc((size_t)std::max<u32>(g->n_files, 1) * 8);
    if (st) return st;
    st = gtars_igd_count_device(g, q.c, q.s, q.e, nq, min_overlap, binary, d.as<u64>(), nullptr);
    if (st) return st;
    if (g->n_files) GT_HIP(hipMemcpy(hits, d.p, (size_t)g->n_files * 8, hipMemcpyDeviceToHost));
    return GTARS_OK;
}

gtars_status gtars_igd_count_per_query(const gtars_igd_t *g, const uint32_t *qc, const uint32_t *qs,
                                       const uint32_t *qe, uint64_t nq, int32_t min_overlap,
                                       uint32_t *counts) {
    GT_ON_DEVICE_OF(g);
    gtars_status st = check_query_args(g, qc, qs, qe, nq);
    if (st) return st;
    st = require_device();
    if (st) return st;
    if (!nq) return GTARS_OK;
    if (min_overlap < 1 && (st = g->ensure_ntiles())) return st;  // the walk's tile test (kernels.hip, IgdQual)
    DevQueries q;
    st = q.upload(qc, qs, qe, nq);
    if (st) return st;
    ScopedDev d;
    st = d.alloc(nq * 4);
    if (st) return st;
    bool uniq = false;
    if ((st = g->ensure_values_unique(&uniq))) return st;
    st = launch_igd_count_per_query(g->view(), q.c, q.s, q.e, nq, min_overlap, d.as<u32>(), uniq, nullptr);
    if (st) return st;
    GT_HIP(hipMemcpy(counts, d.p, nq * 4, hipMemcpyDeviceToHost));
    return GTARS_OK;
}

gtars_status gtars_igd_find_pairs(const gtars_igd_t *g, const uint32_t *qc, const uint32_t *qs,
                                  const uint32_t *qe, uint64_t nq, int32_t min_overlap, uint32_t **out_q,
                                  uint32_t **out_s, uint64_t *out_n) {
    GT_ON_DEVICE_OF(g);
    gtars_status st = check_query_args(g, qc, qs, qe, nq);
    if (st) return st;
    if (!out_q || !out_s || !out_n) return fail(GTARS_ERR_INVALID_ARG, "NULL output");
    *out_q = *out_s = nullptr;
    *out_n = 0;
    st = require_device();
    if (st) return st;
    if (min_overlap < 1 && (st = g->ensure_ntiles())) return st;  // the walk's tile test (kernels.hip, IgdQual)
    DevQueries q;
    st = q.upload(qc, qs, qe, nq);
    if (st) return st;
    ScopedDev d_cnt, d_off, d_ws;
    if ((st = d_cnt.alloc(nq * 4))) return st;
    if ((st = d_off.alloc((nq + 1) * 8))) return st;
    const size_t wsb = scan_ws_bytes(nq);
    if ((st = d_ws.alloc(wsb))) return st;
    bool uniq = false;
    if ((st = g->ensure_values_unique(&uniq))) return st;
    st = launch_igd_count_per_query(g->view(), q.c, q.s, q.e, nq, min_overlap, d_cnt.as<u32>(), uniq, nullptr);
    if (st) return st;
    st = launch_scan_u32_to_u64(d_cnt.as<u32>(), nq, d_off.as<u64>(), d_ws.p, wsb, nullptr);
    if (st) return st;
    u64 h = 0;
    GT_HIP(hipMemcpy(&h, d_off.as<u64>() + nq, 8, hipMemcpyDeviceToHost));
    ScopedDev d_q, d_s;
    if ((st = d_q.alloc(h * 4))) return st;
    if ((st = d_s.alloc(h * 4))) return st;
    st = launch_igd_fill_pairs(g->view(), q.c, q.s, q.e, nq, min_overlap, d_off.as<u64>(), d_q.as<u32>(),
                               d_s.as<u32>(), uniq, nullptr);
    if (st) return st;
    *out_q = host_alloc<u32>(h);
    *out_s = host_alloc<u32>(h);
    if (!*out_q || !*out_s) return fail(GTARS_ERR_INTERNAL, "out of host memory");
    if (h) {
        GT_HIP(hipMemcpy(*out_q, d_q.p, h * 4, hipMemcpyDeviceToHost));
        GT_HIP(hipMemcpy(*out_s, d_s.p, h * 4, hipMemcpyDeviceToHost));
    }
    *out_n = h;
    return GTARS_OK;
}

gtars_status gtars_lola_contingency_device(const uint64_t *d_user_hits, const uint64_t *d_universe_hits,
                                           uint64_t n_files, int64_t user_size, int64_t universe_size,
                                           int64_t *d_a, int64_t *d_b, int64_t *d_c, int64_t *d_d,
                                           void *stream) {
    if (n_files && (!d_user_hits || !d_universe_hits || !d_a || !d_b || !d_c || !d_d))
        return fail(GTARS_ERR_INVALID_ARG, "NULL argument");
    return launch_lola_contingency(d_user_hits, d_universe_hits, n_files, user_size, universe_size, d_a, d_b,
                                   d_c, d_d, (hipStream_t)stream);
}

}  // extern "C"

// ---- Bits::insert / Bits::seek (host side; SURVEY 8 a4': no caller on the hot path)
// bits.rs:304-322 bsearch_seq_ref over the (start, end) order of one chromosome's stored intervals
static u32 bits_insert_pos(const u32 *S, const u32 *E, u32 n, u32 start, u32 end) {
    auto lt = [&](u32 i) { return S[i] < start || (S[i] == start && E[i] < end); };  // elems[i] < key (interval.rs:18-30)
    if (n == 0 || !lt(0)) return 0;
    if (lt(n - 1)) return n;
    u32 cursor = 0, length = n;
    while (length > 1) {
        const u32 half = length >> 1;
        length -= half;
        cursor += lt(cursor + half - 1) ? half : 0u;
    }
    return cursor;
}

static gtars_status gtars_index_insert_impl(const gtars_index_t *ix, uint32_t chrom, uint32_t start, uint32_t end,
                                            uint32_t val, gtars_index_t **out) {
    GT_ON_DEVICE_OF(ix);
    if (!out) return fail(GTARS_ERR_INVALID_ARG, "out is NULL");
    *out = nullptr;
    if (!ix) return fail(GTARS_ERR_INVALID_ARG, "NULL index");
    if (ix->kind != GTARS_KIND_BITS) return fail(GTARS_ERR_INVALID_ARG, "insert is a Bits operation (bits.rs:209-222)");
    if (chrom >= ix->n_chrom) return fail(GTARS_ERR_INVALID_ARG, "interval chromosome id >= n_chrom");
    const u64 n = ix->n + 1;
    std::vector<u32> c(n), s(n), e(n), v(n);
    const u32 lo = ix->h_chrom_off[chrom], hi = ix->h_chrom_off[chrom + 1];
    const u32 at = lo + bits_insert_pos(ix->h_starts.data() + lo, ix->h_ends.data() + lo, hi - lo, start, end);
    u64 w = 0;
    for (u32 k = 0; k < ix->n_chrom; ++k)
        for (u32 p = ix->h_chrom_off[k]; p <= ix->h_chrom_off[k + 1]; ++p) {
            if (k == chrom && p == at) {  // in front of equal (start, end) keys: the stable build keeps it there
                c[w] = chrom, s[w] = start, e[w] = end, v[w] = val;
                ++w;
            }
            if (p == ix->h_chrom_off[k + 1]) break;
            c[w] = k, s[w] = ix->h_starts[p], e[w] = ix->h_ends[p], v[w] = ix->h_vals[p];
            ++w;
        }
    return gtars_index_build_impl(c.data(), s.data(), e.data(), v.data(), n, ix->n_chrom, GTARS_KIND_BITS, out);
}

static gtars_status gtars_index_seek_impl(const gtars_index_t *ix, uint32_t chrom, uint32_t start, uint32_t stop,
                                          uint64_t *cursor, uint32_t *out_vals, uint64_t capacity, uint64_t *n_hits) {
    if (!ix || !cursor || !n_hits) return fail(GTARS_ERR_INVALID_ARG, "NULL argument");
    if (ix->kind != GTARS_KIND_BITS) return fail(GTARS_ERR_INVALID_ARG, "seek is a Bits operation (bits.rs:364-386)");
    *n_hits = 0;
    if (chrom >= ix->n_chrom) return GTARS_OK;
    const u32 lo = ix->h_chrom_off[chrom];
    const u64 n = ix->h_chrom_off[chrom + 1] - lo;
    const u32 *S = ix->h_starts.data() + lo, *E = ix->h_ends.data() + lo, *V = ix->h_vals.data() + lo;
    const u32 max_len = ix->h_chrom_aux[chrom];
    const u32 key = start >= max_len ? start - max_len : 0u;  // checked_sub(..).unwrap_or(0)
    u64 cur = *cursor;
    if (cur == 0 || (cur < n && S[cur] > start)) {
        // Bits::lower_bound (bits.rs:250-264), same probe sequence
        u64 size = n, low = 0;
        while (size > 0) {
            const u64 half = size / 2, other_half = size - half, probe = low + half, other_low = low + other_half;
            size = half;
            low = S[probe] < key ? other_low : low;
        }
        cur = low;
    }
    while (cur + 1 < n && S[cur + 1] < key) ++cur;
    *cursor = cur;
    u64 k = 0;
    for (u64 off = cur; off < n; ++off) {  // IterFind::next, bits.rs:433-446
        if (S[off] < stop && E[off] > start) {
            if (out_vals && k < capacity) out_vals[k] = V[off];
            ++k;
        } else if (S[off] >= stop) {
            break;
        }
    }
    *n_hits = k;
    if (out_vals && k > capacity) return fail(GTARS_ERR_CAPACITY, "seek: more hits than capacity");
    return GTARS_OK;
}

// ---- the C ABI never lets a C++ exception cross it (std::bad_alloc of a huge build, std::length_error ...)
extern "C" {

gtars_status gtars_index_build(const uint32_t *chrom, const uint32_t *start, const uint32_t *end,
                               const uint32_t *val, uint64_t n, uint32_t n_chrom, int kind,
                               gtars_index_t **out) {
    return guarded([&]() -> gtars_status { return gtars_index_build_impl(chrom, start, end, val, n, n_chrom, kind, out); });
}

gtars_status gtars_index_insert(const gtars_index_t *ix, uint32_t chrom, uint32_t start, uint32_t end, uint32_t val,
                                gtars_index_t **out) {
    return guarded([&]() -> gtars_status { return gtars_index_insert_impl(ix, chrom, start, end, val, out); });
}

gtars_status gtars_index_seek(const gtars_index_t *ix, uint32_t chrom, uint32_t start, uint32_t stop, uint64_t *cursor,
                              uint32_t *out_vals, uint64_t capacity, uint64_t *n_hits) {
    return guarded([&]() -> gtars_status { return gtars_index_seek_impl(ix, chrom, start, stop, cursor, out_vals, capacity, n_hits); });
}

gtars_status gtars_igd_build(const uint32_t *chrom, const int32_t *start, const int32_t *end,
                             const int32_t *value, const uint32_t *file_idx, uint64_t n, uint32_t n_chrom,
                             uint32_t n_files, gtars_igd_t **out) {
    return guarded([&]() -> gtars_status { return gtars_igd_build_impl(chrom, start, end, value, file_idx, n, n_chrom, n_files, out); });
}

gtars_status gtars_tokenize(const gtars_index_t *ix, const uint32_t *qc, const uint32_t *qs,
                            const uint32_t *qe, uint64_t nq, uint64_t *offsets, uint32_t **out_ids,
                            uint64_t *out_n) {
    return guarded([&]() -> gtars_status { return gtars_tokenize_impl(ix, qc, qs, qe, nq, offsets, out_ids, out_n); });
}

gtars_status gtars_tokenize_into(const gtars_index_t *ix, const uint32_t *qc, const uint32_t *qs, const uint32_t *qe,
                                 uint64_t nq, uint64_t *offsets, uint32_t *ids, uint64_t ids_capacity, uint64_t *out_n) {
    return guarded([&]() -> gtars_status { return gtars_tokenize_into_impl(ix, qc, qs, qe, nq, offsets, ids, ids_capacity, out_n); });
}

}  // extern "C"
