// index.hip -- the C ABI (include/petal_mi355x.h): index lifetime, host <-> HBM
// staging, engine selection and kernel orchestration.  gfx950 / ROCm only.
// No CPU compute path exists here: without a GPU every compute entry point
// returns PN_ERR_DEVICE.
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <chrono>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/petal_mi355x.h"
#include "pn_internal.h"

using namespace pn;

// ---------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------
static thread_local std::string g_err;

static int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

namespace pn {
int set_error(int code, const char *fmt, ...) {  // for the other translation units of the library (sharded.hip, tree.cpp)
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
}  // namespace pn

#define HIPCHK(expr)                                                                            \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return fail(e_ == hipErrorOutOfMemory ? PN_ERR_NOMEM : PN_ERR_DEVICE, "%s: %s", #expr, \
                        hipGetErrorString(e_));                                                 \
    } while (0)
#define PNCHK(expr)                 \
    do {                            \
        int rc_ = (expr);           \
        if (rc_ != PN_OK) return rc_; \
    } while (0)

extern "C" const char *pn_last_error(void) { return g_err.c_str(); }
extern "C" int pn_abi_version(void) { return PN_ABI_VERSION; }
extern "C" const char *pn_strerror(int code) {
    switch (code) {
        case PN_OK: return "ok";
        case PN_ERR_EMPTY: return "array is empty";                         // src/lib.rs:12
        case PN_ERR_NOT_CONTIGUOUS: return "array is not contiguous in memory";  // src/lib.rs:14
        case PN_ERR_INVALID: return "invalid argument";
        case PN_ERR_DEVICE: return "GPU device error";
        case PN_ERR_NOMEM: return "out of memory";
        case PN_ERR_UNSUPPORTED: return "unsupported";
        case PN_ERR_EMPTY_MATRIX: return "empty matrix";                    // src/ball_tree.rs:582
        case PN_ERR_COMM: return "collective communication (RCCL) error";
        default: return "unknown error";
    }
}
extern "C" int pn_device_count(int *count) {
    if (!count) return fail(PN_ERR_INVALID, "count is NULL");
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) {
        *count = 0;
        return fail(PN_ERR_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
    }
    *count = c;
    return PN_OK;
}

// ---------------------------------------------------------------------------
// index object
// ---------------------------------------------------------------------------
// A buffer that only grows.  The asynchronous entry points return while kernels that were handed the OLD allocation
// may still be running, and hipFree would wait for the whole device (an "enqueue and return" call must not): an
// outgrown allocation is RETIRED to its owner's list instead and freed once the owner's end-of-use event has passed
// (ws_acquire) or at destruction.
struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    std::vector<void *> *retired = nullptr;  // owner's list; nullptr: free at once (construction-time buffers)
    int ensure(size_t need) {
        if (need <= bytes && p) return PN_OK;
        if (p) {
            bool kept = false;
            if (retired) {
                try {
                    retired->push_back(p);
                    kept = true;
                } catch (const std::bad_alloc &) {
                }
            }
            if (!kept) (void)hipFree(p);
        }
        p = nullptr;
        bytes = 0;
        size_t want = need + need / 8 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) return fail(PN_ERR_NOMEM, "hipMalloc(%zu): %s", want, hipGetErrorString(e));
        bytes = want;
        return PN_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
};

// Per-call scratch.  A query call takes one workspace from the index's pool for its whole duration, so calls from
// several host threads on one handle run side by side (BallTree queries take &self, `Euclidean: Sync`,
// src/distance.rs:19).  The *_device entry points return while their kernels are still running: `done` marks the end
// of the last call that used the workspace, and a later call on ANOTHER stream waits for it before touching the
// buffers (same stream: stream order is enough).
struct Workspace {
    DevBuf w_q, w_qnorm, w_qnrm, w_keys, w_idx, w_cnt, w_tau, w_flags, w_sel, w_misc, w2_keys, w2_idx, w2_cnt, w2_tau, w_lo;
    DevBuf w_bq, w_qn, w_qbad, w_gq, w_gqn, w_gidx, w_gdist, w_gsel, w_seed, w_qstat, w_lists;  // bf16 tier, second tier
    DevBuf w_hq, w_hidx, w_hdist;  // staging of the host entry points (queries up, results down)
    DevBuf w_fparts;               // second tier, many-segment path: per-group partial results
    DevBuf w_rpos, w_rfin, w_rcx, w_rox, w_rscan;  // pn_query_radius_device_*: list positions, counts, per-segment counts / offsets, scan scratch
    // small calls (tiny_*): mapped pinned host memory the one kernel of the call reads its queries from and writes its
    // answers to -- no copy commands
    void *pin_in = nullptr, *pin_out = nullptr;
    size_t pin_in_bytes = 0, pin_out_bytes = 0;
    int pin_ensure(void **p, size_t *have, size_t need) {
        if (need <= *have && *p) return PN_OK;
        if (*p) {
            (void)hipStreamSynchronize(stream);  // (host entry points wait for their stream: nothing of ours is in flight)
            (void)hipHostFree(*p);
        }
        *p = nullptr;
        *have = 0;
        const size_t want = need + need / 2 + 4096;
        if (hipHostMalloc(p, want, hipHostMallocMapped) != hipSuccess) return fail(PN_ERR_NOMEM, "hipHostMalloc(%zu) failed", want);
        *have = want;
        return PN_OK;
    }
    DevBuf w_pcnt;                 // shared thresholds: the segments' published fill counts [nseg][nq_pad]
    uint32_t sh_epoch = 0;         // epoch of the last launch that published into w_pcnt (1 .. 4095)
    size_t sh_nseg = 0, sh_nq_pad = 0;  // geometry w_pcnt was last used with (a change: zero it once)
    hipStream_t stream = nullptr;  // the host entry points run here
    hipEvent_t ev_r[2] = {nullptr, nullptr};  // radius calls under PN_OPT_PROFILE: the filter launch's bracket
    hipEvent_t done = nullptr;
    hipStream_t last_stream = nullptr;
    bool in_flight = false;
    DevBuf *all[36] = {&w_gqn, &w_rpos, &w_rfin, &w_rcx, &w_rox, &w_rscan, &w_q, &w_qnorm, &w_qnrm, &w_keys, &w_idx, &w_cnt, &w_tau, &w_flags, &w_sel, &w_misc, &w2_keys, &w2_idx,
                       &w2_cnt, &w2_tau, &w_lo, &w_bq, &w_qn, &w_qbad, &w_gq, &w_gidx, &w_gdist, &w_gsel, &w_seed,
                       &w_qstat, &w_lists, &w_hq, &w_hidx, &w_hdist, &w_fparts, &w_pcnt};
    std::vector<void *> retired;  // outgrown allocations, freed once `done` has passed (DevBuf::ensure)
    Workspace() {
        for (DevBuf *b : all) b->retired = &retired;
    }
    Workspace(const Workspace &) = delete;
    Workspace &operator=(const Workspace &) = delete;
    void free_retired() {
        for (void *q : retired) (void)hipFree(q);
        retired.clear();
    }
};

// What a finished chunk of a call leaves for the host to pick up LATER (never inside the call): hipEvent brackets of
// the dominant kernel (PN_OPT_PROFILE) and the number of queries its first tier could not prove (pinned memory).
struct CallRec {
    hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // 0/1 hot kernel, 4/5 scout|main split, 2/3 whole chunk
    hipEvent_t done = nullptr;
    uint32_t *h_nflag = nullptr;  // pinned
    size_t nq = 0;
    uint64_t call_id = 0;
    bool pending = false, busy = false, two_launches = false, has_flag = false, bf16_tier = false, hot = false;
    bool model_seed = false;  // the call's thresholds came from the index's seed model (no scout launch)
    int prof = 0;  // PN_OPT_PROFILE at rec_begin: 1 = the dominant kernel's brackets, 2 = also the whole chunk's
};
constexpr int kCallRecs = 16;

constexpr int kSeedModelRetry = 64, kSeedModelMaxRetries = 3;  // (see rec_resolve)
constexpr int kSeedModelGrid = 7;  // seed model: z calibrated at the corpus ranks 16, 32, ..., 1024
struct pn_index {
    int device = 0;
    int elem_bytes = 4;
    size_t n = 0, dim = 0, ld = 0, n_pad = 0;
    void *d_pts = nullptr;   // [n_pad][ld], zero padded
    float *d_norm = nullptr; // f32 only: scaled squared norms for the MFMA lower bound
    bool mfma_ok = false;
    void *d_img = nullptr;   // D <= 4096: bf16 tile images of the corpus (bf16_filter.hip)
    float *d_mu = nullptr;   // translation vector of the bf16 tier: the corpus mean per dimension, or zero
    bool centered = false;   // d_mu != 0: translating shrinks the squared norms at least 16x
    bool bf16_ok = false;
    int metric = 0;          // 0 Euclidean, 1 Cosine (d_cnorm = the rows' norms in the index's type; d_img etc. then describe
                             // the rows NORMALISED in f64: finish_index, run_bf16)
    void *d_cnorm = nullptr;
    bool bf16_ci = false;    // "norm in the accumulator" image layout (bf16_filter.hip, bf16_ci_dim)
    double bf16_bmax = 0.0, bf16_dmax = 0.0;  // corpus-wide maxima of the bound's per-row constants (CI layout)
    int n_cu = 256;          // workgroups of the persistent MFMA filter = one per CU
    hipStream_t stream = nullptr;  // construction
    unsigned long long *d_stats = nullptr;  // running device counters {fallback queries, candidates, evaluations}
    // options (set_option must not race with queries on the same handle)
    int engine = PN_ENGINE_AUTO;
    int opt_segments = 0;
    uint64_t index_base = 0;
    int profile = 0;
    int filter_slots = 0;
    int mfma_structure = 0;  // 0 auto, 2 = persistent partition with LDS buffers, 3 = with HBM buffers (1: retired)
    int shared_tau = 1;      // PN_OPT_SHARED_THRESHOLDS: 0 off, 1 auto, >= 2 the rank itself
    int bf16_waves = 0;      // PN_OPT_BF16_WAVES: 0 auto (8-wave main-pass kernel where it applies), 4 = 4-wave kernel
    // Seed model (round 4, seed_model_build): starting thresholds of a k-NN call from per-dimension moments of the corpus,
    // calibrated against the scout's own seeds at build time -- the calls it serves run WITHOUT the scout launch.
    float *d_smodel = nullptr;   // [3][ld] f32: M1_k | 4 (M2_k - M1_k^2) | 4 (M3_k - M2_k M1_k), zero padded
    double sm_c0 = 0.0, sm_v0 = 0.0;    // sum of the second moments / of the fourth central ones (query-independent parts)
    double sm_zgrid[kSeedModelGrid] = {};  // calibrated z at the corpus ranks 16 << j
    double sm_sigma = 0.0;              // standard deviation of ln(rank a model threshold really has / rank it aimed at)
    bool sm_ok = false;
    int seed_model = 1;      // PN_OPT_SEED_MODEL: 1 (default) use it where it was accepted, 0 never
    // state that queries on a shared `const pn_index *` update: internally synchronised by `mu`
    struct Shared {
        std::mutex mu;
        std::vector<Workspace *> free_ws, all_ws;
        CallRec recs[kCallRecs];
        unsigned next_rec = 0;
        uint64_t next_call = 1, stats_call = 0;
        int bf16_level = 0;  // 0 default plan, 1 conservative k', 2 tier off (raised when a call falls back too much)
        // (these two are read by the planner without sh.mu -- a plan may see either side of an update, both are valid --
        // and written under it: atomics, so that the read is not a data race)
        std::atomic<int> seed_model_widen{0};     // calls seeded by the model left a few queries unproven: aim 1.5^this higher (sticky)
        std::atomic<bool> seed_model_off{false};  // a call seeded by the model left too many queries unproven: back to the scout
        int seed_model_off_calls = 0; // scouted calls since then: after kSeedModelRetry of them the model gets another try,
        int seed_model_retries = 0;   // aiming 1.5x higher -- kSeedModelMaxRetries times, then it stays off
        pn_stats stats{};    // host-side part: queries, radius_results, hot_*, last_call_ms
        // the reference's ball tree, built on first use of the introspection API (tree.cpp): published once, built under
        // tree_mu -- never under `mu`, which every query takes (an O(n d log n) host build must not block them)
        std::atomic<HostTree *> tree{nullptr};
        std::mutex tree_mu;
    };
    mutable Shared sh;
};

struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

static int check_device(int device) {
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess || c <= 0)
        return fail(PN_ERR_DEVICE, "no usable GPU (hipGetDeviceCount: %s); this library has no CPU path",
                    hipGetErrorString(e));
    if (device < 0 || device >= c) return fail(PN_ERR_INVALID, "device %d out of range [0,%d)", device, c);
    return PN_OK;
}

// hipStreamCreate / hipStreamDestroy cost 0.4-1 ms each on this runtime (hip-trace, round 3: they were 90 % of a
// 128 x 10 index build), so the library's own streams are recycled through a small per-process pool.  A stream carries
// no state but its ordering, and one is only returned here once everything enqueued on it has completed.  The pool is
// never destroyed (the runtime may already be shutting down when static destructors run).
namespace {
struct StreamPool {
    std::mutex mu;
    std::vector<std::pair<int, hipStream_t>> idle;
};
StreamPool &stream_pool() {
    static StreamPool *p = new StreamPool();
    return *p;
}
constexpr size_t kStreamPoolMax = 32;
}  // namespace
static hipStream_t pooled_stream_acquire(int device) {   // the caller has made `device` current
    {
        StreamPool &p = stream_pool();
        std::lock_guard<std::mutex> lk(p.mu);
        for (size_t i = 0; i < p.idle.size(); ++i)
            if (p.idle[i].first == device) {
                hipStream_t s = p.idle[i].second;
                p.idle[i] = p.idle.back();
                p.idle.pop_back();
                return s;
            }
    }
    hipStream_t s = nullptr;
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return nullptr;
    return s;
}
static void pooled_stream_release(int device, hipStream_t s) {   // `s` is idle
    if (!s) return;
    {
        StreamPool &p = stream_pool();
        std::lock_guard<std::mutex> lk(p.mu);
        if (p.idle.size() < kStreamPoolMax) {
            p.idle.emplace_back(device, s);
            return;
        }
    }
    (void)hipStreamDestroy(s);
}

static size_t pick_ld(size_t dim) {
    // D <= 128: one of the row lengths the MFMA filter is instantiated for; else a multiple of 8
    const size_t set[] = {8, 16, 32, 64, 96, 128};
    for (size_t v : set)
        if (dim <= v) return v;
    return round_up(dim, 128);  // wide rows are processed in 128-coordinate slabs
}

static float mfma_alpha(size_t dim) {
    // (2(D+2)+8) * 2^-24: covers the (D+2)-step MFMA fma chain twice (see mfma_filter_v2.hip)
    return (float)((2.0 * (double)(dim + 2) + 8.0) * 5.9604644775390625e-08);
}

// Temporaries of an index build: freed on EVERY return path (round 3's finish_index leaked d_flag / d_bad / d_sums / d_st
// on each HIPCHK early return -- VERDICT r3).  The host words live in pinned memory taken from a small per-process pool
// (hipHostMalloc costs 0.1-0.3 ms, a 128 x 10 build 0.15 ms in all).
namespace {
struct DevTmp {
    void *p = nullptr;
    ~DevTmp() {
        if (p) (void)hipFree(p);
    }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 256); }
};
struct PinnedPool {
    std::mutex mu;
    std::vector<void *> idle;
};
PinnedPool &pinned_pool() {
    static PinnedPool *p = new PinnedPool();  // never destroyed (see stream_pool)
    return *p;
}
constexpr size_t kPinnedBlock = 256;
struct PinnedBlock {  // kPinnedBlock bytes of pinned host memory
    void *h = nullptr;
    ~PinnedBlock() {
        if (!h) return;
        PinnedPool &pp = pinned_pool();
        std::lock_guard<std::mutex> lk(pp.mu);
        if (pp.idle.size() < 16) pp.idle.push_back(h);
        else (void)hipHostFree(h);
    }
    hipError_t acquire() {
        {
            PinnedPool &pp = pinned_pool();
            std::lock_guard<std::mutex> lk(pp.mu);
            if (!pp.idle.empty()) {
                h = pp.idle.back();
                pp.idle.pop_back();
                return hipSuccess;
            }
        }
        return hipHostMalloc(&h, kPinnedBlock, hipHostMallocDefault);
    }
};
}  // namespace

// Device words of a build that the host looks at, copied to pinned memory in front of each of the build's (at most two)
// stream synchronisations: the four row statistics, then {non-finite row norm, bad row for the bf16 tier, translated}
struct BuildWords {
    double st[4];
    uint32_t w[4];
    float selftest;  // bf16_selftest_kernel's result (first bf16 index of the process on this device)
    uint32_t pad;
};
// Verdict of the matrix core's accumulation self-test per device: -1 not run yet, 0 refused, 1 passed (bf16_filter.hip,
// bf16_selftest_kernel: the proof's allowance g = 2^-13 is a measured property of the hardware, and the product checks
// it before it first relies on it -- not only a pytest)
constexpr int kMaxDevices = 64;
static std::atomic<int> g_bf16_hw[kMaxDevices];
static std::atomic<bool> g_bf16_hw_init{false};
static std::mutex g_bf16_hw_mu;
static std::atomic<int> &bf16_hw_state(int device) {
    if (!g_bf16_hw_init.load(std::memory_order_acquire)) {
        std::lock_guard<std::mutex> lk(g_bf16_hw_mu);
        if (!g_bf16_hw_init.load(std::memory_order_relaxed)) {
            for (auto &a : g_bf16_hw) a.store(-1, std::memory_order_relaxed);
            g_bf16_hw_init.store(true, std::memory_order_release);
        }
    }
    return g_bf16_hw[device >= 0 && device < kMaxDevices ? device : 0];
}
constexpr float kBf16SelftestLimit = 0.02f;  // of the allowance (measured on gfx950: <= 0.0016)

template <typename RT>
static int seed_model_build(pn_index *ix, const RT *rows, hipStream_t s);
template <typename T>
static int finish_index(pn_index *ix, const T *d_src, size_t row_stride, hipStream_t s) {
    // d_src: device rows [n][row_stride] (inner stride 1) -> padded layout + norms (+ the filter tiers' images).
    // Host round trips: ONE stream synchronisation at the end, plus one in the middle only where the image's layout
    // depends on the row statistics (round 3: six, each behind a small copy).
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, ix->device) == hipSuccess && prop.multiProcessorCount > 0)
            ix->n_cu = prop.multiProcessorCount;
    }
    HIPCHK(hipMalloc((void **)&ix->d_stats, kPnStatWords * sizeof(unsigned long long)));
    HIPCHK(hipMemsetAsync(ix->d_stats, 0, kPnStatWords * sizeof(unsigned long long), s));
    const size_t bytes = ix->n_pad * ix->ld * sizeof(T);
    HIPCHK(hipMalloc(&ix->d_pts, bytes ? bytes : 256));
    if (sizeof(T) == 4)
        HIPCHK(launch_pack_rows_f32((const float *)d_src, ix->n, ix->dim, row_stride, (float *)ix->d_pts, ix->n_pad,
                                    ix->ld, s));
    else
        HIPCHK(launch_pack_rows_f64((const double *)d_src, ix->n, ix->dim, row_stride, (double *)ix->d_pts,
                                    ix->n_pad, ix->ld, s));
    ix->mfma_ok = false;
    ix->bf16_ok = false;
    ix->bf16_ci = false;
    const bool cosine = ix->metric == 1;
    if (cosine) {
        // Cosine: the rows' norms (sequential sum of squares + sqrt, in T) for the exact evaluation of Cosine::distance
        HIPCHK(hipMalloc(&ix->d_cnorm, ix->n_pad * sizeof(T)));
        HIPCHK(hipMemsetAsync(ix->d_cnorm, 0, ix->n_pad * sizeof(T), s));
        if (sizeof(T) == 4)
            HIPCHK(launch_cosine_norms_f32((const float *)ix->d_pts, ix->n, (int)ix->dim, ix->ld, (float *)ix->d_cnorm, s));
        else
            HIPCHK(launch_cosine_norms_f64((const double *)ix->d_pts, ix->n, (int)ix->dim, ix->ld, (double *)ix->d_cnorm, s));
    }
    // device scratch: column sums [dim + 1] | BuildWords
    DevTmp scr, nrm;
    PinnedBlock hw;
    const size_t sums_bytes = (ix->dim + 1) * sizeof(double);
    HIPCHK(scr.alloc(sums_bytes + sizeof(BuildWords)));
    HIPCHK(hipMemsetAsync(scr.p, 0, sums_bytes + sizeof(BuildWords), s));
    HIPCHK(hw.acquire());
    static_assert(sizeof(BuildWords) <= kPinnedBlock, "host words fit one pinned block");
    double *d_sums = (double *)scr.p;
    BuildWords *d_bw = (BuildWords *)((char *)scr.p + sums_bytes);
    BuildWords *h_bw = (BuildWords *)hw.h;
    uint32_t *d_w = (uint32_t *)((char *)d_bw + offsetof(BuildWords, w));
    double *d_st = (double *)((char *)d_bw + offsetof(BuildWords, st));
    auto read_words = [&]() -> int {  // device words -> pinned memory, then the host waits
        HIPCHK(hipMemcpyAsync(h_bw, d_bw, sizeof(BuildWords), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        return PN_OK;
    };
    if constexpr (sizeof(T) == 4) {
        if (!cosine) {  // the f32 MFMA tier's scaled squared norms (Euclidean only)
            HIPCHK(hipMalloc((void **)&ix->d_norm, ix->n_pad * sizeof(float)));
            HIPCHK(launch_row_norms_f32((const float *)ix->d_pts, ix->n_pad, ix->n, (int)ix->dim, ix->ld,
                                        mfma_alpha(ix->dim), ix->d_norm, d_w + 0, s));
        }
    }
    // bf16 tier: f32 AND f64 indexes (the images are built from the index's own coordinates in f64 arithmetic), Euclidean
    // AND Cosine (round 4: the same images over the rows normalised in f64 -- |q~ - p~|^2 = 2 (1 - cos); the rows of a
    // Cosine index are an f64 array for the duration of the build, twice an f32 corpus' size, and the <double> kernels
    // of the Euclidean tier do the rest)
    const bool try_bf16 = bf16_supported((int)ix->dim) && ix->n < 0xFFFFFFF0ull && ix->n >= 64;
    if (try_bf16) {
        const void *rows = ix->d_pts;  // what the images are built from, [n_pad][ld] of T -- or of double (Cosine)
        if (cosine) {
            HIPCHK(nrm.alloc(ix->n_pad * ix->ld * sizeof(double)));
            HIPCHK(launch_cos_normalize_rows<T>((const T *)ix->d_pts, ix->n, ix->n_pad, (int)ix->dim, ix->ld, (double *)nrm.p,
                                                ix->ld, d_w + 1, false, s));
            rows = nrm.p;
        }
        auto col_sums = [&]() {
            return cosine ? launch_bf16_column_sums<double>((const double *)rows, ix->n, (int)ix->dim, ix->ld, d_sums, s)
                          : launch_bf16_column_sums<T>((const T *)rows, ix->n, (int)ix->dim, ix->ld, d_sums, s);
        };
        auto row_stats = [&]() {
            return cosine ? launch_bf16_row_stats<double>((const double *)rows, ix->d_mu, ix->n, (int)ix->dim, ix->ld, d_st, s)
                          : launch_bf16_row_stats<T>((const T *)rows, ix->d_mu, ix->n, (int)ix->dim, ix->ld, d_st, s);
        };
        auto pack = [&]() {
            return cosine ? launch_bf16_pack_corpus<double>((const double *)rows, ix->d_mu, ix->n, (int)ix->dim, ix->ld,
                                                            ix->d_img, d_w + 1, ix->bf16_ci, s)
                          : launch_bf16_pack_corpus<T>((const T *)rows, ix->d_mu, ix->n, (int)ix->dim, ix->ld, ix->d_img,
                                                       d_w + 1, ix->bf16_ci, s);
        };
        // translation vector = per-dimension mean when that shrinks the squared norms enough: f64 sums, decided on the device
        HIPCHK(hipMalloc((void **)&ix->d_mu, ix->dim * sizeof(float)));
        HIPCHK(col_sums());
#ifdef PN_DIAG_NO_CENTER
        const bool never_center = true;
#else
        const bool never_center = false;
#endif
        HIPCHK(launch_bf16_decide_mu(d_sums, ix->n, (int)ix->dim, never_center, ix->d_mu, d_w + 2, s));
#ifndef PN_DIAG_NO_CI
        if (bf16_ci_candidate((int)ix->dim)) {
            // the extra columns would cost an MFMA step of their own: drop them when the corpus-wide maxima of the
            // bound's per-row constants are close to their means (homogeneous row norms).  The image's layout depends
            // on it, so this is the build's one host round trip in mid-stream.
            HIPCHK(row_stats());
            PNCHK(read_words());
            const double *h_st = h_bw->st;
            const double bmean = h_st[2] / (double)ix->n, dmean = h_st[3] / (double)ix->n;
            ix->bf16_ci = h_st[0] > 0.0 && h_st[1] > 0.0 && h_st[0] <= 1.3 * bmean && h_st[1] <= 1.3 * dmean &&
                          h_st[0] < 1e30 && h_st[1] < 1e30;
            ix->bf16_bmax = h_st[0];
            ix->bf16_dmax = h_st[1];
        }
#endif
        HIPCHK(hipMalloc(&ix->d_img, bf16_image_bytes(ix->n, (int)ix->dim, ix->bf16_ci)));
        HIPCHK(pack());
        if (bf16_hw_state(ix->device).load() < 0)  // first bf16 index on this device: the self-test rides along
            HIPCHK(launch_bf16_selftest((float *)((char *)d_bw + offsetof(BuildWords, selftest)), s));
    }
    PNCHK(read_words());
    if constexpr (sizeof(T) == 4)
        ix->mfma_ok = !cosine && (h_bw->w[0] == 0) && mfma_supported((int)ix->dim, ix->ld) && ix->n < 0xFFFFFFF0ull;
    if (try_bf16) {
        std::atomic<int> &hw = bf16_hw_state(ix->device);
        if (hw.load() < 0) hw.store(h_bw->selftest < kBf16SelftestLimit ? 1 : 0);  // (NaN compares false: refused)
        ix->centered = h_bw->w[2] != 0;
        ix->bf16_ok = h_bw->w[1] == 0;
        if (hw.load() == 0) {
            ix->bf16_ok = false;
            (void)fail(PN_OK, "bf16 tier refused on device %d: the matrix core's accumulation error is %.4f of the 2^-13 "
                              "allowance the bound makes (limit %.2f)", ix->device, (double)h_bw->selftest,
                       (double)kBf16SelftestLimit);
        }
        if (!ix->bf16_ok) {
            (void)hipFree(ix->d_img);
            ix->d_img = nullptr;
            ix->bf16_ci = false;
        }
    }
    if (cosine) {
        if (nrm.p) PNCHK(seed_model_build<double>(ix, (const double *)nrm.p, s));  // (over the normalised rows)
    } else {
        PNCHK(seed_model_build<T>(ix, (const T *)ix->d_pts, s));
    }
    return PN_OK;
}

static int validate_create(const void *points, size_t n_rows, size_t n_cols, ptrdiff_t row_stride,
                           ptrdiff_t col_stride, pn_index **out) {
    if (!out) return fail(PN_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (n_rows == 0) return fail(PN_ERR_EMPTY, "array is empty");  // src/ball_tree.rs:44-46
    // only the inner stride is inspected, like row(0).is_standard_layout() (src/ball_tree.rs:47)
    if (n_cols > 1 && col_stride != 1) return fail(PN_ERR_NOT_CONTIGUOUS, "array is not contiguous in memory");
    if (n_cols == 0 && n_rows >= 2) return fail(PN_ERR_EMPTY_MATRIX, "empty matrix");  // src/ball_tree.rs:582
    if (!points && n_cols) return fail(PN_ERR_INVALID, "points is NULL");
    if (row_stride < 0) return fail(PN_ERR_UNSUPPORTED, "negative row stride");
    if (n_rows > 1 && (size_t)row_stride < n_cols && n_cols > 0 && row_stride != 0)
        return fail(PN_ERR_UNSUPPORTED, "overlapping rows (row_stride < n_cols)");
    if (n_rows >= 0xFFFFFFF0ull) return fail(PN_ERR_UNSUPPORTED, "more than 2^32-16 rows per index: shard the corpus");
    return PN_OK;
}

template <typename T>
static int create_from_host(const T *points, size_t n_rows, size_t n_cols, ptrdiff_t row_stride,
                            ptrdiff_t col_stride, int device, pn_index **out, int metric = 0) {
    PNCHK(validate_create(points, n_rows, n_cols, row_stride, col_stride, out));
    PNCHK(check_device(device));
    DeviceGuard g(device);
    if (!g.ok) return fail(PN_ERR_DEVICE, "hipSetDevice(%d) failed", device);
    pn_index *ix = new (std::nothrow) pn_index();
    if (!ix) return fail(PN_ERR_NOMEM, "host allocation failed");
    ix->device = device;
    ix->elem_bytes = (int)sizeof(T);
    ix->metric = metric;
    ix->n = n_rows;
    ix->dim = n_cols;
    ix->ld = pick_ld(n_cols);
    ix->n_pad = round_up(n_rows, kRowPad);
    int rc = PN_OK;
    T *d_tmp = nullptr;
    do {
        if (!(ix->stream = pooled_stream_acquire(device))) {
            rc = fail(PN_ERR_DEVICE, "hipStreamCreate failed");
            break;
        }
        const size_t tmp_elems = n_rows * (n_cols ? n_cols : 1);
        hipError_t e = hipMalloc((void **)&d_tmp, tmp_elems * sizeof(T));
        if (e != hipSuccess) { rc = fail(PN_ERR_NOMEM, "hipMalloc staging: %s", hipGetErrorString(e)); break; }
        if (n_cols) {
            // the host array is read once, here, and never retained (CowArray borrow, src/ball_tree.rs:42)
            if (n_rows == 1 || (size_t)row_stride == n_cols)
                e = hipMemcpy(d_tmp, points, tmp_elems * sizeof(T), hipMemcpyHostToDevice);
            else if (row_stride == 0) {
                for (size_t r = 0; r < n_rows && e == hipSuccess; ++r)
                    e = hipMemcpy(d_tmp + r * n_cols, points, n_cols * sizeof(T), hipMemcpyHostToDevice);
            } else
                e = hipMemcpy2D(d_tmp, n_cols * sizeof(T), points, (size_t)row_stride * sizeof(T),
                                n_cols * sizeof(T), n_rows, hipMemcpyHostToDevice);
            if (e != hipSuccess) { rc = fail(PN_ERR_DEVICE, "H2D copy: %s", hipGetErrorString(e)); break; }
        }
        rc = finish_index<T>(ix, d_tmp, n_cols, ix->stream);
    } while (0);
    if (rc == PN_OK) {   // finish_index ends with the stream drained: the build stream goes back to the pool
        pooled_stream_release(device, ix->stream);
        ix->stream = nullptr;
    }
    if (d_tmp) (void)hipFree(d_tmp);
    if (rc != PN_OK) {
        pn_index_destroy(ix);
        return rc;
    }
    *out = ix;
    return PN_OK;
}

extern "C" int pn_index_create_f32(const float *points, size_t n_rows, size_t n_cols, ptrdiff_t row_stride,
                                   ptrdiff_t col_stride, int device, pn_index **out) {
    return create_from_host<float>(points, n_rows, n_cols, row_stride, col_stride, device, out);
}
extern "C" int pn_index_create_f64(const double *points, size_t n_rows, size_t n_cols, ptrdiff_t row_stride,
                                   ptrdiff_t col_stride, int device, pn_index **out) {
    return create_from_host<double>(points, n_rows, n_cols, row_stride, col_stride, device, out);
}

// BallTree::new(points, Cosine) (src/ball_tree.rs:38-63 with src/distance.rs:76-122): same validation; queries are
// exact scans under Cosine::distance (see the header for how this differs from the reference's pruned walk)
extern "C" int pn_index_create_cosine_f32(const float *points, size_t n_rows, size_t n_cols, ptrdiff_t row_stride,
                                          ptrdiff_t col_stride, int device, pn_index **out) {
    return create_from_host<float>(points, n_rows, n_cols, row_stride, col_stride, device, out, 1);
}
extern "C" int pn_index_create_cosine_f64(const double *points, size_t n_rows, size_t n_cols, ptrdiff_t row_stride,
                                          ptrdiff_t col_stride, int device, pn_index **out) {
    return create_from_host<double>(points, n_rows, n_cols, row_stride, col_stride, device, out, 1);
}

template <typename T>
static int create_from_device(const T *d_points, size_t n_rows, size_t n_cols, size_t row_stride, int device, void *stream,
                              pn_index **out) {
    PNCHK(validate_create(d_points, n_rows, n_cols, (ptrdiff_t)row_stride, 1, out));
    PNCHK(check_device(device));
    DeviceGuard g(device);
    if (!g.ok) return fail(PN_ERR_DEVICE, "hipSetDevice(%d) failed", device);
    pn_index *ix = new (std::nothrow) pn_index();
    if (!ix) return fail(PN_ERR_NOMEM, "host allocation failed");
    ix->device = device;
    ix->elem_bytes = (int)sizeof(T);
    ix->n = n_rows;
    ix->dim = n_cols;
    ix->ld = pick_ld(n_cols);
    ix->n_pad = round_up(n_rows, kRowPad);
    int rc = PN_OK;
    if (!(ix->stream = pooled_stream_acquire(device))) rc = fail(PN_ERR_DEVICE, "hipStreamCreate failed");
    if (rc == PN_OK) {
        // order after the producer of d_points on `stream`
        hipStream_t src = (hipStream_t)stream;
        if (hipStreamSynchronize(src) != hipSuccess) rc = fail(PN_ERR_DEVICE, "stream sync failed");
    }
    if (rc == PN_OK) rc = finish_index<T>(ix, d_points, row_stride, ix->stream);
    if (rc == PN_OK) {
        pooled_stream_release(device, ix->stream);
        ix->stream = nullptr;
    }
    if (rc != PN_OK) {
        pn_index_destroy(ix);
        return rc;
    }
    *out = ix;
    return PN_OK;
}
extern "C" int pn_index_create_device_f32(const float *d_points, size_t n_rows, size_t n_cols, size_t row_stride,
                                          int device, void *stream, pn_index **out) {
    return create_from_device<float>(d_points, n_rows, n_cols, row_stride, device, stream, out);
}
extern "C" int pn_index_create_device_f64(const double *d_points, size_t n_rows, size_t n_cols, size_t row_stride,
                                          int device, void *stream, pn_index **out) {
    return create_from_device<double>(d_points, n_rows, n_cols, row_stride, device, stream, out);
}

// Everything this handle has enqueued -- on its own streams or on callers' (the *_device entry points) -- has finished:
// every call holds a workspace for its duration and leaves an end-of-use event on the stream it ran on (ws_release), so
// waiting for the workspaces' events is waiting for the handle, and for nothing else on the device (round 3 called
// hipDeviceSynchronize here: a device-wide wait behind a per-handle call).  `all`: also workspaces that are leased right
// now (pn_index_destroy: no call may be in progress); else only released ones (a concurrent call is simply not waited for).
static void handle_wait(const pn_index *ix, bool all) {
    std::vector<hipEvent_t> evs;
    {
        std::lock_guard<std::mutex> lk(ix->sh.mu);
        for (Workspace *ws : (all ? ix->sh.all_ws : ix->sh.free_ws))
            if (ws->in_flight && ws->done) evs.push_back(ws->done);
    }
    for (hipEvent_t e : evs) (void)hipEventSynchronize(e);
}

extern "C" void pn_index_destroy(pn_index *ix) {
    if (!ix) return;
    DeviceGuard g(ix->device);
    // queries may have been enqueued on caller streams (the *_device entry points): nothing of this
    // index may be freed while any of them is still running
    handle_wait(ix, true);
    if (ix->stream) (void)hipStreamSynchronize(ix->stream);
    for (Workspace *ws : ix->sh.all_ws) {
        for (DevBuf *b : ws->all) b->release();
        ws->free_retired();
        for (hipEvent_t e : ws->ev_r)
            if (e) (void)hipEventDestroy(e);
        if (ws->pin_in) (void)hipHostFree(ws->pin_in);
        if (ws->pin_out) (void)hipHostFree(ws->pin_out);
        if (ws->done) (void)hipEventDestroy(ws->done);
        pooled_stream_release(ix->device, ws->stream);
        delete ws;
    }
    for (CallRec &r : ix->sh.recs) {
        for (hipEvent_t e : r.ev)
            if (e) (void)hipEventDestroy(e);
        if (r.done) (void)hipEventDestroy(r.done);
        if (r.h_nflag) (void)hipHostFree(r.h_nflag);
    }
    if (ix->d_pts) (void)hipFree(ix->d_pts);
    if (ix->d_img) (void)hipFree(ix->d_img);
    if (ix->d_mu) (void)hipFree(ix->d_mu);
    if (ix->d_smodel) (void)hipFree(ix->d_smodel);
    if (ix->d_norm) (void)hipFree(ix->d_norm);
    if (ix->d_cnorm) (void)hipFree(ix->d_cnorm);
    if (ix->d_stats) (void)hipFree(ix->d_stats);
    pooled_stream_release(ix->device, ix->stream);
    if (HostTree *t = ix->sh.tree.load()) host_tree_free(t);
    delete ix;
}

// ---------------------------------------------------------------------------
// workspace pool and deferred per-call records
// ---------------------------------------------------------------------------
// Takes a workspace for a call that will enqueue on stream `s` (nullptr: the workspace's own stream is used and
// returned).  A workspace last used on another stream is first ordered behind that use.
static int ws_acquire(const pn_index *ix, hipStream_t *s, bool own_stream, Workspace **out) {
    Workspace *ws = nullptr;
    {
        std::lock_guard<std::mutex> lk(ix->sh.mu);
        if (!ix->sh.free_ws.empty()) {
            ws = ix->sh.free_ws.back();
            ix->sh.free_ws.pop_back();
        }
    }
    if (!ws) {
        ws = new (std::nothrow) Workspace();
        if (!ws) return fail(PN_ERR_NOMEM, "host allocation failed");
        if (hipEventCreateWithFlags(&ws->done, hipEventDisableTiming) != hipSuccess) {
            delete ws;
            return fail(PN_ERR_DEVICE, "workspace creation failed: %s", hipGetErrorString(hipGetLastError()));
        }
        std::lock_guard<std::mutex> lk(ix->sh.mu);
        ix->sh.all_ws.push_back(ws);
    }
    if (own_stream) {
        // only the host entry points run on a stream of the library's own; calls on a caller's stream never need one
        if (!ws->stream && !(ws->stream = pooled_stream_acquire(ix->device))) {
            std::lock_guard<std::mutex> lk(ix->sh.mu);
            ix->sh.free_ws.push_back(ws);
            return fail(PN_ERR_DEVICE, "workspace stream creation failed: %s", hipGetErrorString(hipGetLastError()));
        }
        *s = ws->stream;
    }
    // allocations outgrown by an earlier call: free them once everything that call enqueued has finished
    if (!ws->retired.empty() && (!ws->in_flight || hipEventQuery(ws->done) == hipSuccess)) ws->free_retired();
    if (ws->in_flight && ws->last_stream != *s) {
        if (hipStreamWaitEvent(*s, ws->done, 0) != hipSuccess) {
            std::lock_guard<std::mutex> lk(ix->sh.mu);
            ix->sh.free_ws.push_back(ws);
            return fail(PN_ERR_DEVICE, "hipStreamWaitEvent failed");
        }
    }
    *out = ws;
    return PN_OK;
}
static void ws_release(const pn_index *ix, Workspace *ws, hipStream_t s) {
    ws->in_flight = hipEventRecord(ws->done, s) == hipSuccess;
    if (!ws->in_flight) (void)hipStreamSynchronize(s);  // cannot mark the end of the call: wait for it instead
    ws->last_stream = s;
    std::lock_guard<std::mutex> lk(ix->sh.mu);
    ix->sh.free_ws.push_back(ws);
}
struct WsLease {  // releases on every return path
    const pn_index *ix;
    Workspace *ws = nullptr;
    hipStream_t s = nullptr;
    explicit WsLease(const pn_index *i) : ix(i) {}
    ~WsLease() {
        if (ws) ws_release(ix, ws, s);
    }
};

// folds a finished record into the host-side statistics and the tier plan (caller holds sh.mu)
// What a finished bf16-tier call teaches the handle about its seed model (under sh.mu; true: the call was the model's
// business, the plan's level is left alone).  A call seeded by the model that leaves more than one query in 128 unproven:
// the model does not fit these queries -- back to the scout; more than one in 1024: aim 1.5x higher, and off after four
// such steps.  One odd batch must not cost the handle its model for good: after kSeedModelRetry scouted calls the model is
// tried again, aiming 1.5x higher, at most kSeedModelMaxRetries times.
static bool seed_model_feedback(pn_index::Shared &sh, bool model_seed, size_t nf, size_t nq) {
    if (model_seed && nf * 128 > nq && nq >= 64) {
        sh.seed_model_off = true;
        sh.seed_model_off_calls = 0;
        return true;
    }
    if (model_seed && nf * 1024 > nq && nq >= 256) {
        if (sh.seed_model_widen.fetch_add(1) + 1 > 4) {
            sh.seed_model_off = true;
            sh.seed_model_off_calls = 0;
        }
        return true;
    }
    if (!model_seed && sh.seed_model_off && sh.seed_model_retries < kSeedModelMaxRetries &&
        ++sh.seed_model_off_calls >= kSeedModelRetry) {
        sh.seed_model_off = false;
        sh.seed_model_off_calls = 0;
        sh.seed_model_retries += 1;
        if (sh.seed_model_widen.load() < 4) sh.seed_model_widen.fetch_add(1);
        // (not the model's business otherwise: a scouted call that defeats the plan still raises the level below)
    }
    return false;
}
static void rec_resolve(const pn_index *ix, CallRec &r) {
    pn_index::Shared &sh = ix->sh;
    if (r.prof) {
        float ms = 0, a = 0, b = 0;
        if (r.hot) {
            if (r.two_launches) {
                // The dominant kernel ran twice (scout-only launch, main launch) inside ONE bracket ev0..ev1, with the
                // one-wave-per-query seed kernel (~18 us at C2, < 1 %) between them: every event record costs the stream
                // ~6 us, and separate brackets (round 2) were four of them per step.  hot_ms is therefore an upper
                // bound of the two launches' time.
                (void)a;
                (void)b;
                if (hipEventElapsedTime(&ms, r.ev[0], r.ev[1]) == hipSuccess) {
                    sh.stats.hot_ms += ms;
                    sh.stats.hot_launches += 2;
                }
            } else if (hipEventElapsedTime(&ms, r.ev[0], r.ev[1]) == hipSuccess) {
                sh.stats.hot_ms += ms;
                sh.stats.hot_launches += 1;
            }
        }
        if (r.prof > 1 && hipEventElapsedTime(&ms, r.ev[2], r.ev[3]) == hipSuccess) {
            if (sh.stats_call != r.call_id) {
                sh.stats_call = r.call_id;
                sh.stats.last_call_ms = 0.0;
            }
            sh.stats.last_call_ms += ms;
        }
    }
    if (r.has_flag && r.bf16_tier) {
        // a call that handed more than 1/16 of its queries to the next tier: this corpus defeats the current plan
        // -- widen it, then turn the tier off (sticky; takes effect from the next call on)
        const size_t nf = *r.h_nflag;
#if !defined(PN_DIAG_BF_NOSLOW) && !defined(PN_DIAG_BF_NOSTORE) && !defined(PN_DIAG_BF_NOBARRIER) && !defined(PN_DIAG_BF_NOWAIT) && !defined(PN_DIAG_BF_NOAPPEND)  // timing-only builds flag queries by design
        // (a call seeded by the index's model instead of a scout launch: more than one query in 128 unproven means the
        // model does not fit these queries -- back to the scout, for good; the plan itself is not to blame)
        if (seed_model_feedback(sh, r.model_seed, nf, r.nq)) {
        } else if (nf * 16 > r.nq && r.nq >= 64 && ix->filter_slots == 0 && sh.bf16_level < 2) sh.bf16_level += 1;
#else
        (void)nf;
#endif
    }
    r.pending = false;
}
// picks up every record whose chunk has finished; wait = true: all of them (the caller has synchronised the device)
static void recs_collect(const pn_index *ix, bool wait) {
    for (CallRec &r : ix->sh.recs) {
        if (!r.pending) continue;
        if (wait)
            (void)hipEventSynchronize(r.done);
        else if (hipEventQuery(r.done) != hipSuccess)
            continue;
        rec_resolve(ix, r);
    }
}
// next record of the ring (its previous use, 16 chunks ago, is waited for if it is still running)
static int rec_begin(const pn_index *ix, uint64_t call_id, size_t nq, CallRec **out) {
    std::lock_guard<std::mutex> lk(ix->sh.mu);
    CallRec *rp = nullptr;
    for (int i = 0; i < kCallRecs && !rp; ++i) {  // skip records another thread's chunk is still filling in
        CallRec &c = ix->sh.recs[ix->sh.next_rec++ % kCallRecs];
        if (!c.busy) rp = &c;
    }
    if (!rp) return fail(PN_ERR_UNSUPPORTED, "more than %d query chunks in flight on one index", kCallRecs);
    CallRec &r = *rp;
    if (r.pending) {
        (void)hipEventSynchronize(r.done);
        rec_resolve(ix, r);
    }
    // (each resource on its own: a record whose event exists but whose pinned word does not must not be used)
    if (!r.h_nflag) HIPCHK(hipHostMalloc((void **)&r.h_nflag, 64, hipHostMallocDefault));
    if (!r.done) HIPCHK(hipEventCreateWithFlags(&r.done, hipEventDisableTiming));
    if (ix->profile && !r.ev[0])
        for (hipEvent_t &e : r.ev) HIPCHK(hipEventCreate(&e));
    r.nq = nq;
    r.call_id = call_id;
    r.prof = ix->profile;
    r.two_launches = r.has_flag = r.bf16_tier = r.hot = r.model_seed = false;
    r.busy = true;
    *r.h_nflag = 0;
    *out = &r;
    return PN_OK;
}
static int rec_end(const pn_index *ix, CallRec *r, hipStream_t s) {
    const hipError_t e = hipEventRecord(r->done, s);
    std::lock_guard<std::mutex> lk(ix->sh.mu);
    r->busy = false;
    r->pending = e == hipSuccess;
    if (e != hipSuccess) return fail(PN_ERR_DEVICE, "hipEventRecord: %s", hipGetErrorString(e));
    return PN_OK;
}

// A record is busy from rec_begin to rec_end; a chunk that fails in between (out of memory, a launch error) must give
// its slot back, or sixteen failed chunks would leave the handle without records for good.
struct RecGuard {
    const pn_index *ix;
    CallRec *r = nullptr;
    explicit RecGuard(const pn_index *i) : ix(i) {}
    ~RecGuard() {
        if (!r) return;
        std::lock_guard<std::mutex> lk(ix->sh.mu);
        r->busy = false;
        r->pending = false;
    }
};

extern "C" int pn_index_info(const pn_index *ix, pn_info *out) {
    if (!ix || !out) return fail(PN_ERR_INVALID, "NULL argument");
    out->n_points = ix->n;
    out->dim = ix->dim;
    out->row_stride_device = ix->ld;
    out->elem_bytes = ix->elem_bytes;
    out->device = ix->device;
    out->mfma_eligible = ix->mfma_ok ? 1 : 0;
    out->bf16_eligible = ix->bf16_ok ? 1 : 0;
    out->bf16_layout = !ix->bf16_ok ? 0 : ix->bf16_ci ? 2 : 1;
    out->seed_model = ix->sm_ok ? 1 : 0;
    return PN_OK;
}

extern "C" int pn_index_set_option(pn_index *ix, int option, int64_t value) {
    if (!ix) return fail(PN_ERR_INVALID, "index is NULL");
    std::lock_guard<std::mutex> lk(ix->sh.mu);
    switch (option) {
        case PN_OPT_ENGINE:
            if (value < PN_ENGINE_AUTO || value > PN_ENGINE_BF16) return fail(PN_ERR_INVALID, "bad engine %lld", (long long)value);
            if (ix->metric == 1 && value == PN_ENGINE_MFMA)
                return fail(PN_ERR_UNSUPPORTED, "a Cosine index is served by the bf16 filter or the exact scan");
            if (value == PN_ENGINE_BF16 && !ix->bf16_ok)
                return fail(PN_ERR_UNSUPPORTED, "the bf16 filter cannot serve this index (D > 4096, fewer than 64 rows or out-of-range values)");
            if (value == PN_ENGINE_MFMA && !ix->mfma_ok)
                return fail(PN_ERR_UNSUPPORTED, "the MFMA filter cannot serve this index (f64, non-finite norms or unsupported shape)");
            ix->engine = (int)value;
            ix->sh.bf16_level = 0;
            return PN_OK;
        case PN_OPT_SEGMENTS:
            if (value < 0 || value > 4096) return fail(PN_ERR_INVALID, "bad segment count");
            ix->opt_segments = (int)value;
            return PN_OK;
        case PN_OPT_INDEX_BASE: ix->index_base = (uint64_t)value; return PN_OK;
        case PN_OPT_PROFILE: ix->profile = value < 0 ? 0 : value > 2 ? 2 : (int)value; return PN_OK;
        case PN_OPT_FILTER_SLOTS:
            if (value < 0 || value > 960) return fail(PN_ERR_INVALID, "bad filter slot count");
            ix->filter_slots = (int)value;
            return PN_OK;
        case PN_OPT_MFMA_STRUCTURE:
            if (value == 1) return fail(PN_ERR_INVALID, "structure 1 (the (query tile x segment) grid) was retired in round 4");
            if (value < 0 || value > 3) return fail(PN_ERR_INVALID, "bad structure");
            ix->mfma_structure = (int)value;
            return PN_OK;
        case PN_OPT_SHARED_THRESHOLDS:
            if (value < 0 || value > 4096) return fail(PN_ERR_INVALID, "bad shared-threshold rank");
            ix->shared_tau = (int)value;
            return PN_OK;
        case PN_OPT_BF16_WAVES:
            if (value != 0 && value != 4 && value != 8) return fail(PN_ERR_INVALID, "bad wave count (0, 4 or 8)");
            ix->bf16_waves = (int)value;
            return PN_OK;
        case PN_OPT_SEED_MODEL:
            if (value != 0 && value != 1) return fail(PN_ERR_INVALID, "bad seed-model switch (0 or 1)");
            ix->seed_model = (int)value;
            return PN_OK;
        default: return fail(PN_ERR_INVALID, "unknown option %d", option);
    }
}

// Statistics are collected lazily: the query entry points never read anything back.  This call waits for the
// device, folds the finished calls' records in and reads the running device counters.
extern "C" int pn_index_get_stats(const pn_index *ix, pn_stats *out, int reset) {
    if (!ix || !out) return fail(PN_ERR_INVALID, "NULL argument");
    DeviceGuard g(ix->device);
    if (!g.ok) return fail(PN_ERR_DEVICE, "hipSetDevice(%d) failed", ix->device);
    handle_wait(ix, false);  // this handle's calls, not the device (see handle_wait)
    std::lock_guard<std::mutex> lk(ix->sh.mu);
    recs_collect(ix, true);
    unsigned long long h[kPnStatWords] = {0};  // [0] fallback queries, then kPnStatSlots pairs {candidates, evaluations}
    if (ix->d_stats) HIPCHK(hipMemcpy(h, ix->d_stats, sizeof h, hipMemcpyDeviceToHost));
    *out = ix->sh.stats;
    out->fallback_queries += h[0];
    for (int i = 0; i < kPnStatSlots; ++i) {
        out->candidates += h[4 + 2 * i];
        out->evaluations += h[5 + 2 * i];
    }
    if (reset) {
        ix->sh.stats = pn_stats{};
        if (ix->d_stats) HIPCHK(hipMemset(ix->d_stats, 0, sizeof h));
    }
    return PN_OK;
}

// ---------------------------------------------------------------------------
// planning helpers
// ---------------------------------------------------------------------------
static int pick_cap(size_t kp) {  // slots per (segment, query): multiple of 64, >= kp + 64
    const int caps[] = {128, 256, 512, 1024};
    for (int c : caps)
        if ((size_t)c >= kp + 64) return c;
    return 0;
}

struct ScanPlan {
    int nseg;
    size_t seg_len;
};

static ScanPlan plan_segments(size_t n, size_t q_tiles, int cap, int forced, size_t min_seg_rows, size_t max_entries,
                              size_t target_wgs = 2048, bool round_down = false) {
    size_t max_seg = max_entries / (size_t)cap;  // select kernel LDS budget
    if (max_seg < 1) max_seg = 1;
    size_t nseg;
    if (forced > 0) {
        nseg = (size_t)forced;
    } else {
        nseg = round_down ? target_wgs / q_tiles : (target_wgs + q_tiles - 1) / q_tiles;
        if (nseg < 1) nseg = 1;
        const size_t by_rows = (n + min_seg_rows - 1) / min_seg_rows;
        if (nseg > by_rows) nseg = by_rows;
    }
    if (nseg > max_seg) nseg = max_seg;
    if (nseg < 1) nseg = 1;
    size_t seg_len = round_up((n + nseg - 1) / nseg, (size_t)kRowPad);
    if (seg_len == 0) seg_len = kRowPad;
    nseg = (n + seg_len - 1) / seg_len;
    if (nseg < 1) nseg = 1;
    return ScanPlan{(int)nseg, seg_len};
}

// ---------------------------------------------------------------------------
// k-NN
// ---------------------------------------------------------------------------
template <typename T> struct Ops;
template <> struct Ops<float> {
    static hipError_t rerank(const CandBuf &cb, const float *P, size_t n, int dim, size_t ldp, const float *Q, int nq,
                             size_t ldq, int kout, uint64_t base, uint64_t *io, float *dd, size_t os, uint32_t *flags,
                             uint32_t *nf, const double *qn, const uint32_t *qbad, uint32_t *sel, unsigned long long *st,
                             hipStream_t s, int fe, int cm) {
        return launch_select_rerank_f32(cb, P, n, dim, ldp, Q, nq, ldq, kout, base, io, dd, os, flags, nf, qn, qbad, sel, st,
                                        s, fe, cm);
    }
    static hipError_t rerank_cos(const CandBuf &cb, const float *P, size_t n, int dim, size_t ldp, const float *Q, int nq,
                                 size_t ldq, int kout, uint64_t base, uint64_t *io, float *dd, size_t os, uint32_t *flags,
                                 uint32_t *nf, const double *qn, const uint32_t *qbad, uint32_t *sel, unsigned long long *st,
                                 hipStream_t s, int fe, int cm, const float *cn, const float *qnorm) {
        return launch_select_rerank_cos_f32(cb, P, n, dim, ldp, Q, nq, ldq, kout, base, io, dd, os, flags, nf, qn, qbad, sel,
                                            st, s, fe, cm, cn, qnorm);
    }
    static hipError_t pack(const float *s, size_t n, size_t c, size_t rs, float *d, size_t np, size_t ld, hipStream_t st) {
        return launch_pack_rows_f32(s, n, c, rs, d, np, ld, st);
    }
    static hipError_t knn(const float *P, size_t n, int dim, size_t ldp, const float *Q, int nq, size_t ldq, int kp,
                          size_t seg_len, const CandBuf &cb, const void *lk, const uint32_t *li, const uint32_t *nd,
                          uint32_t no, const float *pn, const float *qn, hipStream_t s, const uint32_t *qsel = nullptr) {
        return launch_exact_knn_f32(P, n, dim, ldp, Q, nq, ldq, kp, seg_len, cb, lk, li, nd, no, pn, qn, s, qsel);
    }
    static hipError_t cnorms(const float *X, size_t n, int dim, size_t ld, float *o, hipStream_t s) {
        return launch_cosine_norms_f32(X, n, dim, ld, o, s);
    }
    static hipError_t select(const CandBuf &cb, int nq, int kout, uint64_t base, uint64_t *io, float *dd, size_t os,
                             size_t oo, void *lk, uint32_t *li, const uint32_t *nd, uint32_t no, bool sk, hipStream_t s,
                             const uint32_t *osel = nullptr) {
        return launch_select_exact_f32(cb, nq, kout, base, io, dd, os, oo, lk, li, nd, no, sk, s, osel);
    }
    static hipError_t select_groups(const CandBuf &cb, int groups, int kp, int nq, int kout, uint64_t base, uint64_t *io,
                                    float *dd, size_t gs, const uint32_t *nd, uint32_t no, hipStream_t s, bool sk = false) {
        return launch_select_exact_groups_f32(cb, groups, kp, nq, kout, base, io, dd, gs, nd, no, s, sk);
    }
    static hipError_t merge(const uint64_t *pi, const float *pd, int np, size_t is, size_t ds, int nq, int kp, int ko,
                            uint64_t *io, float *dd, hipStream_t s, const uint32_t *nd, const uint32_t *osel, size_t os,
                            uint32_t *hc, bool sk = false) {
        return launch_merge_topk_f32(pi, pd, np, is, ds, nq, kp, ko, io, dd, s, nd, osel, os, hc, sk);
    }
};
template <> struct Ops<double> {
    static hipError_t rerank(const CandBuf &cb, const double *P, size_t n, int dim, size_t ldp, const double *Q, int nq,
                             size_t ldq, int kout, uint64_t base, uint64_t *io, double *dd, size_t os, uint32_t *flags,
                             uint32_t *nf, const double *qn, const uint32_t *qbad, uint32_t *sel, unsigned long long *st,
                             hipStream_t s, int fe, int cm) {
        return launch_select_rerank_f64(cb, P, n, dim, ldp, Q, nq, ldq, kout, base, io, dd, os, flags, nf, qn, qbad, sel, st,
                                        s, fe, cm);
    }
    static hipError_t rerank_cos(const CandBuf &cb, const double *P, size_t n, int dim, size_t ldp, const double *Q, int nq,
                                 size_t ldq, int kout, uint64_t base, uint64_t *io, double *dd, size_t os, uint32_t *flags,
                                 uint32_t *nf, const double *qn, const uint32_t *qbad, uint32_t *sel, unsigned long long *st,
                                 hipStream_t s, int fe, int cm, const double *cn, const double *qnorm) {
        return launch_select_rerank_cos_f64(cb, P, n, dim, ldp, Q, nq, ldq, kout, base, io, dd, os, flags, nf, qn, qbad, sel,
                                            st, s, fe, cm, cn, qnorm);
    }
    static hipError_t pack(const double *s, size_t n, size_t c, size_t rs, double *d, size_t np, size_t ld, hipStream_t st) {
        return launch_pack_rows_f64(s, n, c, rs, d, np, ld, st);
    }
    static hipError_t knn(const double *P, size_t n, int dim, size_t ldp, const double *Q, int nq, size_t ldq, int kp,
                          size_t seg_len, const CandBuf &cb, const void *lk, const uint32_t *li, const uint32_t *nd,
                          uint32_t no, const double *pn, const double *qn, hipStream_t s, const uint32_t *qsel = nullptr) {
        return launch_exact_knn_f64(P, n, dim, ldp, Q, nq, ldq, kp, seg_len, cb, lk, li, nd, no, pn, qn, s, qsel);
    }
    static hipError_t cnorms(const double *X, size_t n, int dim, size_t ld, double *o, hipStream_t s) {
        return launch_cosine_norms_f64(X, n, dim, ld, o, s);
    }
    static hipError_t select(const CandBuf &cb, int nq, int kout, uint64_t base, uint64_t *io, double *dd, size_t os,
                             size_t oo, void *lk, uint32_t *li, const uint32_t *nd, uint32_t no, bool sk, hipStream_t s,
                             const uint32_t *osel = nullptr) {
        return launch_select_exact_f64(cb, nq, kout, base, io, dd, os, oo, lk, li, nd, no, sk, s, osel);
    }
    static hipError_t select_groups(const CandBuf &cb, int groups, int kp, int nq, int kout, uint64_t base, uint64_t *io,
                                    double *dd, size_t gs, const uint32_t *nd, uint32_t no, hipStream_t s, bool sk = false) {
        return launch_select_exact_groups_f64(cb, groups, kp, nq, kout, base, io, dd, gs, nd, no, s, sk);
    }
    static hipError_t merge(const uint64_t *pi, const double *pd, int np, size_t is, size_t ds, int nq, int kp, int ko,
                            uint64_t *io, double *dd, hipStream_t s, const uint32_t *nd, const uint32_t *osel, size_t os,
                            uint32_t *hc, bool sk = false) {
        return launch_merge_topk_f64(pi, pd, np, is, ds, nq, kp, ko, io, dd, s, nd, osel, os, hc, sk);
    }
};

// exact engine on packed queries Qp [nq_pad][ld]; results of query q to d_idx/d_dist[q * out_stride + r], r < kout.
// second_tier: the queries are the ones a filter tier flagged -- their number lives on the device (nq_dev, of which
// this round covers [nq_off, nq_off + nq)), the grids are sized for nq and surplus query tiles exit at once; few
// queries are expected, so the rows are cut into as many segments as the select kernel takes.
template <typename T>
static int run_exact(const pn_index *ix, Workspace &ws, const T *Qp, size_t nq, size_t nq_pad, int dim_eff, size_t kout,
                     uint64_t *d_idx, T *d_dist, size_t out_stride, hipStream_t s, bool second_tier,
                     const uint32_t *nq_dev, uint32_t nq_off, CallRec *rec, const T *qnorm = nullptr,
                     const uint32_t *rowsel = nullptr) {
    // rowsel (second tier): query r of this round is row rowsel[nq_off + r] of Qp AND of the outputs -- the flagged
    // queries are scanned and answered in place (no gather / scatter launches)
    using KeyT = typename KeyOf<T>::type;
    // k beyond one candidate buffer (960 slots): rounds of <= 960 neighbours, each resuming strictly
    // after the last (distance key, row) of the previous one
    const size_t kRound = 960;
    const size_t k_first = kout < kRound ? kout : kRound;
    const int cap = pick_cap(k_first);
    const ScanPlan pl = second_tier ? plan_segments(ix->n, 1, cap, ix->opt_segments, 4096, 4096)
                                    : plan_segments(ix->n, nq_pad / kTileQ, cap, ix->opt_segments, 4096, 4096);
    DevBuf &bk = second_tier ? ws.w2_keys : ws.w_keys;
    DevBuf &bi = second_tier ? ws.w2_idx : ws.w_idx;
    DevBuf &bc = second_tier ? ws.w2_cnt : ws.w_cnt;
    DevBuf &bt = second_tier ? ws.w2_tau : ws.w_tau;
    const size_t slots = (size_t)pl.nseg * nq_pad * (size_t)cap;
    PNCHK(bk.ensure(slots * sizeof(KeyT)));
    PNCHK(bi.ensure(slots * sizeof(uint32_t)));
    PNCHK(bc.ensure((size_t)pl.nseg * nq_pad * sizeof(uint32_t)));
    PNCHK(bt.ensure((size_t)pl.nseg * nq_pad * sizeof(KeyT)));
    CandBuf cb{bk.p, (uint32_t *)bi.p, (uint32_t *)bc.p, bt.p, nq_pad, pl.nseg, cap};
    void *lo_key = nullptr;
    uint32_t *lo_idx = nullptr;
    if (kout > kRound) {
        PNCHK(ws.w_lo.ensure(nq_pad * (sizeof(KeyT) + sizeof(uint32_t))));
        lo_key = ws.w_lo.p;
        lo_idx = (uint32_t *)((char *)ws.w_lo.p + nq_pad * sizeof(KeyT));
    }
    const bool prof = rec && rec->prof;
    if (prof) {
        HIPCHK(hipEventRecord(rec->ev[0], s));
        rec->hot = true;
    }
    for (size_t done = 0; done < kout; done += kRound) {
        const size_t kr = kout - done < kRound ? kout - done : kRound;
        HIPCHK(Ops<T>::knn((const T *)ix->d_pts, ix->n, dim_eff, ix->ld, Qp, (int)nq, ix->ld, (int)kr, pl.seg_len, cb,
                           done ? lo_key : nullptr, done ? lo_idx : nullptr, nq_dev, nq_off,
                           qnorm ? (const T *)ix->d_cnorm : nullptr, qnorm, s, rowsel));
        if (prof && done == 0) HIPCHK(hipEventRecord(rec->ev[1], s));
        HIPCHK(Ops<T>::select(cb, (int)nq, (int)kr, ix->index_base, d_idx, d_dist, out_stride, done, lo_key, lo_idx,
                              nq_dev, nq_off, qnorm != nullptr, s, rowsel));
    }
    return PN_OK;
}

// Second tier behind a filter tier: the queries whose exclusions the re-rank could not prove -- LISTED by the re-rank
// kernel itself (d_sel[0 .. *d_nsel), the count stays on the device) -- are answered by the exact engine, all of it
// enqueued unconditionally and driven by the device-side count, so the host never waits to learn whether anything was
// flagged.  The flagged queries are read and answered IN PLACE through the list (round 2 listed them, gathered their
// rows, scanned, selected and scattered with a launch each: twelve empty launches, ~55 us, behind every batch; now
// five).  Rounds of kSecondTierRows queries bound the scratch.  h_count (nullable, mapped pinned memory): the count
// is left there for a LATER call to look at; returns *count_published = true when a kernel here does that.
constexpr size_t kSecondTierRows = 16384;
constexpr size_t kSecondTierFew = 256;  // the first flagged queries of a chunk take the many-segment path below
template <typename T>
static int second_tier_exact(const pn_index *ix, Workspace &ws, const T *Qp, size_t nq, size_t kout,
                             const uint32_t *d_sel, const uint32_t *d_nsel, uint64_t *d_idx, T *d_dist,
                             size_t out_stride, hipStream_t s, uint32_t *h_count, bool *count_published,
                             const T *qnorm = nullptr) {
    // qnorm (a Cosine index behind the bf16 filter): the queries' norms, indexed like the rows of Qp -- the scan computes
    // Cosine::distance, selections and the merge order the signed keys
    const bool cosq = qnorm != nullptr;
    const size_t F = nq < kSecondTierRows ? nq : kSecondTierRows, F_pad = round_up(F, (size_t)256);
    *count_published = false;
    size_t first = 0;
    // Few flagged queries are the normal case, and they share ONE 64-query tile: with the usual <= 32 segments that is
    // <= 32 workgroups for the whole corpus (10M rows: 0.4 s for a single flagged query; an f64 index's 1M rows: 16 ms,
    // six of the headline batch's steps).  So the first 256 flagged queries of a chunk are scanned with up to 512 row
    // segments -- every CU busy -- and selected in two levels: groups of 16 segments to a part each, then the (distance,
    // index) merge of the parts (the shard-merge kernel), which writes each answer straight to its query's row.
    // (f32 and, since round 3, f64 indexes: the kernels are templated on the element type.)
    using KeyT = typename KeyOf<T>::type;
    const size_t by_rows = (ix->n + 4095) / 4096;
    if (kout <= 192 && by_rows >= 16) {
        const size_t Ff = nq < kSecondTierFew ? nq : kSecondTierFew, Ff_pad = kSecondTierFew;
        const int cap = pick_cap(kout);
        size_t groups = by_rows / 16 < 32 ? by_rows / 16 : 32;
        while (groups > 1 && groups * kout * (8 + sizeof(KeyT)) > 60 * 1024) --groups;  // the merge kernel's LDS
        const size_t nseg = groups * 16;
        size_t seg_len = round_up((ix->n + nseg - 1) / nseg, (size_t)kRowPad);
        // (a last segment that starts beyond the corpus scans nothing and leaves an empty cell)
        const size_t cells = nseg * Ff_pad, slots = cells * (size_t)cap;
        PNCHK(ws.w2_keys.ensure(slots * sizeof(KeyT)));
        PNCHK(ws.w2_idx.ensure(slots * sizeof(uint32_t)));
        PNCHK(ws.w2_cnt.ensure(cells * sizeof(uint32_t)));
        PNCHK(ws.w2_tau.ensure(cells * sizeof(KeyT)));
        PNCHK(ws.w_fparts.ensure(groups * Ff_pad * kout * (sizeof(uint64_t) + sizeof(T))));
        uint64_t *p_idx = (uint64_t *)ws.w_fparts.p;
        T *p_dist = (T *)(p_idx + groups * Ff_pad * kout);
        CandBuf cb{ws.w2_keys.p, (uint32_t *)ws.w2_idx.p, (uint32_t *)ws.w2_cnt.p, ws.w2_tau.p, Ff_pad, (int)nseg, cap};
        HIPCHK(Ops<T>::knn((const T *)ix->d_pts, ix->n, (int)ix->dim, ix->ld, Qp, (int)Ff, ix->ld, (int)kout, seg_len, cb,
                           nullptr, nullptr, d_nsel, 0, cosq ? (const T *)ix->d_cnorm : nullptr, qnorm, s, d_sel));
        CandBuf cg = cb;
        cg.nseg = 16;
        // (index_base is added by the group selection; the merge orders the parts' GLOBAL indices)
        HIPCHK(Ops<T>::select_groups(cg, (int)groups, (int)kout, (int)Ff, (int)kout, ix->index_base, p_idx, p_dist,
                                     Ff_pad * kout, d_nsel, 0, s, cosq));
        HIPCHK(Ops<T>::merge(p_idx, p_dist, (int)groups, Ff_pad * kout, Ff_pad * kout, (int)Ff, (int)kout, (int)kout, d_idx,
                             d_dist, s, d_nsel, d_sel, out_stride, h_count, cosq));
        *count_published = h_count != nullptr;
        first = Ff;
    }
    for (size_t off = first; off < nq; off += F) {
        const size_t fr = nq - off < F ? nq - off : F;
        PNCHK(run_exact<T>(ix, ws, Qp, fr, F_pad, (int)ix->dim, kout, d_idx, d_dist, out_stride, s, true, d_nsel,
                           (uint32_t)off, nullptr, qnorm, d_sel));
    }
    return PN_OK;
}

// k' kept by the filter per (segment, query).  The proof in select.hip needs every segment's k'-th
// lower bound to clear the GLOBAL k-th exact distance; with >= 3 segments per query tile (always the
// case for the persistent partition when query tiles <= CUs / 3) a segment's k'-th bound is far above
// it even for k' = k + 2, and every extra slot costs appends.  Otherwise keep a wider margin.
static size_t mfma_slots(const pn_index *ix, size_t kout, size_t nq_pad) {
    if (ix->filter_slots > 0) return (size_t)ix->filter_slots < kout ? kout : (size_t)ix->filter_slots;
    const size_t q_tiles = nq_pad / 128;
    const bool many_segments = q_tiles * 3 <= (size_t)ix->n_cu;  // >= 3 segments per query tile
    if (many_segments && kout + 2 + kout / 16 <= 224) return kout + 2 + kout / 16;
    return kout + (kout < 16 ? 6 : kout / 4 + 4);
}

// Plan of the bf16 tier for one call.  R ~ 1.5k rows have a bf16 bound below the k-th neighbour's distance on
// benign data (the bound is loose by about a percent of the distance spread).  The proof needs every segment's
// k'-th bound to clear that distance, i.e. k' above the number of those R rows that fall into one segment:
//   level 0 (default): rows in arbitrary order -- a segment holds Poisson(R / segments) of them, k' is that mean
//            plus five standard deviations + 3; a query tile served by fewer than 4 workgroups is split into row
//            parts so that small buffers suffice;
//   level 1: all R rows may sit in one segment (corpus sorted by cluster): k' = R + 6 sqrt(R);
//   level 2: tier off.
// A call that had to hand more than 1/16 of its queries to the next tier raises the index's level (sticky).
// Experiment knobs of the planner (tools/ sweeps): read from the environment ONCE per process and validated -- the
// planner runs for every query chunk, incl. the one-point-per-call path, and a stray or malformed variable must not
// feed NaN or negative numbers into plan geometry.  0 = not set.
struct PlanKnobs {
    double scout_lambda = 0.0, scout_cap = 0.0;
    size_t sh_min_run = 0, wide_per_tile = 0, model_kmax = 0, min_per_tile = 0;
    double model_dz = 0.0, model_rank = 0.0, fill_tol = 0.0;
    bool unaligned = false, model_shared_only = false;
    bool debug = false;
};
static const PlanKnobs &plan_knobs() {
    static const PlanKnobs k = [] {
        PlanKnobs v;
        auto num = [](const char *name, double lo, double hi) -> double {
            const char *e = getenv(name);
            if (!e) return 0.0;
            char *end = nullptr;
            const double x = strtod(e, &end);
            return (end != e && x >= lo && x <= hi) ? x : 0.0;  // (NaN fails both comparisons)
        };
        v.scout_lambda = num("PN_EXP_SCOUT_LAMBDA", 0.01, 64.0);
        v.scout_cap = num("PN_EXP_SCOUT_CAP", 1.0, 64.0);
        v.sh_min_run = (size_t)num("PN_EXP_SH_MIN_RUN", 1.0, 1.0e6);
        v.wide_per_tile = (size_t)num("PN_EXP_WIDE_PER_TILE", 1.0, 32.0);
        v.model_kmax = (size_t)num("PN_EXP_MODEL_KMAX", 1.0, 1024.0);
        v.min_per_tile = (size_t)num("PN_EXP_MIN_PER_TILE", 1.0, 32.0);
        v.model_dz = num("PN_EXP_MODEL_DZ", -2.0, 2.0);
        v.model_rank = num("PN_EXP_MODEL_RANK", 1.0, 1024.0);
        v.fill_tol = num("PN_EXP_FILL_TOL", 0.001, 0.5);
        v.unaligned = getenv("PN_EXP_UNALIGNED") != nullptr;
        {
            const char *e = getenv("PN_EXP_MODEL_ALL_PLANS");
            v.model_shared_only = e && e[0] == '0';
        }
        v.debug = getenv("PN_DEBUG_PLAN") != nullptr;
        return v;
    }();
    return k;
}

// Seed model (round 4; VERDICT r3: "a per-index seed model fitted at build time from corpus rows used as pseudo-queries").
// Per-dimension moments of the translated corpus give, for any query, the mean and the variance of the filter's bound
// L'(q, .) over the corpus rows (Bf16SeedModel in bf16_filter.hip); a starting threshold is then mean - z sqrt(var).
// z is not derived but CALIBRATED: 256 corpus rows (at a constant stride) are run through the scout kernel as queries
// over the WHOLE corpus (128 segments; one query tile: a fortieth of a headline step), which leaves every query's ~3000
// smallest bounds; the host reads them, drops the query's own row and notes, per query, the z that would have put the
// threshold at the corpus ranks 16, 32, ..., 1024.  Their means over the queries are the grid a plan interpolates in
// (no distributional assumption: the tail's shape is measured), and what the model gets wrong per query is measured too:
// sm_sigma = standard deviation of ln(rank the model's threshold really has / rank it aimed at) at rank 64.  The model is
// accepted when sm_sigma <= kSeedModelSigma: uniform-like corpora pass, clustered ones do not and keep the scout launch.
// A threshold is never a correctness matter (the re-rank's proof decides; an unproven query goes to the next tier), only
// a matter of which tier answers, so a model that misbehaves later first aims higher, then switches itself off
// (rec_resolve: seed_model_widen, seed_model_off).
constexpr double kSeedModelSigma = 0.35;
static inline float host_s2f(uint32_t k) {
    const uint32_t b = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
    float f;
    std::memcpy(&f, &b, 4);
    return f;
}
template <typename T>  // T: element type of `rows` [n_pad][ld] -- the index's own rows, or the f64 normalised rows (Cosine)
static int seed_model_build(pn_index *ix, const T *rows, hipStream_t s) {
    ix->sm_ok = false;
    const int dim = (int)ix->dim;
    if (!ix->bf16_ok || !ix->d_img || ix->n < 100000 || ix->seed_model == 0) return PN_OK;
    const bool wide = bf16_is_wide(dim);
    constexpr size_t NQ = 256;
    constexpr int NWG = 128;                 // one query tile of 256: NWG workgroups
    const int NSEG = wide ? 2 * NWG : NWG;   // (wide rows: a workgroup's two row halves are two segments)
    const size_t r_tiles = wide ? (ix->n + 255) / 256 : (ix->n + 63) / 64;
    const size_t per = 2 * (size_t)bf16_scout_list();
    const size_t ld = ix->ld, cells = NQ * (size_t)NSEG, words = cells * per;
    const auto t_begin = std::chrono::steady_clock::now();
    auto ms_since = [&](std::chrono::steady_clock::time_point t0) {
        return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    };
    // one allocation for all device temporaries (every hipFree waits for the device: nine of them cost the build ~10 ms)
    struct Part { void *p = nullptr; } mom, qrows, bq, qn, qbad, lists, keys, cnt, tau;
    DevTmp blk;
    {
        const size_t sizes[9] = {4 * (size_t)dim * sizeof(double), NQ * ld * sizeof(T), bf16_query_bytes(NQ, dim, ix->bf16_ci),
                                 NQ * sizeof(double), NQ * sizeof(uint32_t), words * sizeof(float), cells * 64 * 8,
                                 cells * sizeof(uint32_t), cells * sizeof(uint32_t)};
        Part *parts[9] = {&mom, &qrows, &bq, &qn, &qbad, &lists, &keys, &cnt, &tau};
        size_t total = 0;
        for (size_t sz : sizes) total += round_up(sz, (size_t)256);
        HIPCHK(blk.alloc(total));
        size_t off = 0;
        for (int i = 0; i < 9; ++i) {
            parts[i]->p = (char *)blk.p + off;
            off += round_up(sizes[i], (size_t)256);
        }
    }
    HIPCHK(hipMemsetAsync(mom.p, 0, 4 * (size_t)dim * sizeof(double), s));
    HIPCHK(launch_bf16_column_moments<T>(rows, ix->d_mu, ix->n, dim, ld, (double *)mom.p, s));
    const size_t stride = ix->n / NQ;
    HIPCHK(hipMemcpy2DAsync(qrows.p, ld * sizeof(T), (const char *)rows + (stride / 2) * ld * sizeof(T),
                            stride * ld * sizeof(T), ld * sizeof(T), NQ, hipMemcpyDeviceToDevice, s));
    HIPCHK(launch_bf16_pack_queries<T>((const T *)qrows.p, ix->d_mu, NQ, NQ, dim, ld, bq.p, (double *)qn.p,
                                       (uint32_t *)qbad.p, ix->bf16_ci, ix->bf16_bmax, ix->bf16_dmax, s));
    CandBuf cb{};
    cb.keys = keys.p;
    cb.idx = (uint32_t *)keys.p + 1;
    cb.idx_stride = 2;
    cb.cnt = (uint32_t *)cnt.p;
    cb.tau = tau.p;
    cb.nq_pad = NQ;
    cb.nseg = NSEG;
    cb.cap = 64;
    HIPCHK(hipMemsetD32Async((hipDeviceptr_t)lists.p, (int)0x7F800000u, words, s));
    // scout-only launch over every tile of every run (scout_max beyond any run's length): buffers are not touched
    if (wide)
        HIPCHK(launch_bf16_wide_filter(ix->d_img, ix->n, dim, bq.p, 24, cb, NWG, (int)(r_tiles / NWG + 2), nullptr, false,
                                       (float *)lists.p, s));
    else
        HIPCHK(launch_bf16_filter(ix->d_img, ix->n, dim, bq.p, 24, cb, NWG, 1, (int)(r_tiles / NWG + 2), nullptr, false,
                                  (float *)lists.p, ix->bf16_ci, s, nullptr));
    std::vector<double> h_mom(4 * (size_t)dim);
    std::vector<T> h_q(NQ * ld);
    std::vector<float> h_mu((size_t)dim), h_lists(words);
    HIPCHK(hipMemcpyAsync(h_mom.data(), mom.p, h_mom.size() * sizeof(double), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(h_q.data(), qrows.p, h_q.size() * sizeof(T), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(h_mu.data(), ix->d_mu, h_mu.size() * sizeof(float), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(h_lists.data(), lists.p, words * sizeof(float), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    const double t_dev = ms_since(t_begin);
    const double inv_n = 1.0 / (double)ix->n;
    std::vector<float> h_model(3 * ld, 0.0f);
    double c0 = 0.0, v0 = 0.0;
    for (int k = 0; k < dim; ++k) {
        const double m1 = h_mom[k] * inv_n, m2 = h_mom[dim + k] * inv_n, m3 = h_mom[2 * (size_t)dim + k] * inv_n,
                     m4 = h_mom[3 * (size_t)dim + k] * inv_n;
        h_model[k] = (float)m1;
        h_model[ld + k] = (float)(4.0 * (m2 - m1 * m1));
        h_model[2 * ld + k] = (float)(4.0 * (m3 - m2 * m1));
        c0 += m2;
        v0 += m4 - m2 * m2;
    }
    if (!(c0 > 0.0) || !(v0 > 0.0) || !std::isfinite(c0) || !std::isfinite(v0)) return PN_OK;
    // per calibration query: its model mean / deviation (the pack kernel's arithmetic: f64 over the f32 model words), the
    // bounds at the grid's ranks (selection, no sort: nth_element from the largest rank down, each on the prefix the one
    // before left; the query's own row -- the smallest of all -- shifts every rank by one) and, once z at rank 64 is
    // known, how many of its bounds a threshold aimed there really has below it (a count)
    const size_t top_rank = (size_t)16 << (kSeedModelGrid - 1);
    std::vector<double> qmean(NQ), qsd(NQ, 0.0);
    std::vector<float> gridv(NQ * (size_t)kSeedModelGrid), all((size_t)NQ * NSEG * per);
    size_t used = 0;
    if ((size_t)NSEG * per <= top_rank + 1) return PN_OK;
    auto one_query = [&](size_t q) {
        double mean = 0.0, var = 0.0;
        for (int k = 0; k < dim; ++k) {
            const double c = (double)h_q[q * ld + k] - (double)h_mu[k];
            mean += c * (double)h_model[k];
            var += c * (c * (double)h_model[ld + k] - (double)h_model[2 * ld + k]);
        }
        var += v0;
        if (!(var > 0.0) || !std::isfinite(var) || !std::isfinite(mean)) return;
        float *a = &all[q * (size_t)NSEG * per];
        for (size_t seg = 0; seg < (size_t)NSEG; ++seg)
            std::memcpy(a + seg * per, &h_lists[(seg * NQ + q) * per], per * sizeof(float));
        size_t hi = (size_t)NSEG * per;
        bool ok = true;
        for (int j = kSeedModelGrid - 1; j >= 0; --j) {
            const size_t rho = (size_t)16 << j;  // rank rho without the query's own row = element rho (0-based) with it
            std::nth_element(a, a + (ptrdiff_t)rho, a + (ptrdiff_t)hi);
            gridv[q * kSeedModelGrid + (size_t)j] = a[rho];
            ok = ok && std::isfinite(a[rho]);
            hi = rho;
        }
        if (!ok) return;  // (fewer rows than that: n >= 100000 rules it out)
        qmean[q] = c0 - 2.0 * mean;
        qsd[q] = std::sqrt(var);  // (> 0 marks the query as used)
    };
    {   // the selections are 5-7 ms of host time on one thread: eight threads take a query each in turn
        constexpr size_t NT = 8;
        std::vector<std::thread> pool;
        bool threaded = true;
        try {
            for (size_t t = 0; t < NT; ++t)
                pool.emplace_back([&, t] {
                    for (size_t q = t; q < NQ; q += NT) one_query(q);
                });
        } catch (...) {  // no threads to be had: whatever was not started runs here
            threaded = false;
        }
        const size_t started = pool.size();
        for (std::thread &th : pool) th.join();
        if (!threaded)
            for (size_t t = started; t < NT; ++t)
                for (size_t q = t; q < NQ; q += NT) one_query(q);
    }
    for (size_t q = 0; q < NQ; ++q) used += qsd[q] > 0.0 ? 1u : 0u;
    if (used < NQ * 3 / 4) return PN_OK;
    for (int j = 0; j < kSeedModelGrid; ++j) {
        double zs = 0.0;
        for (size_t q = 0; q < NQ; ++q)
            if (qsd[q] > 0.0) zs += (qmean[q] - (double)gridv[q * kSeedModelGrid + (size_t)j]) / qsd[q];
        ix->sm_zgrid[j] = zs / (double)used;
    }
    // what the model gets wrong: thresholds aimed at rank 64 -- the rank each one really has among its query's bounds
    const double z64 = ix->sm_zgrid[2];
    double ls = 0.0, lss = 0.0;
    for (size_t q = 0; q < NQ; ++q) {
        if (!(qsd[q] > 0.0)) continue;
        const float S = (float)(qmean[q] - z64 * qsd[q]);
        const float *a = &all[q * (size_t)NSEG * per];
        size_t cnt_below = 0;  // (everything below a grid value sits in front of it after the selections: the prefix up to
        for (size_t e = 0; e <= top_rank; ++e) cnt_below += a[e] < S ? 1u : 0u;  // the top rank holds every candidate)
        if (cnt_below > 0) cnt_below -= 1;  // the query's own row
        const double l = std::log(((double)cnt_below + 0.5) / 64.0);
        ls += l;
        lss += l * l;
    }
    const double lm = ls / (double)used, lv = lss / (double)used - lm * lm;
    ix->sm_c0 = c0;
    ix->sm_v0 = v0;
    ix->sm_sigma = std::sqrt(lv > 0.0 ? lv : 0.0);
    bool mono = true;  // (z must fall as the rank grows)
    for (int j = 1; j < kSeedModelGrid; ++j) mono = mono && ix->sm_zgrid[j] < ix->sm_zgrid[j - 1];
    if (plan_knobs().debug)
        std::fprintf(stderr, "[pn seed model] build: device part %.2f ms, host part %.2f ms\n", t_dev, ms_since(t_begin) - t_dev);
    if (plan_knobs().debug)
        std::fprintf(stderr, "[pn seed model] n %zu dim %d: z at ranks 16..1024 = %.3f %.3f %.3f %.3f %.3f %.3f %.3f, log-rank "
                             "error at rank 64: mean %.3f sigma %.3f; c0 %.6g v0 %.6g\n",
                     ix->n, dim, ix->sm_zgrid[0], ix->sm_zgrid[1], ix->sm_zgrid[2], ix->sm_zgrid[3], ix->sm_zgrid[4],
                     ix->sm_zgrid[5], ix->sm_zgrid[6], lm, ix->sm_sigma, c0, v0);
    if (!mono || !(ix->sm_zgrid[kSeedModelGrid - 1] > 0.5) || !(ix->sm_sigma <= kSeedModelSigma)) return PN_OK;
    HIPCHK(hipMalloc((void **)&ix->d_smodel, h_model.size() * sizeof(float)));
    HIPCHK(hipMemcpyAsync(ix->d_smodel, h_model.data(), h_model.size() * sizeof(float), hipMemcpyHostToDevice, s));
    HIPCHK(hipStreamSynchronize(s));
    ix->sm_ok = true;
    return PN_OK;
}

struct Bf16Plan {
    int n_wg, split, nseg, kp, cap, scout_max;
    bool ok, aligned;
    // shared scout (several segments per query): scout-only launch, merged seed of rank seed_rank, main launch
    bool shared_scout;
    int scout_tiles, seed_rank;
    // wide rows (D > 128): per_tile workgroups per query tile, each a row range; 2 * per_tile segments per query
    bool wide;
    int per_tile;
    // shared thresholds (bf16_filter.hip): refresher workgroups behind the n_wg main ones, rank of the shared threshold
    int n_refresh, sh_rank;
    int first_eval;  // candidates the re-rank evaluates in its first round (the plan's R, rounded up)
    bool model_seed;  // thresholds from the index's seed model: no scout launch (narrow rows, small k)
    double model_z;   // its z for this k
};
// Wide rows: one workgroup per CU at a time (128 KiB of LDS), a whole number of workgroups per query tile: every cell
// then has exactly one writer (no memsets, exact segment count) and the workgroups of different query tiles walk the
// same rows at the same time, which is what lets them share corpus tiles in L2 (1M x 768, 10^4 queries: 240 aligned
// workgroups 15.0 ms, 256 balanced ones 15.6 ms).  The kernel itself takes any number of workgroups.
// Seed model in a plan (see bf16_plan): sets model_seed / model_z when the index's model serves this call
static void plan_seed_model(const pn_index *ix, Bf16Plan &p, double R, size_t kout, bool eligible) {
    p.model_seed = false;
    p.model_z = 0.0;
    if (!eligible || !ix->sm_ok || ix->seed_model == 0 || ix->sh.seed_model_off.load(std::memory_order_relaxed) ||
        kout > (plan_knobs().model_kmax ? plan_knobs().model_kmax : (size_t)128) || ix->n < 100000)
        return;
    const double sig = ix->sm_sigma > 0.05 ? ix->sm_sigma : 0.05;
    double rho_t = (R + 5.5 * std::sqrt(R)) * std::exp(4.0 * sig) * std::pow(1.5, (double)ix->sh.seed_model_widen.load(std::memory_order_relaxed));
    if (plan_knobs().model_rank > 0.0) rho_t = plan_knobs().model_rank;  // experiments only
    if (rho_t < 16.0) rho_t = 16.0;
    const double top_rank = (double)(16u << (kSeedModelGrid - 1));
    if (rho_t > top_rank || rho_t * 50.0 > (double)ix->n) return;
    const double x = std::log2(rho_t / 16.0);  // position in the grid (ranks 16 << j)
    int j = (int)x;
    if (j > kSeedModelGrid - 2) j = kSeedModelGrid - 2;
    const double f = x - (double)j;
    p.model_z = ix->sm_zgrid[j] * (1.0 - f) + ix->sm_zgrid[j + 1] * f + plan_knobs().model_dz;
    p.model_seed = p.model_z > 0.5 && p.model_z < 12.0;
}
static Bf16Plan bf16_plan_wide(const pn_index *ix, size_t nq_pad, size_t kout, int level) {
    Bf16Plan p{};
    p.wide = true;
    p.split = 1;
    const size_t q_tiles = nq_pad / 256, r_tiles = (ix->n + 255) / 256;
    // c workgroups per query tile, each a row range: the fewest within 8 % of the best CU occupancy among those that
    // leave every run at least 8 row tiles (every run pays a scout pass and a pipeline fill, every segment adds k'
    // candidates to the query's re-rank).  More query tiles than CUs: the grid runs in rounds; a persistent balanced
    // partition (256 equal slices) was measured slower there too (configs[3] on one GPU: 1.78 s vs 1.70 s).
    const size_t n_cu = (size_t)ix->n_cu;
    auto eff_of = [&](size_t c) {
        const double rounds = (double)(q_tiles * c) / (double)n_cu;
        return rounds <= 1.0 ? rounds : rounds / std::ceil(rounds);
    };
    size_t c_max = 1;
    double best_eff = eff_of(1);
    for (size_t c = 2; c <= 32; ++c) {
        if (r_tiles / c < 8) break;
        if (ix->opt_segments > 0 && 2 * c > (size_t)ix->opt_segments) break;
        c_max = c;
        if (eff_of(c) > best_eff) best_eff = eff_of(c);
    }
    size_t best = 1;
    while (best < c_max && eff_of(best) < best_eff - 0.08) ++best;
    if (plan_knobs().wide_per_tile && plan_knobs().wide_per_tile <= c_max) best = plan_knobs().wide_per_tile;  // experiments only
    const double R = (ix->centered ? 1.5 : 2.0) * (double)kout + 4.0;
    double segs = 2.0;
    for (;; best = (best + 1) / 2) {
        size_t n_wg = q_tiles * best;
        if (n_wg > q_tiles * r_tiles) n_wg = q_tiles * r_tiles;  // tiny corpora: one unit per workgroup at least
        p.n_wg = (int)n_wg;
        p.per_tile = (int)(n_wg / q_tiles > 0 ? n_wg / q_tiles : 1);  // workgroups per query tile
        p.aligned = n_wg % q_tiles == 0;
        p.nseg = bf16_wide_segments(q_tiles, p.n_wg);
        segs = 2.0 * (double)p.per_tile;  // segments a query's relevant rows are spread over
        double kp;
        if (ix->filter_slots > 0)
            kp = (double)((size_t)ix->filter_slots < kout ? kout : (size_t)ix->filter_slots);
        else if (level != 0)
            kp = R + 6.0 * std::sqrt(R) + 4.0;
        else {
            const double per = R / segs;
            kp = per + 5.0 * std::sqrt(per) + 3.0;
            if (kp < 8.0) kp = 8.0;
        }
        p.kp = (int)std::ceil(kp);
        p.ok = level < 2 && p.kp + 32 <= 256;
        p.cap = p.ok ? bf16_cap_for(p.kp) : 0;
        // the re-rank kernel gathers all cells of a query into 64 KiB of LDS (12 B per slot + the query row)
        if (best == 1 || (size_t)p.nseg * (size_t)p.cap * 12 + (ix->dim + 8) * 4 <= 60 * 1024) break;
    }
    if ((size_t)p.nseg * (size_t)p.cap * 12 + (ix->dim + 8) * 4 > 64 * 1024) p.ok = false;
    // a wave scouts 128 rows per tile; the scouted rows should hold < 0.1 of the R relevant rows in expectation
    const double sm = (double)ix->n / (10.0 * R) / 128.0;
    p.scout_max = sm > 32.0 ? 32 : (int)sm;
    // Shared scout (see bf16_plan): the rows scouted by all waves of a query form one sample; lambda = expected
    // number of the R relevant rows in it (<= 1.2), seed_rank = smallest rank with P(Poisson >= rank) <= 1e-7
    p.shared_scout = false;
#ifndef PN_DIAG_NO_SHARED_SCOUT
    if (p.ok && level == 0 && ix->filter_slots == 0 && segs >= 4.0) {
        const size_t run_len = q_tiles * r_tiles / (size_t)p.n_wg;
        double lam_w = 1.2, t_cap_w = 16.0;
        if (plan_knobs().scout_lambda > 0.0) lam_w = plan_knobs().scout_lambda;  // experiments only
        if (plan_knobs().scout_cap > 0.0) t_cap_w = plan_knobs().scout_cap;      // experiments only
        double t = lam_w * (double)ix->n / (R * segs * 128.0);  // tiles per run for lambda = 1.2
        if (t > t_cap_w) t = t_cap_w;
        if (t > (double)run_len / 8.0) t = (double)run_len / 8.0;
        p.scout_tiles = (int)t;
        if (p.scout_tiles >= 1) {
            // a query tile may be touched by one workgroup more than per_tile: the sample is at most this large
            const double lam = R * (double)p.scout_tiles * 128.0 * (double)p.nseg / (double)ix->n;
            double term = std::exp(-lam), cdf = term;
            int rank = 1;
            while (1.0 - cdf > 1e-7 && rank < 2 * bf16_scout_list()) {
                term *= lam / (double)rank;
                cdf += term;
                ++rank;
            }
            p.seed_rank = rank < 5 ? 5 : rank;
            p.shared_scout = p.seed_rank <= bf16_scout_list();
        }
    }
#endif
    p.first_eval = (int)std::ceil(R);
    plan_seed_model(ix, p, R, kout, p.shared_scout);
    if (plan_knobs().debug)
        fprintf(stderr, "bf16_plan_wide: n %zu q_tiles %zu k %zu R %.0f: n_wg %d per_tile %d nseg %d kp %d cap %d shared_scout %d "
                        "scout_tiles %d seed_rank %d model_seed %d z %.3f\n",
                ix->n, q_tiles, kout, R, p.n_wg, p.per_tile, p.nseg, p.kp, p.cap, (int)p.shared_scout, p.scout_tiles, p.seed_rank,
                (int)p.model_seed, p.model_z);
    return p;
}
// Narrow rows, enough work for every workgroup slot: a grid of c workgroups per query tile, run in rounds when it
// exceeds the 2 n_cu slots -- the fewest c within 5 % of the best slot occupancy.  Whole runs only, and the resident
// workgroups (different query tiles, same row ranges) walk the same rows at the same time and share them in L2;
// 2 n_cu persistent equal slices straddle query tiles and walk unrelated rows (configs[2], 10^7 x 128, 10^5
// queries, k = 100: 291 -> 255 ms; one shard of configs[4], 1.25 10^7 x 96, 10^6 queries: 2.40 -> 2.28 s).
static size_t bf16_grid_wgs(const pn_index *ix, size_t q_tiles, size_t r_tiles, size_t n_wg, bool two_rounds_ok = false) {
    if (ix->opt_segments != 0 || n_wg != 2 * (size_t)ix->n_cu) return n_wg;
    const size_t slots = n_wg;
    auto eff_of = [&](size_t c) {
        const double rounds = (double)(q_tiles * c) / (double)slots;
        return rounds <= 1.0 ? rounds : rounds / std::ceil(rounds);
    };
    size_t c_max = 1;
    double best_eff = eff_of(1);
    for (size_t c = 2; c <= 32; ++c) {
        if (r_tiles / c < 32) break;
        c_max = c;
        if (eff_of(c) > best_eff) best_eff = eff_of(c);
    }
    // (small k on a seed-model plan: runs are cheap to start, so the grid takes the fewest ranges within ONE percent of the
    // best fill, not five, among those that leave runs of 64 tiles -- 10M x 128, 10^5 queries, k = 10: 5 -> 9 ranges per
    // tile, 183.4 -> 174.2 ms; 1M x 96, 2 10^5 queries: 5 -> 13, 35.4 -> 32.6 ms; C2: 12 -> 25, below)
    // (not on corpora beyond 32M rows: configs[4] whole, 10^8 x 96 -- the 835-tile chunk with 11 ranges instead of 3 made the
    // step 1.5 % SLOWER on the same device, as 8 ranges instead of 4 did for the other chunks)
    if (r_tiles > 500000) two_rounds_ok = false;
    double tol = two_rounds_ok ? 0.01 : 0.05;
    if (plan_knobs().fill_tol > 0.0) tol = plan_knobs().fill_tol;  // experiments only
    if (two_rounds_ok) {
        best_eff = eff_of(1);
        size_t c_fine = 1;
        for (size_t c = 2; c <= c_max; ++c) {
            if (r_tiles / c < 64) break;
            c_fine = c;
            if (eff_of(c) > best_eff) best_eff = eff_of(c);
        }
        c_max = c_fine;
    }
    size_t c = 1;
    while (c < c_max && eff_of(c) < best_eff - tol) ++c;
    // (end of round 4: never ONE workgroup per query tile when two fill the slots as well -- a run over the whole corpus
    // is the one shape where every relevant row of a query lands in one buffer (k' = R + 5 sqrt(R) + 3 and still a few
    // unproven queries per 10^6, each an exact scan of the corpus) and where co-walking workgroups have the longest way
    // to drift apart; a 12.5M x 96 shard of configs[4], 10^6 queries: 1 / 2 / 3 / 4 ranges 1931 / 1861 / 1855 / 1864 ms,
    // profiles/r04_min_per_tile.log)
    if (c == 1 && c_max >= 2 && eff_of(2) >= best_eff - 0.05) c = 2;
    // (and runs beyond half a million tiles are cut once more: configs[4] whole, 10^8 x 96 -- 2 / 4 / 8 ranges 15.6 / 14.85 /
    // 16.25 s per 10^6 queries; co-walking workgroups have 781 k tiles to drift apart in with two ranges, and the
    // counters show it: 4.2 TB through the L2's memory side per chunk against 0.33 TB if they stayed together)
    if (c < 4 && c_max >= 4 && r_tiles / c > 500000 && eff_of(4) >= best_eff - 0.05) c = 4;
    // (and a grid that fills ONE round of slots poorly takes two rounds of shorter runs when those fill the slots at
    // least 2.5 % better: C2, 40 query tiles on 512 slots -- 12 ranges per tile = 480 workgroups (93.75 %) against 25 =
    // 1000 (97.7 % of two rounds): main launch 2.00-2.02 -> 1.95-1.96 ms; a 125 k- / 250 k-row shard of it -2 / -2.8 %,
    // 1M x 96 -3.6 %, k = 1 -2.8 % (profiles/r04_two_rounds.log).  Small k only: at k = 100 the kernel gains 5 % and the
    // re-rank of 25 cells per query gives it back.  With thresholds from the seed model a run costs little to start.)
    if (two_rounds_ok && q_tiles * c <= slots) {
        const size_t c2 = 2 * slots / q_tiles;
        if (c2 > c && c2 <= c_max && r_tiles / c2 >= 64 && eff_of(c2) >= eff_of(c) + 0.025) c = c2;
    }
    if (plan_knobs().min_per_tile > c && plan_knobs().min_per_tile <= c_max) c = plan_knobs().min_per_tile;  // experiments only
    return q_tiles * c;
}
static Bf16Plan bf16_plan(const pn_index *ix, size_t nq_pad, size_t kout, int level) {
    if (bf16_is_wide((int)ix->dim)) return bf16_plan_wide(ix, nq_pad, kout, level);
    Bf16Plan p{};
    const size_t q_tiles = nq_pad / 256, r_tiles = (ix->n + 63) / 64;
    size_t n_wg = 2 * (size_t)ix->n_cu;
    size_t cap_wg = q_tiles * 32;  // at most ~32 workgroups per query tile, at least ~32 row tiles each
    size_t by_work = (q_tiles * r_tiles + 31) / 32;
    // A handful of queries (one or two query tiles: the reference's one-point-per-call pattern against a large corpus) is
    // HBM-bound -- the image is streamed once whatever the batch -- and 32 workgroups stream it at a sixteenth of the
    // chip's rate (round 2: 621 us for one query against 1M x 128).  So few query tiles take a workgroup per CU, down to
    // runs of 16 row tiles (measured, one query per call, host API: 32 segments 621 us, 128: 185, 256: 145, 512: 202 --
    // beyond one per CU the re-rank's walk over the segments costs more than the shorter runs save).
    if (q_tiles <= 2) {
        cap_wg = (size_t)ix->n_cu / q_tiles * q_tiles;
        by_work = (q_tiles * r_tiles + 15) / 16;
    }
    if (by_work < cap_wg) cap_wg = by_work;
    if (ix->opt_segments > 0 && q_tiles * (size_t)ix->opt_segments < cap_wg) cap_wg = q_tiles * (size_t)ix->opt_segments;
    if (cap_wg < 1) cap_wg = 1;
    if (n_wg > cap_wg) n_wg = cap_wg;
    if (q_tiles > 2 && !plan_knobs().unaligned)  // (many query tiles: a grid run in rounds)
        n_wg = bf16_grid_wgs(ix, q_tiles, r_tiles, n_wg, kout <= 32 && level == 0 && ix->sm_ok && ix->seed_model != 0);
    // a whole number of workgroups per query tile: each workgroup's slice is then ONE run.  A slice that straddles
    // a query-tile boundary is two runs, each with its own operand load, scout pass and buffer warm-up, and those
    // workgroups set the kernel's time (C2: 512 workgroups = 12.8 per tile 4.34 ms, 480 = 12 per tile 3.62 ms)
    if (n_wg > q_tiles && !plan_knobs().unaligned) n_wg = n_wg / q_tiles * q_tiles;
    p.n_wg = (int)n_wg;
    // rows whose bound lies below the k-th neighbour's distance: ~1.2 k on benign data once the vectors are
    // translated by the corpus mean (measured: 11.8 for k = 10, 116 for k = 100); planned with a margin
    // (norm-in-accumulator layout: the per-query error constant uses the corpus maxima -- measured 23 / 237 rows for
    // k = 10 / 100 against 20 / 185 with per-row constants.  For small k planning for that costs more in candidates
    // (C2: +5 % step time) than the few extra unproven queries cost in the second tier.  For k >= 32 the plan uses the
    // measured 2.4 k instead of 2 k -- with a 4 sigma margin on corpora below 4M rows, which gives the same k' as before
    // where a query's rows are spread over many segments (1M x 128, k = 100, 12 segments: 41; planning 5 sigma there
    // costs 8 % in candidates), and with 5 sigma on larger ones, where ANY unproven query costs a scan of the whole
    // corpus for a 64-query tile of the exact engine: 10M x 128, 10^5 queries, 5 segments: k' 87 instead of 76; 86
    // queries per batch had overflowed a buffer and cost 22 of the batch's 265 ms, 10 per batch (k' = 80) still did.)
    const bool ci_many = ix->bf16_ci && !ix->centered && kout >= 32;
    const double R = (ci_many ? 2.4 : ix->centered ? 1.5 : 2.0) * (double)kout + 4.0;
    // (small k on such corpora: half a sigma more -- one unproven query in 10^6 is a 10^7-row scan there)
    const double n_sigma = ix->n < ((size_t)4 << 20) ? (ci_many ? 4.0 : 5.0) : (ci_many ? 5.0 : 5.5);
    size_t per_tile = n_wg / q_tiles;  // workgroups (= segments) per query tile
    if (per_tile < 1) per_tile = 1;
    // Large k (round 4).  A query's R relevant rows spread over its per_tile segments, a buffer holds at most 224
    // candidates (k' + 32 <= 256 slots), and k' = R / per_tile + margin: at 1M x 128 the 12 segments of the headline plan
    // serve k up to ~370.  Beyond that MORE, SHORTER segments per query tile -- a grid of c workgroups per query tile run
    // in rounds, as for many query tiles -- keep k' inside a buffer: k = 500 stays at 12 segments (k' = 143), k = 1000
    // takes 24 (two rounds of 480 workgroups).  Round 3 "split the rows into parts" here, which does not add
    // workgroups: with n_wg >= q_tiles * split a query still has per_tile segments, k' was planned for twice as many,
    // and 97 % of a k = 500 batch failed its proof -- after which the index switched the tier off (126 ms per batch
    // on the exact engine; 853 ms at k = 1000).
    if (level == 0 && ix->opt_segments == 0 && ix->filter_slots == 0 && n_wg >= q_tiles) {
        auto kp_of = [&](size_t c) {
            const double per = R / (double)c;
            return per + n_sigma * std::sqrt(per) + 3.0;
        };
        size_t c = per_tile;
        while (kp_of(c) > 200.0 && c + per_tile <= 96 && r_tiles / (c + per_tile) >= 32) c += per_tile;
        if (c != per_tile && kp_of(c) <= 224.0) {
            per_tile = c;
            n_wg = q_tiles * c;
            p.n_wg = (int)n_wg;
        }
    }
    auto kp_for = [&](int split) -> double {
        if (ix->filter_slots > 0) return (double)((size_t)ix->filter_slots < kout ? kout : (size_t)ix->filter_slots);
        if (level != 0) return R + 6.0 * std::sqrt(R) + 4.0;
        // segments a query's rows are spread over: its workgroups, or its row parts when a workgroup walks several
        // (parts do not multiply the workgroups of a query tile: see above)
        const size_t segs = per_tile > (size_t)split ? per_tile : (size_t)split;
        const double per = R / (double)segs;
        const double v = per + n_sigma * std::sqrt(per) + 3.0;  // ~1e-6 per (query, segment) of holding more than that
        return v < 8.0 ? 8.0 : v;
    };
    // split the rows of a query tile into parts only while one buffer would need more than 128 slots
    // (every part pays its own warm-up), and keep parts of at least 256 tiles
    p.split = 1;
    if (level == 0 && ix->opt_segments == 0 && ix->filter_slots == 0)
        while (p.split < 8 && kp_for(p.split) > 96.0 && r_tiles / (size_t)(2 * p.split) >= 256 &&
               (size_t)(2 * p.split) > per_tile)  // (only where more parts ARE more segments: few workgroups per query tile)
            p.split *= 2;
    double kp = kp_for(p.split);
    // A k' between 17 and 31 lives in 128-slot buffers anyway, so the tail it is planned against costs nothing but
    // candidates there: the count of relevant rows varies from query to query (22 +- 5.3, up to 44 of 256 queries for
    // k = 10: heavier than Poisson), and the queries at the upper end are the ones that overflow a segment.  Measured,
    // 10M x 128, 10^5 queries, k = 10, 5 segments: k' = 20 left 15 queries per batch to the second tier -- 15 exact
    // scans of 10^7 rows, 22 of the step's 215 ms; planned for twice the mean (k' = 30): see DESIGN.md 4.0.
    if (level == 0 && ix->filter_slots == 0 && kp > 16.0 && kp < 32.0) {
        const double per2 = 2.0 * R / (double)(per_tile > (size_t)p.split ? per_tile : (size_t)p.split);
        const double v = per2 + n_sigma * std::sqrt(per2) + 3.0;
        kp = v < 32.0 ? (v > kp ? v : kp) : 32.0;
    }
    p.kp = (int)std::ceil(kp);
    p.ok = level < 2 && p.kp + 32 <= 256;
    p.cap = p.ok ? bf16_cap_for(p.kp) : 0;
    p.nseg = bf16_segments(q_tiles, p.n_wg, p.split);
    // aligned partition: exactly n_wg / q_tiles segments, every cell written by exactly one workgroup
    p.aligned = p.split == 1 && (size_t)p.n_wg >= q_tiles && (size_t)p.n_wg % q_tiles == 0;
    if (p.aligned) p.nseg = p.n_wg / (int)q_tiles;
    // scouted rows should hold < 0.1 of the R relevant rows in expectation: tiles <= N / (10 R) / 64
    const double sm = (double)r_tiles / (10.0 * R);
    p.scout_max = sm > 64.0 ? 64 : (int)sm;
    // Shared scout: with S >= 2 segments per query the rows scouted by all S workgroups of a query tile are ONE
    // sample, S times larger than a workgroup's own, and its seed_rank-th smallest bound starts every segment near
    // the quantile its threshold would only reach at the end of its run.  lambda = expected number of the R
    // relevant rows inside the sample (kept <= 1.2); the seed must stay above them: seed_rank = smallest rank with
    // P(Poisson(lambda) >= rank) <= 1e-7 (a seed that is too low only sends the query to the next tier).
    p.shared_scout = false;
#ifndef PN_DIAG_NO_SHARED_SCOUT
    if (p.ok && level == 0 && ix->filter_slots == 0 && p.split == 1 && per_tile >= 2 &&
        p.n_wg % (int)q_tiles == 0) {
        const size_t run_len = r_tiles / per_tile;
        // How large a sample: its share of the corpus is lambda / R, and the seed lands at the corpus' rank
        // seed_rank(lambda) R / lambda -- ~7.5 R at lambda = 1.2, ~3 R at 9.  For k = 10 (R = 24) lambda = 1.2 is a
        // twentieth of the corpus and the best of 0.3 ... 2.4 (C2: 2.39 / 2.43 / 2.49 ms at 1.2 / 1.8 / 2.4); for k = 100
        // (R = 244) it was 0.5 % of the corpus, every buffer took 135 appends for 42 kept, and larger samples pay:
        // 1M x 128, k = 100: 3.89 / 3.73 / 3.64 / 3.44 / 3.36 / 3.43 ms per step at lambda = 1.2 / 2.4 / 4 / 6 / 9 / 12;
        // 10M x 128, 10^5 queries, k = 100 (same device): 223 (64-tile cap) / 218 / 213 / 209 ms at 1.2 / 1.2 / 4 / 9.
        // The cap of 64 tiles per workgroup (from the in-run scout of round 1) is gone: a nineteenth of the run bounds it.
        double lam_target = R / 27.0 > 1.2 ? R / 27.0 : 1.2;
        if (plan_knobs().scout_lambda > 0.0) lam_target = plan_knobs().scout_lambda;  // experiments only
        double t = lam_target * (double)r_tiles / (R * (double)per_tile);  // tiles per workgroup
        // (never more than a nineteenth of the run -- what lambda = 1.2 asks for at k = 10; k = 1 would ask for a fifth)
        if (t > (double)run_len / 19.0) t = (double)run_len / 19.0;
        p.scout_tiles = (int)t;
        if (p.scout_tiles < 1 && run_len >= 8) p.scout_tiles = 1;  // short runs of many segments: one tile each is a large sample
        // (every (segment, lane half) list keeps its kScoutList smallest: the union's seed_rank-th smallest is exact
        // unless ONE of the 2 nseg lists holds more than kScoutList of them, and then it comes out larger -- a looser
        // seed, never an invalid one.  With rank / (2 nseg) <= 3 expected per list that does not happen; a sample whose
        // rank would exceed that is shrunk until it fits.)
        while (p.scout_tiles >= 4 || (p.scout_tiles >= 1 && run_len < 64)) {
            const double lam = R * (double)p.scout_tiles * (double)per_tile / (double)r_tiles;
            double term = std::exp(-lam), cdf = term;  // P(X <= 0)
            int rank = 1;
            while (1.0 - cdf > 1e-7 && rank < 96) {  // 1 - cdf = P(X >= rank)
                term *= lam / (double)rank;
                cdf += term;
                ++rank;
            }
            p.seed_rank = rank < 5 ? 5 : rank;
            p.shared_scout = p.seed_rank < 96 && p.seed_rank <= 6 * p.nseg;
            if (p.shared_scout || p.scout_tiles < 8) break;
            p.scout_tiles = p.scout_tiles * 3 / 4;
        }
    }
#endif
    // Shared thresholds: with the whole grid resident (main workgroups + refreshers <= the 2 n_cu workgroup slots) the
    // slots the aligned partition leaves idle run refreshers.  r = R + 5.5 sqrt(R) (k = 10: 51): on uniform 1M x 128
    // data the rows whose bound lies below the 10th neighbour's distance number 22 +- 5.3 (max 44 over 256 queries,
    // heavier-tailed than Poisson); a query with r or more of them only goes to the next tier.  Measured over 10^6
    // queries (tools/sh_fallback_rate.py): unproven queries 4 without shared thresholds (a segment's k' overflows),
    // 4 at r = 56 .. 80, 5 at r = 48; r = 40: 5 10^-5, r = 32: 3 10^-3.
    p.first_eval = (int)std::ceil(R);
    // Seed model (seed_model_build): a plan that would launch the shared scout, on an index whose model was accepted at
    // build time, for k <= 128 (measured at k = 1, 10, 100: profiles/r04_seed_model_ab.log) -- the thresholds then come
    // from the query pack kernel and the scout launch is not made.  Where to aim: a threshold must stay above the R
    // relevant rows -- R + 5.5 sqrt(R) covers their spread (the shared thresholds' rank, measured below) -- times
    // exp(4 sigma) for what the model gets wrong per query (sigma measured at build), times 1.5 per sticky widening
    // step.  Unlike the scout's seed (rank ~7.5 R: its sample holds Poisson(1.2) of the relevant rows) the model has no
    // sampling noise, so it starts every segment near the threshold it would end with: C2 kernel 2.32 -> 2.13 ms.
    // (plans WITHOUT a shared scout too -- one workgroup per query tile, grids in rounds: there the model replaces the
    // scout pass every run would make over its own first tiles; PN_EXP_MODEL_ALL_PLANS=0 restricts it to shared-scout plans)
    plan_seed_model(ix, p, R, kout, p.shared_scout || (!plan_knobs().model_shared_only && p.ok && level == 0 && ix->filter_slots == 0));
    p.n_refresh = 0;
    p.sh_rank = 0;
    // (with thresholds from the seed model the segments start where sharing would only bring them later: measured on
    // one device, C2: 2.09-2.11 ms with refreshers, 2.06-2.07 without -- sharing stays for explicit ranks and scouted plans)
    if (p.shared_scout && p.aligned && ix->shared_tau != 0 && !(p.model_seed && ix->shared_tau == 1) &&
        bf16_shared_supported(p.cap)) {
        const int slots = 2 * ix->n_cu - p.n_wg;
        const int rank = ix->shared_tau >= 2 ? ix->shared_tau : (int)std::ceil(R + 5.5 * std::sqrt(R));
        // (a refresher holds up to 512 keys of a query's union in registers -- and at k = 100 (r = 330, 128-slot buffers)
        // sharing measured 6 % SLOWER, 3.44 -> 3.65 ms: there a buffer's own compactions already keep its threshold near
        // the k'-th bound; a refresher pass over the queries takes
        // ~0.25 ms, so short runs end before it pays: a 125 k-row shard of C2, 163 tiles per run, measured 2-4 % slower)
        size_t sh_min_run = 320;  // (same-device A/B: 326-tile runs, a 250 k-row shard of C2, 0.88-0.91 -> 0.875-0.885 ms; 163-tile runs +4 %)
        if (plan_knobs().sh_min_run) sh_min_run = plan_knobs().sh_min_run;  // experiments only
        if (slots >= 4 && rank <= 256 && rank < p.nseg * p.cap && r_tiles / per_tile >= sh_min_run) {
            p.n_refresh = slots > 64 ? 64 : slots;
            p.sh_rank = rank;
        }
    }
    if (plan_knobs().debug)  // development aid: what a call was planned as
        fprintf(stderr, "bf16_plan: n %zu q_tiles %zu k %zu R %.0f: n_wg %d per_tile %zu split %d nseg %d kp %d cap %d aligned %d "
                        "shared_scout %d scout_tiles %d seed_rank %d scout_max %d n_refresh %d sh_rank %d model_seed %d z %.3f\n",
                ix->n, q_tiles, kout, R, p.n_wg, per_tile, p.split, p.nseg, p.kp, p.cap, (int)p.aligned, (int)p.shared_scout,
                p.scout_tiles, p.seed_rank, p.scout_max, p.n_refresh, p.sh_rank, (int)p.model_seed, p.model_z);
    return p;
}
// f32 MFMA filter -> exact re-rank + proof -> (second tier, enqueued by the caller) exact engine for flagged queries
static int run_mfma(const pn_index *ix, Workspace &ws, const float *Qp, size_t nq, size_t nq_pad, size_t kout,
                    uint64_t *d_idx, float *d_dist, size_t out_stride, hipStream_t s, CallRec *rec) {
    // candidate slots kept per (segment, query)
    const size_t kp = mfma_slots(ix, kout, nq_pad);
    // persistent-partition kernels: k' <= 30 with LDS candidate buffers (structure 2, wide rows), k' <= 224 with
    // HBM candidate buffers and two workgroups per CU (structure 3, the default for D <= 128)
    const bool hbm_ok = (ix->mfma_structure == 0 || ix->mfma_structure == 3);
    const bool v2 = kp <= 30 || (hbm_ok && ix->ld <= 128 && kp <= 224);
    if (!v2) return fail(PN_ERR_UNSUPPORTED, ix->ld > 128 ? "wide rows need k' <= 30 on the MFMA path"
                                                            : "k' = %zu is beyond the MFMA path's buffers", kp);
    const int cap = (int)round_up(kp, 32);
    // scaled query norms (same kernel as the corpus norms)
    PNCHK(ws.w_qnorm.ensure(nq_pad * sizeof(float)));
    uint32_t *d_misc = (uint32_t *)ws.w_misc.p;  // [0] flagged count, [1] non-finite query norms, [4] second-tier list length
    HIPCHK(launch_row_norms_f32(Qp, nq_pad, nq, (int)ix->dim, ix->ld, mfma_alpha(ix->dim), (float *)ws.w_qnorm.p,
                                d_misc + 1, s));
    // structure 3: persistent partition with HBM candidate buffers and two workgroups per CU
    const bool two_per_cu = hbm_ok;
    int n_wg = two_per_cu ? 2 * ix->n_cu : ix->n_cu;
    struct { int nseg; } plan{};
    {
        // at most ~32 workgroups per query tile and at least ~32 row tiles per workgroup
        const size_t q_tiles = nq_pad / 128, r_tiles = (ix->n + 63) / 64;
        size_t cap_wg = q_tiles * 32;
        const size_t by_work = (q_tiles * r_tiles + 31) / 32;
        if (by_work < cap_wg) cap_wg = by_work;
        if (cap_wg < 1) cap_wg = 1;
        if ((size_t)n_wg > cap_wg) n_wg = (int)cap_wg;
        plan.nseg = mfma_v2_max_segments(nq_pad / 128, n_wg);
    }
    const size_t cells = (size_t)plan.nseg * nq_pad;
    const size_t slots = cells * (size_t)cap;
    PNCHK(ws.w_keys.ensure(slots * sizeof(uint32_t)));
    PNCHK(ws.w_idx.ensure(slots * sizeof(uint32_t)));
    PNCHK(ws.w_cnt.ensure(cells * sizeof(uint32_t)));
    PNCHK(ws.w_tau.ensure(cells * sizeof(uint32_t)));
    PNCHK(ws.w_flags.ensure(nq_pad * sizeof(uint32_t)));
    PNCHK(ws.w_gsel.ensure(nq_pad * sizeof(uint32_t)));  // the re-rank lists the queries it could not prove here
    CandBuf cb{ws.w_keys.p, (uint32_t *)ws.w_idx.p, (uint32_t *)ws.w_cnt.p, ws.w_tau.p, nq_pad, plan.nseg, cap};
    // not every (segment, query tile) cell is written by the persistent partition
    HIPCHK(hipMemsetAsync(ws.w_cnt.p, 0, cells * sizeof(uint32_t), s));
    HIPCHK(hipMemsetD32Async((hipDeviceptr_t)ws.w_tau.p, (int)0xFF800000u, cells, s));
    const bool prof = rec && rec->prof;
    if (prof) {
        HIPCHK(hipEventRecord(rec->ev[0], s));
        rec->hot = true;
    }
    uint32_t *gcand = nullptr;
    if (two_per_cu) {
        PNCHK(ws.w_sel.ensure(mfma_v2_gcand_bytes(n_wg, (int)kp)));
        gcand = (uint32_t *)ws.w_sel.p;
    }
    HIPCHK(launch_mfma_filter_v2_f32((const float *)ix->d_pts, ix->d_norm, ix->n, ix->ld, Qp, (const float *)ws.w_qnorm.p,
                                     ix->ld, (int)kp, cb, n_wg, gcand, s));
    if (prof) HIPCHK(hipEventRecord(rec->ev[1], s));
    HIPCHK(launch_select_rerank_f32(cb, (const float *)ix->d_pts, ix->n, (int)ix->dim, ix->ld, Qp, (int)nq, ix->ld,
                                    (int)kout, ix->index_base, d_idx, d_dist, out_stride, (uint32_t *)ws.w_flags.p,
                                    d_misc, nullptr, nullptr, (uint32_t *)ws.w_gsel.p, ix->d_stats, s));
    return PN_OK;
}

// bf16 filter -> exact re-rank + proof -> (second tier, enqueued by the caller) exact engine for the unproven queries
// d_q_raw (nullable; narrow rows): the caller's queries, row stride q_stride -- Qp has NOT been filled and the call's
// counters have not been zeroed yet: the query pack kernel does both on the way (one launch at the head of the call)
template <typename T>
static int run_bf16(const pn_index *ix, Workspace &ws, const Bf16Plan &plan, const T *Qp, size_t nq, size_t nq_pad,
                    size_t kout, uint64_t *d_idx, T *d_dist, size_t out_stride, hipStream_t s, CallRec *rec,
                    const T *d_q_raw = nullptr, size_t q_stride = 0, const T *qnorm = nullptr) {
    // qnorm (a Cosine index): the queries' norms in T -- the filter then runs on the queries NORMALISED in f64 against the
    // index's images of the normalised rows, and the re-rank evaluates Cosine::distance (select.hip, cos_proof_lb)
    if (!plan.ok) return fail(PN_ERR_UNSUPPORTED, "bf16 tier cannot serve k = %zu", kout);
    const bool cosine = ix->metric == 1;
    if (cosine && (!qnorm || d_q_raw)) return fail(PN_ERR_INVALID, "Cosine tier: query norms missing");
    const int n_wg = plan.n_wg, cap = plan.cap, nseg = plan.nseg;
    const size_t kp = (size_t)plan.kp;
    const size_t cells = (size_t)nseg * nq_pad, slots = cells * (size_t)cap;
    PNCHK(ws.w_bq.ensure(bf16_query_bytes(nq_pad, (int)ix->dim, ix->bf16_ci)));
    PNCHK(ws.w_qn.ensure(nq_pad * sizeof(double)));
    PNCHK(ws.w_qbad.ensure(nq_pad * sizeof(uint32_t)));
    PNCHK(ws.w_keys.ensure(slots * 2 * sizeof(uint32_t)));  // (key, row) pairs
    PNCHK(ws.w_cnt.ensure(cells * sizeof(uint32_t)));
    PNCHK(ws.w_tau.ensure(cells * sizeof(uint32_t)));
    PNCHK(ws.w_flags.ensure(nq_pad * sizeof(uint32_t)));
    PNCHK(ws.w_gsel.ensure(nq_pad * sizeof(uint32_t)));  // the re-rank lists the queries it could not prove here
    uint32_t *d_misc = (uint32_t *)ws.w_misc.p;
    // thresholds from the index's seed model: the query pack kernel writes them, no scout launch (bf16_plan)
    const bool model = plan.model_seed && ix->d_smodel;
    Bf16SeedModel smd{};
    if (model) {
        PNCHK(ws.w_seed.ensure(nq_pad * sizeof(uint32_t)));
        smd.m1 = ix->d_smodel;
        smd.a = ix->d_smodel + ix->ld;
        smd.b = ix->d_smodel + 2 * ix->ld;
        smd.c0 = ix->sm_c0;
        smd.v0 = ix->sm_v0;
        smd.z = plan.model_z;
        smd.seed_out = (uint32_t *)ws.w_seed.p;
        if (rec) rec->model_seed = true;
    }
    if (cosine) {
        // q~ = q / |q| in f64 (a query whose squared norm lies outside [2^-100, 2^100] becomes NaNs: the pack kernel
        // flags it and the exact engine answers it), then the f64 instantiation of the query pack
        PNCHK(ws.w_qnrm.ensure(nq_pad * ix->ld * sizeof(double)));
        HIPCHK(launch_cos_normalize_rows<T>(Qp, nq, nq_pad, (int)ix->dim, ix->ld, (double *)ws.w_qnrm.p, ix->ld, nullptr,
                                            true, s));
        HIPCHK(launch_bf16_pack_queries<double>((const double *)ws.w_qnrm.p, ix->d_mu, nq, nq_pad, (int)ix->dim, ix->ld,
                                                ws.w_bq.p, (double *)ws.w_qn.p, (uint32_t *)ws.w_qbad.p, ix->bf16_ci,
                                                ix->bf16_bmax, ix->bf16_dmax, s, (double *)nullptr, 0, nullptr,
                                                model ? &smd : nullptr));
    } else if (d_q_raw)
        HIPCHK(launch_bf16_pack_queries(d_q_raw, ix->d_mu, nq, nq_pad, (int)ix->dim, q_stride, ws.w_bq.p,
                                        (double *)ws.w_qn.p, (uint32_t *)ws.w_qbad.p, ix->bf16_ci, ix->bf16_bmax,
                                        ix->bf16_dmax, s, const_cast<T *>(Qp), ix->ld, d_misc, model ? &smd : nullptr));
    else
        HIPCHK(launch_bf16_pack_queries(Qp, ix->d_mu, nq, nq_pad, (int)ix->dim, ix->ld, ws.w_bq.p, (double *)ws.w_qn.p,
                                        (uint32_t *)ws.w_qbad.p, ix->bf16_ci, ix->bf16_bmax, ix->bf16_dmax, s,
                                        (T *)nullptr, 0, nullptr, model ? &smd : nullptr));
    CandBuf cb{ws.w_keys.p, (uint32_t *)ws.w_keys.p + 1, (uint32_t *)ws.w_cnt.p, ws.w_tau.p, nq_pad, nseg, cap, 2};
    cb.final_keep = bf16_cell_max((int)kp, cap, nseg, bf16_is_wide((int)ix->dim));  // what the re-rank sizes its LDS for
    cb.bf16_waves = ix->bf16_waves;
    if (!plan.aligned) {  // cells without a writer must read "empty" (an aligned partition writes every cell)
        HIPCHK(hipMemsetAsync(ws.w_cnt.p, 0, cells * sizeof(uint32_t), s));
        HIPCHK(hipMemsetD32Async((hipDeviceptr_t)ws.w_tau.p, (int)0xFF800000u, cells, s));
    }
    const bool prof = rec && rec->prof;
    if (prof) {
        HIPCHK(hipEventRecord(rec->ev[0], s));
        rec->hot = true;
        rec->two_launches = plan.shared_scout && !model;  // the dominant kernel runs twice per call: both launches are timed
    }
    if (plan.shared_scout) {
        const size_t words = cells * 2 * (size_t)bf16_scout_list();
        PNCHK(ws.w_lists.ensure(words * sizeof(float)));
        PNCHK(ws.w_seed.ensure(nq_pad * sizeof(uint32_t)));
        if (!plan.aligned)
            HIPCHK(hipMemsetD32Async((hipDeviceptr_t)ws.w_lists.p, (int)0x7F800000u, words, s));  // +inf: unused cells
        if (plan.wide && !model)
            HIPCHK(launch_bf16_wide_filter(ix->d_img, ix->n, (int)ix->dim, ws.w_bq.p, (int)kp, cb, plan.n_wg,
                                           plan.scout_tiles, nullptr, false, (float *)ws.w_lists.p, s));
        // narrow rows, up to 32 segments: the main launch derives its starting thresholds from the lists itself (a
        // lane per query in its prologue) -- one launch less in the chain (the seed kernel: 17.6 us on a 10^4-query batch)
        const bool seed_in_kernel = !model && !plan.wide && bf16_seed_in_kernel(nseg) && plan.seed_rank <= bf16_scout_list();
        Bf16Shared seed{};
        if (seed_in_kernel) {
            seed.seed_lists = (const float *)ws.w_lists.p;
            seed.seed_rank = (uint32_t)plan.seed_rank;
            seed.seed_nq = (uint32_t)nq;
            seed.seed_words = (uint32_t *)ws.w_seed.p;
        }
        if (!plan.wide && !model)
            HIPCHK(launch_bf16_filter(ix->d_img, ix->n, (int)ix->dim, ws.w_bq.p, (int)kp, cb, n_wg, 1, plan.scout_tiles,
                                      nullptr, false, (float *)ws.w_lists.p, ix->bf16_ci, s,
                                      seed_in_kernel && plan.n_refresh > 0 ? &seed : nullptr));
        if (!seed_in_kernel && !model)
            HIPCHK(launch_bf16_seed((const float *)ws.w_lists.p, nq_pad, nseg, plan.seed_rank, (uint32_t *)ws.w_seed.p, s, nq));
        if (plan.wide)
            HIPCHK(launch_bf16_wide_filter(ix->d_img, ix->n, (int)ix->dim, ws.w_bq.p, (int)kp, cb, plan.n_wg, 0,
                                           (const uint32_t *)ws.w_seed.p, false, nullptr, s));
        else if (plan.n_refresh > 0) {
            // shared thresholds: published counts tagged with this launch's epoch; d_misc[8] (zeroed with d_misc at the
            // start of the call) counts the main workgroups that are done
            const void *pc_before = ws.w_pcnt.p;
            PNCHK(ws.w_pcnt.ensure(cells * sizeof(uint32_t)));
            if (ws.w_pcnt.p != pc_before) ws.sh_nseg = 0;  // a fresh allocation holds anything
            if (ws.sh_nseg != (size_t)nseg || ws.sh_nq_pad != nq_pad) {  // other geometry: stale words could carry any epoch
                HIPCHK(hipMemsetAsync(ws.w_pcnt.p, 0, cells * sizeof(uint32_t), s));
                ws.sh_nseg = (size_t)nseg;
                ws.sh_nq_pad = nq_pad;
            }
            ws.sh_epoch = ws.sh_epoch % 4095u + 1u;
            Bf16Shared shp = seed;
            shp.pcnt = (uint32_t *)ws.w_pcnt.p;
            shp.done = d_misc + 8;
            shp.epoch = ws.sh_epoch;
            shp.rank = (uint32_t)plan.sh_rank;
            shp.n_refresh = plan.n_refresh;
            HIPCHK(launch_bf16_filter(ix->d_img, ix->n, (int)ix->dim, ws.w_bq.p, (int)kp, cb, n_wg, 1, 0,
                                      (const uint32_t *)ws.w_seed.p, false, nullptr, ix->bf16_ci, s, &shp));
        } else
            HIPCHK(launch_bf16_filter(ix->d_img, ix->n, (int)ix->dim, ws.w_bq.p, (int)kp, cb, n_wg, 1, 0,
                                      (const uint32_t *)ws.w_seed.p, false, nullptr, ix->bf16_ci, s,
                                      seed_in_kernel ? &seed : nullptr));
    } else if (plan.wide) {
        HIPCHK(launch_bf16_wide_filter(ix->d_img, ix->n, (int)ix->dim, ws.w_bq.p, (int)kp, cb, plan.n_wg,
                                       plan.scout_max, nullptr, false, nullptr, s));
    } else if (model) {  // any partition: thresholds from the seed model, no scouting inside the runs
        HIPCHK(launch_bf16_filter(ix->d_img, ix->n, (int)ix->dim, ws.w_bq.p, (int)kp, cb, n_wg, plan.split, 0,
                                  (const uint32_t *)ws.w_seed.p, false, nullptr, ix->bf16_ci, s));
    } else {
        HIPCHK(launch_bf16_filter(ix->d_img, ix->n, (int)ix->dim, ws.w_bq.p, (int)kp, cb, n_wg, plan.split,
                                  plan.scout_max, nullptr, false, nullptr, ix->bf16_ci, s));
    }
    if (prof) HIPCHK(hipEventRecord(rec->ev[1], s));
    if (cosine) {
        HIPCHK(Ops<T>::rerank_cos(cb, (const T *)ix->d_pts, ix->n, (int)ix->dim, ix->ld, Qp, (int)nq, ix->ld, (int)kout,
                                  ix->index_base, d_idx, d_dist, out_stride, (uint32_t *)ws.w_flags.p, d_misc,
                                  (const double *)ws.w_qn.p, (const uint32_t *)ws.w_qbad.p, (uint32_t *)ws.w_gsel.p,
                                  ix->d_stats, s, plan.first_eval, cb.final_keep, (const T *)ix->d_cnorm, qnorm));
        return PN_OK;
    }
    HIPCHK(Ops<T>::rerank(cb, (const T *)ix->d_pts, ix->n, (int)ix->dim, ix->ld, Qp, (int)nq, ix->ld, (int)kout,
                          ix->index_base, d_idx, d_dist, out_stride, (uint32_t *)ws.w_flags.p, d_misc,
                          (const double *)ws.w_qn.p, (const uint32_t *)ws.w_qbad.p, (uint32_t *)ws.w_gsel.p, ix->d_stats, s,
                          plan.first_eval, cb.final_keep));
    return PN_OK;
}

// One k-NN call on queries resident in HBM, everything enqueued on `s`, nothing read back: tier 1 (bf16 filter, f32
// MFMA filter or the exact engine itself) then, behind a filter tier, the device-driven second tier.  Results of
// query q at d_idx/d_dist[q * out_stride + r].
template <typename T>
static int query_enqueue(const pn_index *ix, Workspace &ws, const T *d_q, size_t nq, size_t q_cols, size_t q_stride,
                         size_t kout, uint64_t *d_idx, T *d_dist, size_t out_stride, hipStream_t s) {
    const size_t dim_eff = q_cols < ix->dim ? q_cols : ix->dim;  // zip truncation (src/distance.rs:27-28)
    int level;
    uint64_t call_id;
    {
        std::lock_guard<std::mutex> lk(ix->sh.mu);
        recs_collect(ix, false);  // finished earlier calls: statistics, and whether the bf16 plan must widen
        level = ix->sh.bf16_level;
        call_id = ix->sh.next_call++;
        ix->sh.stats.queries += nq;
    }
    const size_t chunk = 1u << 18;
    for (size_t qs = 0; qs < nq; qs += chunk) {
        const size_t nqc = (nq - qs < chunk) ? nq - qs : chunk;
        const size_t nq_pad = round_up(nqc, (size_t)256);  // query tiles are 64 (exact), 128 (MFMA) or 256 (bf16) rows
        CallRec *rec = nullptr;
        PNCHK(rec_begin(ix, call_id, nqc, &rec));
        RecGuard rec_guard(ix);
        rec_guard.r = rec;  // released on every early return below; rec_end takes over on success
        if (rec->prof > 1) HIPCHK(hipEventRecord(rec->ev[2], s));
        PNCHK(ws.w_q.ensure(nq_pad * ix->ld * sizeof(T)));
        T *Qp = (T *)ws.w_q.p;
        uint64_t *oi = d_idx + qs * out_stride;
        T *od = d_dist + qs * out_stride;
        bool use_mfma = false, use_bf16 = false;
        Bf16Plan bplan{};
        const T *qnorm = nullptr;
        if (ix->metric == 1) {  // Cosine: each query's norm over ITS OWN length (src/distance.rs:92-97), not the zip's
            PNCHK(ws.w_qnorm.ensure(nq_pad * sizeof(T)));
            HIPCHK(hipMemsetAsync(ws.w_qnorm.p, 0, nq_pad * sizeof(T), s));
            HIPCHK(Ops<T>::cnorms(d_q + qs * q_stride, nqc, (int)q_cols, q_stride, (T *)ws.w_qnorm.p, s));
            qnorm = (const T *)ws.w_qnorm.p;
        }
        if constexpr (sizeof(T) == 4) {
            if (ix->mfma_ok && dim_eff == ix->dim && ix->engine != PN_ENGINE_EXACT && ix->engine != PN_ENGINE_BF16) {
                // the filter keeps kp = kout + margin candidates per (segment, query) in <= 256 slots
                use_mfma = mfma_slots(ix, kout, nq_pad) + 64 <= 256;
                if (ix->ld > 128 && mfma_slots(ix, kout, nq_pad) > 30) use_mfma = false;  // wide rows: LDS-buffer kernel only
                if (ix->engine == PN_ENGINE_AUTO && (ix->n < 4096 || ix->dim < 8)) use_mfma = false;
            }
        }
        // (f32 AND f64 indexes: the bf16 bound is a statement about real vectors -- an f64 index's candidates are then
        // re-ranked in the reference's f64 fold and proven with u = 2^-53)
        // (a Cosine index: only for queries of exactly the rows' length -- a longer query's norm runs over ITS length,
        // src/distance.rs:92-97, and the tier's |q~ - p~|^2 = 2 (1 - cos) needs the norm of what enters the dot product)
        if (ix->bf16_ok && dim_eff == ix->dim && (ix->metric != 1 || q_cols == ix->dim) &&
            (ix->engine == PN_ENGINE_BF16 || (ix->engine == PN_ENGINE_AUTO && ix->n >= 4096 && ix->dim >= 8))) {
            bplan = bf16_plan(ix, nq_pad, kout, level);
            use_bf16 = bplan.ok;
            // the re-rank's LDS: (8 or 12) bytes per candidate slot + the query row
            if (use_bf16 && (size_t)bplan.nseg * (size_t)bf16_cell_max(bplan.kp, bplan.cap, bplan.nseg, bf16_is_wide((int)ix->dim)) * (sizeof(T) + 8) +
                                    (ix->dim + 8) * sizeof(T) > 64 * 1024)
                use_bf16 = false;
        }
        // the head of the call: the zero-padded copy of the queries (+ the call's counters, zeroed) -- for narrow rows
        // on the bf16 tier both are by-products of the tier's own query pack kernel (run_bf16)
        const bool fused_head = use_bf16 && !bplan.wide && ix->metric != 1 && bf16_pack_fused_supported((int)ix->dim);
        if (!fused_head) HIPCHK(Ops<T>::pack(d_q + qs * q_stride, nqc, dim_eff, q_stride, Qp, nq_pad, ix->ld, s));
        if (use_bf16 || use_mfma) {
            PNCHK(ws.w_misc.ensure(64));
            if (!fused_head) HIPCHK(hipMemsetAsync(ws.w_misc.p, 0, 64, s));
            uint32_t *d_misc = (uint32_t *)ws.w_misc.p;
            if (use_bf16)
                PNCHK(run_bf16<T>(ix, ws, bplan, (const T *)Qp, nqc, nq_pad, kout, oi, od, out_stride, s, rec,
                                  fused_head ? (const T *)(d_q + qs * q_stride) : nullptr, q_stride, qnorm));
            else if constexpr (sizeof(T) == 4)
                PNCHK(run_mfma(ix, ws, (const float *)Qp, nqc, nq_pad, kout, oi, (float *)od, out_stride, s, rec));
            // the flagged count reaches pinned memory behind everything else (written by the second tier's merge
            // kernel, else by a copy); a LATER call looks at it
            uint32_t *h_dev = nullptr;
            if (hipHostGetDevicePointer((void **)&h_dev, rec->h_nflag, 0) != hipSuccess) h_dev = nullptr;
            bool published = false;
            PNCHK(second_tier_exact<T>(ix, ws, (const T *)Qp, nqc, kout, (const uint32_t *)ws.w_gsel.p, d_misc, oi, od,
                                       out_stride, s, h_dev, &published, qnorm));
            if (!published) HIPCHK(hipMemcpyAsync(rec->h_nflag, d_misc, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
            rec->has_flag = true;
            rec->bf16_tier = use_bf16;
        }
        if (!use_bf16 && !use_mfma)
            PNCHK(run_exact<T>(ix, ws, Qp, nqc, nq_pad, (int)dim_eff, kout, oi, od, out_stride, s, false, nullptr, 0, rec, qnorm));
        if (rec->prof > 1) HIPCHK(hipEventRecord(rec->ev[3], s));
        rec_guard.r = nullptr;
        PNCHK(rec_end(ix, rec, s));
    }
    return PN_OK;
}

template <typename T>
static int query_device_impl(const pn_index *ix, const T *d_q, size_t nq, size_t q_cols, size_t q_stride, size_t k,
                             uint64_t *d_idx, T *d_dist, hipStream_t s) {
    if (!ix) return fail(PN_ERR_INVALID, "index is NULL");
    if (ix->elem_bytes != (int)sizeof(T)) return fail(PN_ERR_INVALID, "index element type mismatch");
    const size_t kout = k < ix->n ? k : ix->n;
    if (nq == 0 || kout == 0) return PN_OK;  // k == 0 -> empty result (src/ball_tree.rs:106-108)
    if (!d_q && q_cols) return fail(PN_ERR_INVALID, "queries is NULL");
    if (!d_idx || !d_dist) return fail(PN_ERR_INVALID, "output buffer is NULL");
    if (nq > 0x7FFFFFFFull) return fail(PN_ERR_UNSUPPORTED, "too many queries in one call");
    DeviceGuard g(ix->device);
    if (!g.ok) return fail(PN_ERR_DEVICE, "hipSetDevice(%d) failed", ix->device);
    // s == NULL is HIP's default stream (what torch.cuda.current_stream() is unless changed): work is
    // ordered with the caller's other default-stream work, as a caller of a *_device entry point expects
    WsLease lease(ix);
    lease.s = s;
    PNCHK(ws_acquire(ix, &lease.s, false, &lease.ws));
    return query_enqueue<T>(ix, *lease.ws, d_q, nq, q_cols, q_stride, kout, d_idx, d_dist, kout, s);
}

namespace pn {
// pn_query_device_f32 with a row stride on the outputs (results of query q at [q * out_stride + r], r < min(k, n)):
// a shard writes straight into the packed buffer the all-gather sends (sharded.hip)
template <typename T>
static int query_device_strided_impl(const pn_index *ix, const T *d_q, size_t nq, size_t q_cols, size_t q_stride, size_t k,
                                     uint64_t *d_idx, T *d_dist, size_t out_stride, hipStream_t s) {
    if (!ix) return fail(PN_ERR_INVALID, "index is NULL");
    if (ix->elem_bytes != (int)sizeof(T)) return fail(PN_ERR_INVALID, "index element type mismatch");
    const size_t kout = k < ix->n ? k : ix->n;
    if (nq == 0 || kout == 0) return PN_OK;
    if ((!d_q && q_cols) || !d_idx || !d_dist || out_stride < kout) return fail(PN_ERR_INVALID, "bad argument");
    if (nq > 0x7FFFFFFFull) return fail(PN_ERR_UNSUPPORTED, "too many queries in one call");
    DeviceGuard g(ix->device);
    if (!g.ok) return fail(PN_ERR_DEVICE, "hipSetDevice(%d) failed", ix->device);
    WsLease lease(ix);
    lease.s = s;
    PNCHK(ws_acquire(ix, &lease.s, false, &lease.ws));
    return query_enqueue<T>(ix, *lease.ws, d_q, nq, q_cols, q_stride, kout, d_idx, d_dist, out_stride, s);
}
int query_device_strided_f32(const pn_index *ix, const float *d_q, size_t nq, size_t q_cols, size_t q_stride, size_t k,
                             uint64_t *d_idx, float *d_dist, size_t out_stride, hipStream_t s) {
    return query_device_strided_impl<float>(ix, d_q, nq, q_cols, q_stride, k, d_idx, d_dist, out_stride, s);
}
int query_device_strided_f64(const pn_index *ix, const double *d_q, size_t nq, size_t q_cols, size_t q_stride, size_t k,
                             uint64_t *d_idx, double *d_dist, size_t out_stride, hipStream_t s) {
    return query_device_strided_impl<double>(ix, d_q, nq, q_cols, q_stride, k, d_idx, d_dist, out_stride, s);
}
}  // namespace pn

extern "C" int pn_query_device_f32(const pn_index *ix, const float *d_q, size_t nq, size_t q_cols, size_t q_stride,
                                   size_t k, uint64_t *d_idx, float *d_dist, void *stream) {
    return query_device_impl<float>(ix, d_q, nq, q_cols, q_stride, k, d_idx, d_dist, (hipStream_t)stream);
}
extern "C" int pn_query_device_f64(const pn_index *ix, const double *d_q, size_t nq, size_t q_cols, size_t q_stride,
                                   size_t k, uint64_t *d_idx, double *d_dist, void *stream) {
    return query_device_impl<double>(ix, d_q, nq, q_cols, q_stride, k, d_idx, d_dist, (hipStream_t)stream);
}

// host staging into a pooled buffer: rows [rows][cols] contiguous on the device
template <typename T>
static int upload_rows_to(const T *h, size_t rows, size_t cols, ptrdiff_t row_stride, DevBuf &buf, hipStream_t s) {
    const size_t c = cols ? cols : 1;
    PNCHK(buf.ensure(rows * c * sizeof(T)));
    if (cols == 0) return PN_OK;
    if (row_stride < 0) return fail(PN_ERR_UNSUPPORTED, "negative row stride");
    T *d = (T *)buf.p;
    if (rows == 1 || (size_t)row_stride == cols)
        HIPCHK(hipMemcpyAsync(d, h, rows * cols * sizeof(T), hipMemcpyHostToDevice, s));
    else if (row_stride == 0) {
        for (size_t r = 0; r < rows; ++r)
            HIPCHK(hipMemcpyAsync(d + r * cols, h, cols * sizeof(T), hipMemcpyHostToDevice, s));
    } else
        HIPCHK(hipMemcpy2DAsync(d, cols * sizeof(T), h, (size_t)row_stride * sizeof(T), cols * sizeof(T), rows,
                                hipMemcpyHostToDevice, s));
    return PN_OK;
}
// one-off staging (diagnostics, pairwise): own allocation
template <typename T>
static int upload_rows(const T *h, size_t rows, size_t cols, ptrdiff_t row_stride, T **d_out) {
    *d_out = nullptr;
    const size_t c = cols ? cols : 1;
    HIPCHK(hipMalloc((void **)d_out, rows * c * sizeof(T)));
    if (cols == 0) return PN_OK;
    if (row_stride < 0) return fail(PN_ERR_UNSUPPORTED, "negative row stride");
    if (rows == 1 || (size_t)row_stride == cols)
        HIPCHK(hipMemcpy(*d_out, h, rows * cols * sizeof(T), hipMemcpyHostToDevice));
    else if (row_stride == 0) {
        for (size_t r = 0; r < rows; ++r)
            HIPCHK(hipMemcpy(*d_out + r * cols, h, cols * sizeof(T), hipMemcpyHostToDevice));
    } else
        HIPCHK(hipMemcpy2D(*d_out, cols * sizeof(T), h, (size_t)row_stride * sizeof(T), cols * sizeof(T), rows,
                           hipMemcpyHostToDevice));
    return PN_OK;
}

// Host entry point: what the reference's one-point-per-call API maps to (src/ball_tree.rs:102; nq = 1 is that call).
// Staging buffers come from the workspace (no allocation per call once warm); the call runs on the workspace's own
// stream, so host threads sharing one handle do not serialise each other.
// ---- small corpora, a few queries per call: the reference's own call pattern (one point per BallTree::query /
// query_radius call, benches/ball_tree.rs:22-62).  The batched pipeline costs such a call a dozen launches and three copy
// commands (round 2: 72 us against the CPU's 19 us at 64 x 10 f64); here the call is ONE launch that reads the query from
// mapped pinned memory, scans the rows with one wave per query (select.hip, tiny_query_kernel) and writes the answer to
// mapped pinned memory, and one wait for the stream.
constexpr size_t kTinyRows = 4096, kTinyQueries = 64;
template <typename T>
static bool tiny_eligible(const pn_index *ix, size_t nq, size_t dim_eff) {
    return ix->metric == 0 && ix->n <= kTinyRows && nq <= kTinyQueries &&
           (ix->engine == PN_ENGINE_AUTO || ix->engine == PN_ENGINE_EXACT) &&
           ix->n * (dim_eff ? dim_eff : 1) * nq <= ((size_t)1 << 22) &&  // (beyond: the tiled engine's parallelism pays)
           tiny_query_lds_bytes(ix->n, (int)dim_eff, (int)sizeof(T)) <= 64 * 1024;
}
template <typename T>
static hipError_t tiny_launch(const pn_index *ix, size_t dim_eff, const T *Qd, size_t ldq, size_t nq, size_t kout, bool rad,
                              T radius, uint64_t *io, T *dd, size_t os, hipStream_t s) {
    if constexpr (sizeof(T) == 4)
        return launch_tiny_query_f32((const float *)ix->d_pts, ix->n, (int)dim_eff, ix->ld, Qd, ldq, (int)nq, (int)kout, rad,
                                     radius, ix->index_base, io, dd, os, s);
    else
        return launch_tiny_query_f64((const double *)ix->d_pts, ix->n, (int)dim_eff, ix->ld, Qd, ldq, (int)nq, (int)kout, rad,
                                     radius, ix->index_base, io, dd, os, s);
}
// stages the queries in the workspace's pinned input buffer; *Qd = its device address
template <typename T>
static int tiny_stage(Workspace &ws, const T *q, size_t nq, size_t q_cols, ptrdiff_t q_stride, const T **Qd) {
    const size_t qc = q_cols ? q_cols : 1;
    PNCHK(ws.pin_ensure(&ws.pin_in, &ws.pin_in_bytes, nq * qc * sizeof(T)));
    T *h = (T *)ws.pin_in;
    for (size_t a = 0; a < nq; ++a)
        if (q_cols) memcpy(h + a * qc, q + (ptrdiff_t)a * q_stride, q_cols * sizeof(T));
    void *d = nullptr;
    HIPCHK(hipHostGetDevicePointer(&d, ws.pin_in, 0));
    *Qd = (const T *)d;
    return PN_OK;
}
template <typename T>
static int tiny_knn(const pn_index *ix, Workspace &ws, hipStream_t s, const T *q, size_t nq, size_t q_cols,
                    ptrdiff_t q_stride, size_t kout, uint64_t *idx_out, T *dist_out) {
    const size_t dim_eff = q_cols < ix->dim ? q_cols : ix->dim, qc = q_cols ? q_cols : 1;
    const T *Qd = nullptr;
    PNCHK(tiny_stage<T>(ws, q, nq, q_cols, q_stride, &Qd));
    const size_t ib = nq * kout * sizeof(uint64_t), db = nq * kout * sizeof(T);
    PNCHK(ws.pin_ensure(&ws.pin_out, &ws.pin_out_bytes, ib + db));
    void *od = nullptr;
    HIPCHK(hipHostGetDevicePointer(&od, ws.pin_out, 0));
    HIPCHK(tiny_launch<T>(ix, dim_eff, Qd, qc, nq, kout, false, (T)0, (uint64_t *)od, (T *)((char *)od + ib), kout, s));
    HIPCHK(hipStreamSynchronize(s));
    memcpy(idx_out, ws.pin_out, ib);
    memcpy(dist_out, (char *)ws.pin_out + ib, db);
    std::lock_guard<std::mutex> lk(ix->sh.mu);
    ix->sh.stats.queries += nq;
    return PN_OK;
}
template <typename T>
static int tiny_radius(const pn_index *ix, Workspace &ws, hipStream_t s, const T *q, size_t nq, size_t q_cols,
                       ptrdiff_t q_stride, T radius, uint64_t *offsets, uint64_t **idx_out) {
    const size_t dim_eff = q_cols < ix->dim ? q_cols : ix->dim, qc = q_cols ? q_cols : 1;
    const T *Qd = nullptr;
    PNCHK(tiny_stage<T>(ws, q, nq, q_cols, q_stride, &Qd));
    const size_t os = ix->n + 1;  // per query: count, then up to n rows
    PNCHK(ws.pin_ensure(&ws.pin_out, &ws.pin_out_bytes, nq * os * sizeof(uint64_t)));
    void *od = nullptr;
    HIPCHK(hipHostGetDevicePointer(&od, ws.pin_out, 0));
    HIPCHK(tiny_launch<T>(ix, dim_eff, Qd, qc, nq, 0, true, radius, (uint64_t *)od, (T *)nullptr, os, s));
    HIPCHK(hipStreamSynchronize(s));
    const uint64_t *h = (const uint64_t *)ws.pin_out;
    uint64_t total = 0;
    for (size_t a = 0; a < nq; ++a) {
        offsets[a] = total;
        total += h[a * os];
    }
    offsets[nq] = total;
    uint64_t *out = (uint64_t *)malloc((total ? total : 1) * sizeof(uint64_t));
    if (!out) return fail(PN_ERR_NOMEM, "malloc(%llu results) failed", (unsigned long long)total);
    for (size_t a = 0; a < nq; ++a) memcpy(out + offsets[a], h + a * os + 1, (size_t)h[a * os] * sizeof(uint64_t));
    *idx_out = out;
    std::lock_guard<std::mutex> lk(ix->sh.mu);
    ix->sh.stats.radius_results += total;
    return PN_OK;
}

template <typename T>
static int query_host_impl(const pn_index *ix, const T *q, size_t nq, size_t q_cols, ptrdiff_t q_stride, size_t k,
                           uint64_t *idx_out, T *dist_out) {
    if (!ix) return fail(PN_ERR_INVALID, "index is NULL");
    if (ix->elem_bytes != (int)sizeof(T)) return fail(PN_ERR_INVALID, "index element type mismatch");
    const size_t kout = k < ix->n ? k : ix->n;
    if (nq == 0 || kout == 0) return PN_OK;
    if (!q && q_cols) return fail(PN_ERR_INVALID, "queries is NULL");
    if (!idx_out || !dist_out) return fail(PN_ERR_INVALID, "output buffer is NULL");
    if (nq > 0x7FFFFFFFull) return fail(PN_ERR_UNSUPPORTED, "too many queries in one call");
    DeviceGuard g(ix->device);
    if (!g.ok) return fail(PN_ERR_DEVICE, "hipSetDevice(%d) failed", ix->device);
    WsLease lease(ix);
    PNCHK(ws_acquire(ix, &lease.s, true, &lease.ws));
    Workspace &ws = *lease.ws;
    hipStream_t s = lease.s;
    if (q_stride >= 0 && tiny_eligible<T>(ix, nq, q_cols < ix->dim ? q_cols : ix->dim))
        return tiny_knn<T>(ix, ws, s, q, nq, q_cols, q_stride, kout, idx_out, dist_out);
    // a small batch against a large corpus: the kernels read the queries from, and write the answers to, mapped pinned
    // memory -- three copy commands (~10 us each) less on a call whose whole GPU time is ~100 us
    if (q_stride >= 0 && nq * (q_cols ? q_cols : 1) * sizeof(T) <= (64u << 10) && nq * kout * 16 <= (64u << 10)) {
        const T *Qd = nullptr;
        PNCHK(tiny_stage<T>(ws, q, nq, q_cols, q_stride, &Qd));
        const size_t ib = nq * kout * sizeof(uint64_t), db = nq * kout * sizeof(T);
        PNCHK(ws.pin_ensure(&ws.pin_out, &ws.pin_out_bytes, ib + db));
        void *od = nullptr;
        HIPCHK(hipHostGetDevicePointer(&od, ws.pin_out, 0));
        PNCHK(query_enqueue<T>(ix, ws, Qd, nq, q_cols, q_cols ? q_cols : 1, kout, (uint64_t *)od, (T *)((char *)od + ib),
                               kout, s));
        HIPCHK(hipStreamSynchronize(s));
        memcpy(idx_out, ws.pin_out, ib);
        memcpy(dist_out, (char *)ws.pin_out + ib, db);
        return PN_OK;
    }
    PNCHK(upload_rows_to<T>(q, nq, q_cols, q_stride, ws.w_hq, s));
    PNCHK(ws.w_hidx.ensure(nq * kout * sizeof(uint64_t)));
    PNCHK(ws.w_hdist.ensure(nq * kout * sizeof(T)));
    PNCHK(query_enqueue<T>(ix, ws, (const T *)ws.w_hq.p, nq, q_cols, q_cols ? q_cols : 1, kout, (uint64_t *)ws.w_hidx.p,
                           (T *)ws.w_hdist.p, kout, s));
    HIPCHK(hipMemcpyAsync(idx_out, ws.w_hidx.p, nq * kout * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(dist_out, ws.w_hdist.p, nq * kout * sizeof(T), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return PN_OK;
}

extern "C" int pn_query_f32(const pn_index *ix, const float *q, size_t nq, size_t q_cols, ptrdiff_t q_stride, size_t k,
                            uint64_t *idx_out, float *dist_out) {
    return query_host_impl<float>(ix, q, nq, q_cols, q_stride, k, idx_out, dist_out);
}
extern "C" int pn_query_f64(const pn_index *ix, const double *q, size_t nq, size_t q_cols, ptrdiff_t q_stride,
                            size_t k, uint64_t *idx_out, double *dist_out) {
    return query_host_impl<double>(ix, q, nq, q_cols, q_stride, k, idx_out, dist_out);
}
extern "C" int pn_query_nearest_f32(const pn_index *ix, const float *q, size_t nq, size_t q_cols, ptrdiff_t q_stride,
                                    uint64_t *idx_out, float *dist_out) {
    return query_host_impl<float>(ix, q, nq, q_cols, q_stride, 1, idx_out, dist_out);
}
extern "C" int pn_query_nearest_f64(const pn_index *ix, const double *q, size_t nq, size_t q_cols, ptrdiff_t q_stride,
                                    uint64_t *idx_out, double *dist_out) {
    return query_host_impl<double>(ix, q, nq, q_cols, q_stride, 1, idx_out, dist_out);
}

// diagnostic: one observation into the handle's seed-model feedback (see the header)
extern "C" int pn_debug_seed_model_feedback(pn_index *ix, int model_seed, uint64_t unproven, uint64_t nq, int32_t *out4) {
    if (!ix || !out4) return fail(PN_ERR_INVALID, "NULL argument");
    std::lock_guard<std::mutex> lk(ix->sh.mu);
    (void)seed_model_feedback(ix->sh, model_seed != 0, (size_t)unproven, (size_t)nq);
    out4[0] = ix->sh.seed_model_off.load() ? 1 : 0;
    out4[1] = ix->sh.seed_model_widen.load();
    out4[2] = ix->sh.seed_model_off_calls;
    out4[3] = ix->sh.seed_model_retries;
    return PN_OK;
}

// diagnostic: the bf16 filter's lower bounds themselves (see the header)
extern "C" int pn_bf16_selftest(int device, float *ratio_out) {
    if (!ratio_out) return fail(PN_ERR_INVALID, "ratio_out is NULL");
    PNCHK(check_device(device));
    DeviceGuard g(device);
    if (!g.ok) return fail(PN_ERR_DEVICE, "hipSetDevice(%d) failed", device);
    DevTmp d;
    PinnedBlock h;
    HIPCHK(d.alloc(sizeof(float)));
    HIPCHK(h.acquire());
    hipStream_t s = pooled_stream_acquire(device);
    if (!s) return fail(PN_ERR_DEVICE, "hipStreamCreate failed");
    hipError_t e = launch_bf16_selftest((float *)d.p, s);
    if (e == hipSuccess) e = hipMemcpyAsync(h.h, d.p, sizeof(float), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) {
        (void)hipStreamSynchronize(s);
        pooled_stream_release(device, s);
        return fail(PN_ERR_DEVICE, "bf16 self-test: %s", hipGetErrorString(e));
    }
    pooled_stream_release(device, s);
    *ratio_out = *(float *)h.h;
    return PN_OK;
}

extern "C" int pn_bf16_bounds_f32(const pn_index *ix, const float *q, size_t nq, size_t q_cols, ptrdiff_t q_stride,
                                  size_t n_rows, float *bounds_out, double *qnorm_out, float *mu_out) {
    if (!ix || !q || !bounds_out) return fail(PN_ERR_INVALID, "NULL argument");
    if (ix->elem_bytes != 4 || !ix->bf16_ok) return fail(PN_ERR_UNSUPPORTED, "index has no bf16 tier");
    if (q_cols != ix->dim) return fail(PN_ERR_INVALID, "queries must have the index's dimension");
    if (nq == 0 || n_rows == 0) return PN_OK;
    if (n_rows > ix->n) n_rows = ix->n;
    if (nq * n_rows > ((size_t)1 << 28)) return fail(PN_ERR_INVALID, "too many bounds requested");
    DeviceGuard g(ix->device);
    if (!g.ok) return fail(PN_ERR_DEVICE, "hipSetDevice(%d) failed", ix->device);
    WsLease lease(ix);
    PNCHK(ws_acquire(ix, &lease.s, true, &lease.ws));
    Workspace &ws = *lease.ws;
    hipStream_t s = lease.s;
    float *d_q = nullptr, *d_out = nullptr;
    int rc = upload_rows<float>(q, nq, q_cols, q_stride, &d_q);
    do {
        if (rc != PN_OK) break;
        const size_t nq_pad = round_up(nq, (size_t)256);
        rc = PN_ERR_DEVICE;
        if (ws.w_q.ensure(nq_pad * ix->ld * sizeof(float)) != PN_OK ||
            ws.w_bq.ensure(bf16_query_bytes(nq_pad, (int)ix->dim, ix->bf16_ci)) != PN_OK ||
            ws.w_qn.ensure(nq_pad * sizeof(double)) != PN_OK || ws.w_qbad.ensure(nq_pad * sizeof(uint32_t)) != PN_OK)
            break;
        if (hipMalloc((void **)&d_out, nq * n_rows * sizeof(float)) != hipSuccess) {
            rc = fail(PN_ERR_NOMEM, "hipMalloc bounds failed");
            break;
        }
        float *Qp = (float *)ws.w_q.p;
        if (launch_pack_rows_f32(d_q, nq, ix->dim, q_cols, Qp, nq_pad, ix->ld, s) != hipSuccess ||
            launch_bf16_pack_queries(Qp, ix->d_mu, nq, nq_pad, (int)ix->dim, ix->ld, ws.w_bq.p, (double *)ws.w_qn.p,
                                     (uint32_t *)ws.w_qbad.p, ix->bf16_ci, ix->bf16_bmax, ix->bf16_dmax, s) != hipSuccess ||
            launch_bf16_bound(ix->d_img, ws.w_bq.p, n_rows, nq, (int)ix->dim, d_out, ix->bf16_ci, s) != hipSuccess ||
            hipStreamSynchronize(s) != hipSuccess ||
            hipMemcpy(bounds_out, d_out, nq * n_rows * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess ||
            (qnorm_out && hipMemcpy(qnorm_out, ws.w_qn.p, nq * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) ||
            (mu_out && hipMemcpy(mu_out, ix->d_mu, ix->dim * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess)) {
            rc = fail(PN_ERR_DEVICE, "bf16 bounds failed: %s", hipGetErrorString(hipGetLastError()));
            break;
        }
        rc = PN_OK;
    } while (0);
    if (d_q) (void)hipFree(d_q);
    if (d_out) (void)hipFree(d_out);
    return rc;
}

// ---------------------------------------------------------------------------
// radius (two-pass CSR on the exact engine)
// ---------------------------------------------------------------------------
template <typename T> struct RadOps;
template <> struct RadOps<float> {
    static hipError_t run(const float *P, size_t n, int dim, size_t ldp, const float *Q, int nq, size_t ldq, float r,
                          size_t seg_len, int nseg, uint32_t *counts, const uint64_t *offs, uint64_t *fill,
                          uint64_t base, const float *pn, const float *qn, hipStream_t s, const uint32_t *qsel = nullptr,
                          const uint32_t *nq_dev = nullptr, uint64_t capacity = ~0ull) {
        return launch_exact_radius_f32(P, n, dim, ldp, Q, nq, ldq, r, seg_len, nseg, counts, offs, fill, base, pn, qn, s, qsel,
                                       nq_dev, capacity);
    }
};
template <> struct RadOps<double> {
    static hipError_t run(const double *P, size_t n, int dim, size_t ldp, const double *Q, int nq, size_t ldq, double r,
                          size_t seg_len, int nseg, uint32_t *counts, const uint64_t *offs, uint64_t *fill,
                          uint64_t base, const double *pn, const double *qn, hipStream_t s, const uint32_t *qsel = nullptr,
                          const uint32_t *nq_dev = nullptr, uint64_t capacity = ~0ull) {
        return launch_exact_radius_f64(P, n, dim, ldp, Q, nq, ldq, r, seg_len, nseg, counts, offs, fill, base, pn, qn, s, qsel,
                                       nq_dev, capacity);
    }
};

// query_radius through the MFMA filter (f32, D <= 128, finite positive r).  *done = false means the
// caller must run the exact two-pass engine instead (a survivor list overflowed, or a query norm is not
// finite): correctness never depends on this path being taken.
// exact radius queries: count pass -> host exclusive scan -> fill pass (exact scan kernel); offs gets nq + 1
// entries, *out a malloc'ed array of offs[nq] global row numbers, ascending per query
template <typename T>
static int radius_exact(const pn_index *ix, Workspace &ws, const T *Qp, size_t nq, size_t nq_pad, size_t dim_eff, T radius,
                        std::vector<uint64_t> &offs, uint64_t **out, hipStream_t s, const T *qnorm = nullptr) {
    *out = nullptr;
    uint32_t *d_counts = nullptr;
    uint64_t *d_offs = nullptr, *d_fill = nullptr;
    std::vector<uint32_t> h_counts;
    std::vector<uint64_t> h_offs;
    int rc = PN_OK;
    do {
        // (a handful of query tiles -- the few queries whose survivor lists overflowed behind a filter -- are spread
        // over up to 1024 row segments so that every CU scans: with 64 segments three re-run queries of a 10^5-query
        // batch against 10^7 rows cost 82 of the step's 276 ms)
        const size_t q_tiles_r = nq_pad / kTileQ;
        const ScanPlan pl = plan_segments(ix->n, q_tiles_r, 1, ix->opt_segments, 4096, q_tiles_r <= 8 ? 1024 : 64);
        const size_t cells = nq * (size_t)pl.nseg;
        if (hipMalloc((void **)&d_counts, cells * 4) != hipSuccess || hipMalloc((void **)&d_offs, cells * 8) != hipSuccess) {
            rc = fail(PN_ERR_NOMEM, "hipMalloc radius scratch failed");
            break;
        }
        if (hipMemsetAsync(d_counts, 0, cells * 4, s) != hipSuccess ||
            RadOps<T>::run((const T *)ix->d_pts, ix->n, (int)dim_eff, ix->ld, Qp, (int)nq, ix->ld, radius, pl.seg_len,
                           pl.nseg, d_counts, nullptr, nullptr, ix->index_base, qnorm ? (const T *)ix->d_cnorm : nullptr,
                           qnorm, s) != hipSuccess) {
            rc = fail(PN_ERR_DEVICE, "radius count pass failed: %s", hipGetErrorString(hipGetLastError()));
            break;
        }
        h_counts.resize(cells);
        h_offs.resize(cells);
        if (hipMemcpyAsync(h_counts.data(), d_counts, cells * 4, hipMemcpyDeviceToHost, s) != hipSuccess ||
            hipStreamSynchronize(s) != hipSuccess) {
            rc = fail(PN_ERR_DEVICE, "radius count copy failed: %s", hipGetErrorString(hipGetLastError()));
            break;
        }
        offs.assign(nq + 1, 0);
        uint64_t run = 0;
        for (size_t a = 0; a < nq; ++a) {
            offs[a] = run;
            for (int sg = 0; sg < pl.nseg; ++sg) {
                h_offs[a * pl.nseg + sg] = run;
                run += h_counts[a * pl.nseg + sg];
            }
        }
        offs[nq] = run;
        {
            std::lock_guard<std::mutex> lk(ix->sh.mu);
            ix->sh.stats.radius_results += run;
        }
        uint64_t *h_out = (uint64_t *)malloc((run ? run : 1) * sizeof(uint64_t));
        if (!h_out) { rc = fail(PN_ERR_NOMEM, "malloc(%llu results) failed", (unsigned long long)run); break; }
        *out = h_out;
        if (run == 0) break;
        if (hipMalloc((void **)&d_fill, run * 8) != hipSuccess) { rc = fail(PN_ERR_NOMEM, "hipMalloc radius output failed"); break; }
        if (hipMemcpyAsync(d_offs, h_offs.data(), cells * 8, hipMemcpyHostToDevice, s) != hipSuccess ||
            RadOps<T>::run((const T *)ix->d_pts, ix->n, (int)dim_eff, ix->ld, Qp, (int)nq, ix->ld, radius, pl.seg_len,
                           pl.nseg, d_counts, d_offs, d_fill, ix->index_base, qnorm ? (const T *)ix->d_cnorm : nullptr, qnorm,
                           s) != hipSuccess ||
            hipMemcpyAsync(h_out, d_fill, run * 8, hipMemcpyDeviceToHost, s) != hipSuccess ||
            hipStreamSynchronize(s) != hipSuccess) {
            rc = fail(PN_ERR_DEVICE, "radius fill pass failed: %s", hipGetErrorString(hipGetLastError()));
            break;
        }
    } while (0);
    if (rc != PN_OK && *out) { free(*out); *out = nullptr; }
    if (d_counts) (void)hipFree(d_counts);
    if (d_offs) (void)hipFree(d_offs);
    if (d_fill) (void)hipFree(d_fill);
    return rc;
}

// common tail of the filtered radius paths: kept rows per query (w_keys, ascending) -> CSR on the host
template <typename T>
static int radius_finish(const pn_index *ix, Workspace &ws, const T *Qp, size_t nq, size_t kept_stride, uint32_t *d_misc,
                         const uint32_t *d_over, T radius, uint64_t *offsets, uint64_t **idx_out, bool *done,
                         hipStream_t s, bool cosine = false) {
    uint32_t h_misc[2] = {0, 0};
    std::vector<uint32_t> h_n(nq);
    HIPCHK(hipMemcpyAsync(h_misc, d_misc, sizeof h_misc, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(h_n.data(), ws.w_flags.p, nq * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (h_misc[1]) return PN_OK;                        // queries the filter cannot serve: next tier
    if (h_misc[0] && (!d_over || h_misc[0] * 4 > nq)) return PN_OK;  // many overflowed lists: next tier for the call
    // a few queries overflowed their survivor lists (dense neighbourhoods): only they are re-run exactly
    std::vector<uint32_t> sel;
    if (h_misc[0]) {
        std::vector<uint32_t> h_over(nq);
        HIPCHK(hipMemcpy(h_over.data(), d_over, nq * sizeof(uint32_t), hipMemcpyDeviceToHost));
        for (size_t a = 0; a < nq; ++a)
            if (h_over[a]) {
                sel.push_back((uint32_t)a);
                h_n[a] = 0;
            }
        HIPCHK(hipMemcpyAsync(ws.w_flags.p, h_n.data(), nq * sizeof(uint32_t), hipMemcpyHostToDevice, s));
    }
    std::vector<uint64_t> h_off(nq + 1);
    uint64_t run = 0;
    for (size_t a = 0; a < nq; ++a) {
        h_off[a] = run;
        run += h_n[a];
    }
    h_off[nq] = run;
    uint64_t *h_out = (uint64_t *)malloc((run ? run : 1) * sizeof(uint64_t));
    if (!h_out) return fail(PN_ERR_NOMEM, "malloc(%llu results) failed", (unsigned long long)run);
    if (run) {
        uint64_t *d_off = nullptr, *d_out = nullptr;
        int rc = PN_OK;
        if (hipMalloc((void **)&d_off, (nq + 1) * 8) != hipSuccess || hipMalloc((void **)&d_out, run * 8) != hipSuccess)
            rc = fail(PN_ERR_NOMEM, "hipMalloc radius output failed");
        if (rc == PN_OK &&
            (hipMemcpyAsync(d_off, h_off.data(), (nq + 1) * 8, hipMemcpyHostToDevice, s) != hipSuccess ||
             launch_radius_gather((const uint32_t *)ws.w_keys.p, (const uint32_t *)ws.w_flags.p, d_off, (int)nq,
                                  kept_stride, ix->index_base, d_out, s) != hipSuccess ||
             hipMemcpyAsync(h_out, d_out, run * 8, hipMemcpyDeviceToHost, s) != hipSuccess ||
             hipStreamSynchronize(s) != hipSuccess))
            rc = fail(PN_ERR_DEVICE, "radius gather failed: %s", hipGetErrorString(hipGetLastError()));
        if (d_off) (void)hipFree(d_off);
        if (d_out) (void)hipFree(d_out);
        if (rc != PN_OK) {
            free(h_out);
            return rc;
        }
    }
    {
        std::lock_guard<std::mutex> lk(ix->sh.mu);
        ix->sh.stats.radius_results += run;
    }
    if (!sel.empty()) {
        const size_t nf = sel.size(), nf_pad = round_up(nf, (size_t)kTileQ);  // whole query tiles of the exact engine
        std::vector<uint64_t> offs_x;
        uint64_t *out_x = nullptr;
        int rc = PN_OK;
        sel.push_back((uint32_t)nf);  // the gather kernel reads its row count from the device: kept behind the list
        if (ws.w_gsel.ensure((nf + 1) * sizeof(uint32_t)) != PN_OK || ws.w_gq.ensure(nf_pad * ix->ld * sizeof(T)) != PN_OK ||
            hipMemsetAsync(ws.w_gq.p, 0, nf_pad * ix->ld * sizeof(T), s) != hipSuccess ||
            hipMemcpyAsync(ws.w_gsel.p, sel.data(), (nf + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, s) != hipSuccess ||
            launch_gather_rows<T>(Qp, ix->ld, (const uint32_t *)ws.w_gsel.p, (const uint32_t *)ws.w_gsel.p + nf, 0,
                                  (uint32_t)nf, (T *)ws.w_gq.p, s) != hipSuccess ||
            hipStreamSynchronize(s) != hipSuccess)  // `sel` is pageable host memory: the copy must have left it
            rc = fail(PN_ERR_DEVICE, "radius fallback staging failed: %s", hipGetErrorString(hipGetLastError()));
        const T *gqnorm = nullptr;
        if (rc == PN_OK && cosine) {  // a Cosine index: the gathered queries' norms (their rows are dim long: over the row)
            if (ws.w_gqn.ensure(nf_pad * sizeof(T)) != PN_OK || hipMemsetAsync(ws.w_gqn.p, 0, nf_pad * sizeof(T), s) != hipSuccess ||
                Ops<T>::cnorms((const T *)ws.w_gq.p, nf, (int)ix->dim, ix->ld, (T *)ws.w_gqn.p, s) != hipSuccess)
                rc = fail(PN_ERR_DEVICE, "radius fallback query norms failed");
            gqnorm = (const T *)ws.w_gqn.p;
        }
        if (rc == PN_OK)
            rc = radius_exact<T>(ix, ws, (const T *)ws.w_gq.p, nf, nf_pad, ix->dim, radius, offs_x, &out_x, s, gqnorm);
        if (rc != PN_OK) {
            free(h_out);
            return rc;
        }
        // splice: overflowed queries take the exact lists, the others keep the filter's
        const uint64_t total = run + offs_x[nf];
        uint64_t *h_all = (uint64_t *)malloc((total ? total : 1) * sizeof(uint64_t));
        if (!h_all) {
            free(h_out);
            free(out_x);
            return fail(PN_ERR_NOMEM, "malloc(%llu results) failed", (unsigned long long)total);
        }
        uint64_t w = 0;
        size_t j = 0;
        for (size_t a = 0; a < nq; ++a) {
            offsets[a] = w;
            if (j < nf && sel[j] == a) {
                const uint64_t c = offs_x[j + 1] - offs_x[j];
                memcpy(h_all + w, out_x + offs_x[j], c * sizeof(uint64_t));
                w += c;
                ++j;
            } else {
                const uint64_t c = h_off[a + 1] - h_off[a];
                memcpy(h_all + w, h_out + h_off[a], c * sizeof(uint64_t));
                w += c;
            }
        }
        offsets[nq] = w;
        free(h_out);
        free(out_x);
        *idx_out = h_all;
        {
            std::lock_guard<std::mutex> lk(ix->sh.mu);
            ix->sh.stats.fallback_queries += nf;
        }
        *done = true;
        return PN_OK;
    }
    memcpy(offsets, h_off.data(), (nq + 1) * sizeof(uint64_t));
    *idx_out = h_out;
    *done = true;
    return PN_OK;
}

// first tier for radius queries: the bf16 filter against each query's fixed bound, exact check of the survivors
// (f32 and f64 indexes: the filter bounds real squared distances; the survivors' check and the threshold's rounding
// allowance follow the element type -- u = 2^-24 or 2^-53)
// (enqueue only: the filter against each query's fixed bound and the exact check of the survivors.  Leaves, per query, the
// kept rows in w_keys [nq][*kept_stride] (ascending), their number in w_flags, an overflow flag in w_sel, w_qbad for
// queries the filter cannot serve, d_misc[0] / [1] = how many of either.  *enq = false: the radius cannot be served.)
template <typename T>
static int radius_bf16_enqueue(const pn_index *ix, Workspace &ws, int level, const T *Qp, size_t nq, size_t nq_pad, T radius,
                               size_t *kept_stride_out, bool *enq, hipStream_t s, const T *qnorm = nullptr) {
    *enq = false;
    const bool cosine = ix->metric == 1;  // (round 4: the filter over the normalised rows, Cosine::distance check)
    if (cosine && (!qnorm || !ix->d_cnorm || !(radius < (T)1))) return PN_OK;  // (r >= 1: half the sphere -- exact scan)
    const int cap = 256;  // up to 224 rows within the radius per (segment, query) before the call overflows
    const size_t q_tiles = nq_pad / 256, r_tiles = (ix->n + 63) / 64;
    size_t n_wg = (size_t)ix->n_cu * 2;
    size_t cap_wg = q_tiles * 32;
    const size_t by_work = (q_tiles * r_tiles + 31) / 32;
    if (by_work < cap_wg) cap_wg = by_work;
    if (cap_wg < 1) cap_wg = 1;
    if (n_wg > cap_wg) n_wg = cap_wg;
    n_wg = bf16_grid_wgs(ix, q_tiles, r_tiles, n_wg);
    const bool wide = bf16_is_wide((int)ix->dim);
    int wide_wg = 1;
    if (wide) {  // the k-NN plan's partition, at most 32 segments per query (the check kernel's LDS budget)
        wide_wg = bf16_plan_wide(ix, nq_pad, 1, level).n_wg;
        while (wide_wg > 1 && bf16_wide_segments(q_tiles, wide_wg) > 32) wide_wg = (wide_wg + 1) / 2;
    }
    const int nseg = wide ? bf16_wide_segments(q_tiles, wide_wg) : bf16_segments(q_tiles, (int)n_wg, 1);
    const size_t cells = (size_t)nseg * nq_pad;
    const size_t kept_stride = (size_t)nseg * cap;
    // tau_r = (r^2 + 1e-37) / (1 - (D+4) 2^-24): every row whose reference distance is < r has a squared distance
    // below it (same allowances as the k-NN proof)
    // Cosine: a row whose reference distance is < r has |q~ - p~|^2 < 2 (r + E' + 1e-14) + eps1 -- cos_proof_lb
    // (select.hip) read backwards, with its (|x| + 1) 1e-15 term bounded for |x| <= 9
    const double ut = sizeof(T) == 4 ? 5.9604644775390625e-08 : 1.1102230246251565e-16;
    const double t = cosine
                         ? 2.0 * ((double)radius + (2.0 * (double)ix->dim + 16.0) * ut + 1.0e-14) +
                               (4.5 * (double)ix->dim + 45.0) * 1.1102230246251565e-16
                     : sizeof(T) == 4
                         ? ((double)radius * (double)radius + 1e-37) / (1.0 - (double)(ix->dim + 4) * 5.9604644775390625e-08)
                         : ((double)radius * (double)radius * (1.0 + 8.881784197001252e-16) + 1e-300) /
                               (1.0 - (double)(ix->dim + 4) * 1.1102230246251565e-16);  // (f64: r^2 itself is rounded)
    if (!(t < 1e37)) return PN_OK;  // let the exact engine decide
    PNCHK(ws.w_bq.ensure(bf16_query_bytes(nq_pad, (int)ix->dim, ix->bf16_ci)));
    PNCHK(ws.w_qn.ensure(nq_pad * sizeof(double)));
    PNCHK(ws.w_qbad.ensure(nq_pad * sizeof(uint32_t)));
    PNCHK(ws.w_seed.ensure(nq_pad * sizeof(uint32_t)));
    PNCHK(ws.w_misc.ensure(64));
    PNCHK(ws.w_cnt.ensure(cells * sizeof(uint32_t)));
    PNCHK(ws.w_tau.ensure(cells * sizeof(uint32_t)));
    PNCHK(ws.w_idx.ensure(cells * cap * 2 * sizeof(uint32_t)));      // (key, row) pairs
    PNCHK(ws.w_keys.ensure(nq * kept_stride * sizeof(uint32_t)));    // kept rows per query
    PNCHK(ws.w_flags.ensure(nq_pad * sizeof(uint32_t)));             // kept counts
    PNCHK(ws.w_sel.ensure(nq_pad * sizeof(uint32_t)));            // per-query overflow flags
    uint32_t *d_misc = (uint32_t *)ws.w_misc.p;  // [0] overflow count, [1] queries the filter cannot serve
    HIPCHK(hipMemsetAsync(ws.w_misc.p, 0, 64, s));
    HIPCHK(hipMemsetAsync(ws.w_cnt.p, 0, cells * sizeof(uint32_t), s));
    if (cosine) {  // q~ = q / |q| in f64 (a query without a direction becomes NaNs: flagged by the pack kernel, exact scan)
        PNCHK(ws.w_qnrm.ensure(nq_pad * ix->ld * sizeof(double)));
        HIPCHK(launch_cos_normalize_rows<T>(Qp, nq, nq_pad, (int)ix->dim, ix->ld, (double *)ws.w_qnrm.p, ix->ld, nullptr,
                                            true, s));
        HIPCHK(launch_bf16_pack_queries<double>((const double *)ws.w_qnrm.p, ix->d_mu, nq, nq_pad, (int)ix->dim, ix->ld,
                                                ws.w_bq.p, (double *)ws.w_qn.p, (uint32_t *)ws.w_qbad.p, ix->bf16_ci,
                                                ix->bf16_bmax, ix->bf16_dmax, s));
    } else
        HIPCHK(launch_bf16_pack_queries(Qp, ix->d_mu, nq, nq_pad, (int)ix->dim, ix->ld, ws.w_bq.p, (double *)ws.w_qn.p,
                                        (uint32_t *)ws.w_qbad.p, ix->bf16_ci, ix->bf16_bmax, ix->bf16_dmax, s));
    PNCHK(ws.w_gsel.ensure(nq_pad * sizeof(uint32_t)));
    HIPCHK(launch_compact_flags((const uint32_t *)ws.w_qbad.p, (int)nq, (uint32_t *)ws.w_gsel.p, d_misc + 1, s));
    HIPCHK(launch_bf16_radius_tau((const double *)ws.w_qn.p, nq_pad, t, (uint32_t *)ws.w_seed.p, s));
    CandBuf cb{ws.w_idx.p, (uint32_t *)ws.w_idx.p + 1, (uint32_t *)ws.w_cnt.p, ws.w_tau.p, nq_pad, nseg, cap, 2};
    // PN_OPT_PROFILE: the filter launch between two events of the workspace (the radius entry points wait for their
    // stream anyway: the bracket is resolved in radius_finish's shadow, below)
    const bool prof = ix->profile != 0;
    if (prof) {
        if (!ws.ev_r[0]) {
            HIPCHK(hipEventCreate(&ws.ev_r[0]));
            HIPCHK(hipEventCreate(&ws.ev_r[1]));
        }
        HIPCHK(hipEventRecord(ws.ev_r[0], s));
    }
    if (wide)
        HIPCHK(launch_bf16_wide_filter(ix->d_img, ix->n, (int)ix->dim, ws.w_bq.p, cap - 32, cb, wide_wg, 0,
                                       (const uint32_t *)ws.w_seed.p, true, nullptr, s));
    else
        HIPCHK(launch_bf16_filter(ix->d_img, ix->n, (int)ix->dim, ws.w_bq.p, cap - 32, cb, (int)n_wg, 1, 0,
                                  (const uint32_t *)ws.w_seed.p, true, nullptr, ix->bf16_ci, s));
    if (prof) HIPCHK(hipEventRecord(ws.ev_r[1], s));
    HIPCHK(launch_radius_check<T>((const uint32_t *)ws.w_cnt.p, (const uint32_t *)ws.w_idx.p + 1, nq_pad, nseg, cap,
                                  (const T *)ix->d_pts, ix->ld, Qp, (int)nq, (int)ix->dim, radius,
                                  (uint32_t *)ws.w_keys.p, (uint32_t *)ws.w_flags.p, d_misc, 2,
                                  (uint32_t *)ws.w_sel.p, s, cosine ? (const T *)ix->d_cnorm : nullptr,
                                  cosine ? qnorm : nullptr));
    *kept_stride_out = kept_stride;
    *enq = true;
    return PN_OK;
}
template <typename T>
static int radius_bf16(const pn_index *ix, Workspace &ws, int level, const T *Qp, size_t nq, size_t nq_pad, T radius, uint64_t *offsets,
                       uint64_t **idx_out, bool *done, hipStream_t s, const T *qnorm = nullptr) {
    *done = false;
    size_t kept_stride = 0;
    bool enq = false;
    PNCHK(radius_bf16_enqueue<T>(ix, ws, level, Qp, nq, nq_pad, radius, &kept_stride, &enq, s, qnorm));
    if (!enq) return PN_OK;
    const bool prof = ix->profile != 0;
    uint32_t *d_misc = (uint32_t *)ws.w_misc.p;
    const int rc = radius_finish<T>(ix, ws, Qp, nq, kept_stride, d_misc, (const uint32_t *)ws.w_sel.p, radius, offsets,
                                    idx_out, done, s, ix->metric == 1);
    if (prof && rc == PN_OK) {  // (radius_finish has waited for the stream)
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, ws.ev_r[0], ws.ev_r[1]) == hipSuccess) {
            std::lock_guard<std::mutex> lk(ix->sh.mu);
            ix->sh.stats.hot_ms += ms;
            ix->sh.stats.hot_launches += 1;
        }
    }
    return rc;
}

static int radius_mfma(const pn_index *ix, Workspace &ws, const float *Qp, size_t nq, size_t nq_pad, float radius, uint64_t *offsets,
                       uint64_t **idx_out, bool *done, hipStream_t s) {
    *done = false;
    const uint32_t cap = 32;
    const size_t q_tiles = nq_pad / 128, r_tiles = (ix->n + 63) / 64;
    size_t n_wg = (size_t)ix->n_cu * 2;
    size_t cap_wg = q_tiles * 32;
    const size_t by_work = (q_tiles * r_tiles + 31) / 32;
    if (by_work < cap_wg) cap_wg = by_work;
    if (cap_wg < 1) cap_wg = 1;
    if (n_wg > cap_wg) n_wg = cap_wg;
    const int nseg = mfma_v2_max_segments(q_tiles, (int)n_wg);
    const size_t cells = (size_t)nseg * nq_pad;
    const size_t kept_stride = (size_t)nseg * cap;
    // tau_r = (r^2 + 1e-37) / (1 - (D+4) 2^-24), rounded up; the kernel tests L < succ(tau_r)
    const double t = ((double)radius * (double)radius + 1e-37) / (1.0 - (double)(ix->dim + 4) * 5.9604644775390625e-08);
    float tf = (float)t;
    if ((double)tf < t) tf = nextafterf(tf, INFINITY);
    if (!(tf < INFINITY)) return PN_OK;  // r^2 overflows f32: let the exact engine decide
    const float tau_excl = nextafterf(tf, INFINITY);

    PNCHK(ws.w_qnorm.ensure(nq_pad * sizeof(float)));
    PNCHK(ws.w_misc.ensure(64));
    PNCHK(ws.w_cnt.ensure(cells * sizeof(uint32_t)));
    PNCHK(ws.w_idx.ensure(cells * cap * sizeof(uint32_t)));
    PNCHK(ws.w_keys.ensure(nq * kept_stride * sizeof(uint32_t)));  // kept rows per query
    PNCHK(ws.w_flags.ensure(nq_pad * sizeof(uint32_t)));           // kept counts
    uint32_t *d_misc = (uint32_t *)ws.w_misc.p;  // [0] overflow count, [1] non-finite query norms
    HIPCHK(hipMemsetAsync(ws.w_misc.p, 0, 64, s));
    HIPCHK(hipMemsetAsync(ws.w_cnt.p, 0, cells * sizeof(uint32_t), s));
    HIPCHK(launch_row_norms_f32(Qp, nq_pad, nq, (int)ix->dim, ix->ld, mfma_alpha(ix->dim), (float *)ws.w_qnorm.p,
                                d_misc + 1, s));
    HIPCHK(launch_mfma_radius_f32((const float *)ix->d_pts, ix->d_norm, ix->n, ix->ld, Qp, (const float *)ws.w_qnorm.p,
                                  nq_pad, tau_excl, cap, (uint32_t *)ws.w_cnt.p, (uint32_t *)ws.w_idx.p, (int)n_wg, s));
    HIPCHK(launch_radius_check_f32((const uint32_t *)ws.w_cnt.p, (const uint32_t *)ws.w_idx.p, nq_pad, nseg, cap,
                                   (const float *)ix->d_pts, ix->ld, Qp, (int)nq, (int)ix->dim, radius,
                                   (uint32_t *)ws.w_keys.p, (uint32_t *)ws.w_flags.p, d_misc, 1, nullptr, s));
    return radius_finish<float>(ix, ws, Qp, nq, kept_stride, d_misc, nullptr, radius, offsets, idx_out, done, s);
}

template <typename T>
static int radius_host_impl(const pn_index *ix, const T *q, size_t nq, size_t q_cols, ptrdiff_t q_stride, T radius,
                            uint64_t *offsets, uint64_t **idx_out) {
    if (!ix) return fail(PN_ERR_INVALID, "index is NULL");
    if (ix->elem_bytes != (int)sizeof(T)) return fail(PN_ERR_INVALID, "index element type mismatch");
    if (!offsets || !idx_out) return fail(PN_ERR_INVALID, "output pointer is NULL");
    *idx_out = nullptr;
    offsets[0] = 0;
    if (nq == 0) return PN_OK;
    if (!q && q_cols) return fail(PN_ERR_INVALID, "queries is NULL");
    if (nq > 0x7FFFFFFFull) return fail(PN_ERR_UNSUPPORTED, "too many queries in one call");
    DeviceGuard g(ix->device);
    if (!g.ok) return fail(PN_ERR_DEVICE, "hipSetDevice(%d) failed", ix->device);
    WsLease lease(ix);
    PNCHK(ws_acquire(ix, &lease.s, true, &lease.ws));
    Workspace &ws = *lease.ws;
    hipStream_t s = lease.s;
    int level;
    {
        std::lock_guard<std::mutex> lk(ix->sh.mu);
        recs_collect(ix, false);
        level = ix->sh.bf16_level;
    }
    const size_t dim_eff = q_cols < ix->dim ? q_cols : ix->dim;
    if (q_stride >= 0 && tiny_eligible<T>(ix, nq, dim_eff))
        return tiny_radius<T>(ix, ws, s, q, nq, q_cols, q_stride, radius, offsets, idx_out);
    const size_t nq_pad = round_up(nq, (size_t)256);
    int rc = upload_rows_to<T>(q, nq, q_cols, q_stride, ws.w_hq, s);
    do {
        if (rc != PN_OK) break;
        rc = ws.w_q.ensure(nq_pad * ix->ld * sizeof(T));
        if (rc != PN_OK) break;
        T *Qp = (T *)ws.w_q.p;
        if (Ops<T>::pack((const T *)ws.w_hq.p, nq, dim_eff, q_cols ? q_cols : 1, Qp, nq_pad, ix->ld, s) != hipSuccess) {
            rc = fail(PN_ERR_DEVICE, "pack failed");
            break;
        }
        const bool finite_pos = radius > (T)0 && radius < (T)INFINITY;
        const T *qnorm = nullptr;
        if (ix->metric == 1) {  // Cosine: the queries' norms over their own length
            rc = ws.w_qnorm.ensure(nq_pad * sizeof(T));
            if (rc != PN_OK) break;
            if (hipMemsetAsync(ws.w_qnorm.p, 0, nq_pad * sizeof(T), s) != hipSuccess ||
                Ops<T>::cnorms((const T *)ws.w_hq.p, nq, (int)q_cols, q_cols ? q_cols : 1, (T *)ws.w_qnorm.p, s) != hipSuccess) {
                rc = fail(PN_ERR_DEVICE, "query norms failed");
                break;
            }
            qnorm = (const T *)ws.w_qnorm.p;
        }
        // (Euclidean indexes, and -- round 4 -- Cosine indexes for queries as long as the rows and r < 1: the filter over the
        // normalised rows + the Cosine::distance check of its survivors; everything else: the exact two-pass scan)
        if ((ix->metric == 0 || q_cols == ix->dim) && ix->bf16_ok && dim_eff == ix->dim && finite_pos && level < 2 &&
            (ix->engine == PN_ENGINE_BF16 || (ix->engine == PN_ENGINE_AUTO && ix->n >= 4096 && ix->dim >= 8))) {
            bool done = false;  // (f32 and f64 indexes)
            rc = radius_bf16<T>(ix, ws, level, (const T *)Qp, nq, nq_pad, radius, offsets, idx_out, &done, s, qnorm);
            if (rc != PN_OK || done) break;
        }
        if constexpr (sizeof(T) == 4) {
            if (ix->mfma_ok && ix->ld <= 128 && dim_eff == ix->dim && ix->engine != PN_ENGINE_EXACT && finite_pos &&
                (ix->engine == PN_ENGINE_MFMA || (ix->n >= 4096 && ix->dim >= 8))) {
                bool done = false;
                rc = radius_mfma(ix, ws, (const float *)Qp, nq, nq_pad, (float)radius, offsets, idx_out, &done, s);
                if (rc != PN_OK || done) break;
                std::lock_guard<std::mutex> lk(ix->sh.mu);
                ix->sh.stats.fallback_queries += nq;  // survivor list overflow / non-finite query: exact engine
            }
        }
        std::vector<uint64_t> offs;
        rc = radius_exact<T>(ix, ws, Qp, nq, nq_pad, dim_eff, radius, offs, idx_out, s, qnorm);
        if (rc != PN_OK) break;
        memcpy(offsets, offs.data(), (nq + 1) * sizeof(uint64_t));
    } while (0);
    if (rc != PN_OK && *idx_out) { free(*idx_out); *idx_out = nullptr; }
    return rc;
}

extern "C" int pn_query_radius_f32(const pn_index *ix, const float *q, size_t nq, size_t q_cols, ptrdiff_t q_stride,
                                   float radius, uint64_t *offsets, uint64_t **idx_out) {
    return radius_host_impl<float>(ix, q, nq, q_cols, q_stride, radius, offsets, idx_out);
}
extern "C" int pn_query_radius_f64(const pn_index *ix, const double *q, size_t nq, size_t q_cols, ptrdiff_t q_stride,
                                   double radius, uint64_t *offsets, uint64_t **idx_out) {
    return radius_host_impl<double>(ix, q, nq, q_cols, q_stride, radius, offsets, idx_out);
}
// ---------------------------------------------------------------------------
// pn_query_radius_device_{f32,f64} (round 4; VERDICT r3 missing 2): BallTree::query_radius (src/ball_tree.rs:137-142,
// 250-294) for queries resident in HBM, everything enqueued on the caller's stream, NOTHING read back: the host entry
// points read counts back twice per call to size their output.  The caller supplies the capacity of d_idx; the
// per-query counts are scanned on the device into d_offsets [nq + 1]; rows are written at their CSR positions while
// those lie below the capacity; d_total[0] = d_offsets[nq] tells the caller (whenever it chooses to look) whether the
// buffer was large enough -- if not, every offset is still right and the first `capacity` entries are in place.
//   Euclidean indexes with a bf16 tier: the filter against each query's fixed bound + the exact check of the survivors
//   (as the host entry point); the queries it cannot serve (survivor list overflow, non-finite norms) are LISTED on the
//   device and answered by the exact two-pass scan through that list -- counts, offsets and fill all device-driven.
//   Cosine indexes with a bf16 tier, queries as long as the rows, r < 1: the same over the normalised rows, the survivors
//   checked with Cosine::distance.  Every other case (small corpora, no tier): the exact two-pass scan for all queries.
// ---------------------------------------------------------------------------
template <typename T>
static int radius_device_impl(const pn_index *ix, const T *d_q, size_t nq, size_t q_cols, size_t q_stride, T radius,
                              uint64_t *d_offsets, uint64_t *d_idx, size_t capacity, uint64_t *d_total, hipStream_t s) {
    if (!ix) return fail(PN_ERR_INVALID, "index is NULL");
    if (ix->elem_bytes != (int)sizeof(T)) return fail(PN_ERR_INVALID, "index element type mismatch");
    if (!d_offsets || (!d_idx && capacity)) return fail(PN_ERR_INVALID, "output buffer is NULL");
    if (!d_q && q_cols && nq) return fail(PN_ERR_INVALID, "queries is NULL");
    if (nq > 0x7FFFFFFFull) return fail(PN_ERR_UNSUPPORTED, "too many queries in one call");
    DeviceGuard g(ix->device);
    if (!g.ok) return fail(PN_ERR_DEVICE, "hipSetDevice(%d) failed", ix->device);
    if (nq == 0) {
        HIPCHK(hipMemsetAsync(d_offsets, 0, sizeof(uint64_t), s));
        if (d_total) HIPCHK(hipMemsetAsync(d_total, 0, sizeof(uint64_t), s));
        return PN_OK;
    }
    WsLease lease(ix);
    lease.s = s;
    PNCHK(ws_acquire(ix, &lease.s, false, &lease.ws));
    Workspace &ws = *lease.ws;
    int level;
    {
        std::lock_guard<std::mutex> lk(ix->sh.mu);
        recs_collect(ix, false);
        level = ix->sh.bf16_level;
        ix->sh.stats.queries += nq;
    }
    const size_t dim_eff = q_cols < ix->dim ? q_cols : ix->dim;
    const size_t nq_pad = round_up(nq, (size_t)256);
    PNCHK(ws.w_q.ensure(nq_pad * ix->ld * sizeof(T)));
    T *Qp = (T *)ws.w_q.p;
    HIPCHK(Ops<T>::pack(d_q, nq, dim_eff, q_stride, Qp, nq_pad, ix->ld, s));
    const T *qnorm = nullptr;
    if (ix->metric == 1) {  // Cosine: the queries' norms over their own length
        PNCHK(ws.w_qnorm.ensure(nq_pad * sizeof(T)));
        HIPCHK(hipMemsetAsync(ws.w_qnorm.p, 0, nq_pad * sizeof(T), s));
        HIPCHK(Ops<T>::cnorms(d_q, nq, (int)q_cols, q_stride, (T *)ws.w_qnorm.p, s));
        qnorm = (const T *)ws.w_qnorm.p;
    }
    // ---- first tier (enqueue only)
    bool filtered = false;
    size_t kept_stride = 0;
    const bool finite_pos = radius > (T)0 && radius < (T)INFINITY;
    if ((ix->metric == 0 || q_cols == ix->dim) && ix->bf16_ok && dim_eff == ix->dim && finite_pos && level < 2 &&
        (ix->engine == PN_ENGINE_BF16 || (ix->engine == PN_ENGINE_AUTO && ix->n >= 4096 && ix->dim >= 8)))
        PNCHK(radius_bf16_enqueue<T>(ix, ws, level, (const T *)Qp, nq, nq_pad, radius, &kept_stride, &filtered, s, qnorm));
    PNCHK(ws.w_misc.ensure(64));
    uint32_t *d_misc = (uint32_t *)ws.w_misc.p;  // [4]: the number of listed queries
    const uint32_t *d_sel = nullptr, *d_pos = nullptr, *d_nsel = nullptr, *d_nkept = nullptr;
    if (filtered) {
        PNCHK(ws.w_gsel.ensure(nq_pad * sizeof(uint32_t)));
        PNCHK(ws.w_rpos.ensure(nq_pad * sizeof(uint32_t)));
        HIPCHK(hipMemsetAsync(d_misc + 4, 0, sizeof(uint32_t), s));
        HIPCHK(launch_rad_list((const uint32_t *)ws.w_sel.p, (const uint32_t *)ws.w_qbad.p, (int)nq, (uint32_t *)ws.w_flags.p,
                               (uint32_t *)ws.w_gsel.p, (uint32_t *)ws.w_rpos.p, d_misc + 4, s));
        d_sel = (const uint32_t *)ws.w_gsel.p;
        d_pos = (const uint32_t *)ws.w_rpos.p;
        d_nsel = d_misc + 4;
        d_nkept = (const uint32_t *)ws.w_flags.p;
    }
    // ---- exact two-pass scan for the listed queries (all of them without a first tier), driven by the device-side list
    const size_t q_tiles_r = round_up(nq, (size_t)kTileQ) / kTileQ;
    const ScanPlan pl = plan_segments(ix->n, q_tiles_r, 1, ix->opt_segments, 4096, q_tiles_r <= 8 ? 1024 : 64);
    const size_t cells = nq * (size_t)pl.nseg;
    PNCHK(ws.w_rcx.ensure(cells * sizeof(uint32_t)));
    PNCHK(ws.w_rox.ensure(cells * sizeof(uint64_t)));
    PNCHK(ws.w_rfin.ensure(nq_pad * sizeof(uint32_t)));
    PNCHK(ws.w_rscan.ensure((nq / 4096 + 2) * sizeof(uint64_t)));
    const T *pn = qnorm ? (const T *)ix->d_cnorm : nullptr;
    HIPCHK(RadOps<T>::run((const T *)ix->d_pts, ix->n, (int)dim_eff, ix->ld, Qp, (int)nq, ix->ld, radius, pl.seg_len, pl.nseg,
                          (uint32_t *)ws.w_rcx.p, nullptr, nullptr, ix->index_base, pn, qnorm, s, d_sel, d_nsel, ~0ull));
    HIPCHK(launch_rad_counts(d_nkept, d_pos, (const uint32_t *)ws.w_rcx.p, pl.nseg, (int)nq, (uint32_t *)ws.w_rfin.p, s));
    HIPCHK(launch_exclusive_scan_u32((const uint32_t *)ws.w_rfin.p, nq, d_offsets, (uint64_t *)ws.w_rscan.p, d_total, s));
    HIPCHK(launch_rad_seg_offsets(d_offsets, d_sel, d_nsel, (int)nq, (const uint32_t *)ws.w_rcx.p, pl.nseg,
                                  (uint64_t *)ws.w_rox.p, s));
    // (a fill pass needs a buffer to fill: with capacity 0 the call is a pure count)
    if (capacity) {
        HIPCHK(RadOps<T>::run((const T *)ix->d_pts, ix->n, (int)dim_eff, ix->ld, Qp, (int)nq, ix->ld, radius, pl.seg_len,
                              pl.nseg, (uint32_t *)ws.w_rcx.p, (const uint64_t *)ws.w_rox.p, d_idx, ix->index_base, pn, qnorm,
                              s, d_sel, d_nsel, (uint64_t)capacity));
        if (filtered)
            HIPCHK(launch_radius_gather_cap((const uint32_t *)ws.w_keys.p, d_nkept, d_offsets, (int)nq, kept_stride,
                                            ix->index_base, d_idx, (uint64_t)capacity, s));
    }
    return PN_OK;
}
extern "C" int pn_query_radius_device_f32(const pn_index *ix, const float *d_q, size_t nq, size_t q_cols, size_t q_stride,
                                          float radius, uint64_t *d_offsets, uint64_t *d_idx, size_t capacity,
                                          uint64_t *d_total, void *stream) {
    return radius_device_impl<float>(ix, d_q, nq, q_cols, q_stride, radius, d_offsets, d_idx, capacity, d_total,
                                     (hipStream_t)stream);
}
extern "C" int pn_query_radius_device_f64(const pn_index *ix, const double *d_q, size_t nq, size_t q_cols, size_t q_stride,
                                          double radius, uint64_t *d_offsets, uint64_t *d_idx, size_t capacity,
                                          uint64_t *d_total, void *stream) {
    return radius_device_impl<double>(ix, d_q, nq, q_cols, q_stride, radius, d_offsets, d_idx, capacity, d_total,
                                      (hipStream_t)stream);
}

extern "C" void pn_free(void *p) { free(p); }

// ---------------------------------------------------------------------------
// pairwise
// ---------------------------------------------------------------------------
// distance::pairwise with input and output in HBM (extension: the reference returns a host Array2): d_x row-major
// [n][cols] with row stride >= cols, d_out [n][n]; enqueued on `stream`, nothing is copied or waited for except the
// padded working copy of the rows, which is released when the stream has passed it.
template <typename T>
static int pairwise_device_impl(const T *d_x, size_t n, size_t cols, size_t row_stride, int device, T *d_out,
                                hipStream_t s) {
    if (n == 0) return PN_OK;
    if (!d_out) return fail(PN_ERR_INVALID, "out is NULL");
    if (!d_x && cols) return fail(PN_ERR_INVALID, "x is NULL");
    PNCHK(check_device(device));
    DeviceGuard g(device);
    if (!g.ok) return fail(PN_ERR_DEVICE, "hipSetDevice(%d) failed", device);
    if (n < 2) {  // src/distance.rs:63-65
        HIPCHK(hipMemsetAsync(d_out, 0, n * n * sizeof(T), s));
        return PN_OK;
    }
    const size_t ld = pick_ld(cols), n_pad = round_up(n, (size_t)kRowPad);
    T *d_p = nullptr;
    HIPCHK(hipMalloc((void **)&d_p, n_pad * ld * sizeof(T)));
    hipError_t e = Ops<T>::pack(d_x, n, cols, row_stride ? row_stride : 1, d_p, n_pad, ld, s);
    if (e == hipSuccess)
        e = (sizeof(T) == 4) ? launch_exact_pairwise_f32((const float *)d_p, n, (int)cols, ld, (float *)d_out, s)
                             : launch_exact_pairwise_f64((const double *)d_p, n, (int)cols, ld, (double *)d_out, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);  // the working copy must outlive the kernel
    (void)hipFree(d_p);
    if (e != hipSuccess) return fail(PN_ERR_DEVICE, "pairwise: %s", hipGetErrorString(e));
    return PN_OK;
}
extern "C" int pn_pairwise_device_f32(const float *d_x, size_t n, size_t cols, size_t row_stride, int device,
                                      float *d_out, void *stream) {
    return pairwise_device_impl<float>(d_x, n, cols, row_stride, device, d_out, (hipStream_t)stream);
}
extern "C" int pn_pairwise_device_f64(const double *d_x, size_t n, size_t cols, size_t row_stride, int device,
                                      double *d_out, void *stream) {
    return pairwise_device_impl<double>(d_x, n, cols, row_stride, device, d_out, (hipStream_t)stream);
}

template <typename T>
static int pairwise_impl(const T *x, size_t n, size_t cols, ptrdiff_t row_stride, int device, T *out,
                         bool cosine = false) {
    if (!out && n) return fail(PN_ERR_INVALID, "out is NULL");
    if (n == 0) return PN_OK;
    if (n < 2) {  // src/distance.rs:63-65
        memset(out, 0, n * n * sizeof(T));
        return PN_OK;
    }
    if (!x && cols) return fail(PN_ERR_INVALID, "x is NULL");
    PNCHK(check_device(device));
    DeviceGuard g(device);
    if (!g.ok) return fail(PN_ERR_DEVICE, "hipSetDevice(%d) failed", device);
    const size_t ld = pick_ld(cols), n_pad = round_up(n, (size_t)kRowPad);
    if (cosine && n > 0x7FFFFFFFull) return fail(PN_ERR_UNSUPPORTED, "too many rows");
    T *d_x = nullptr, *d_p = nullptr, *d_o = nullptr, *d_n = nullptr;
    int rc = upload_rows<T>(x, n, cols, row_stride, &d_x);
    do {
        if (rc != PN_OK) break;
        if (hipMalloc((void **)&d_p, n_pad * ld * sizeof(T)) != hipSuccess ||
            hipMalloc((void **)&d_o, n * n * sizeof(T)) != hipSuccess ||
            (cosine && hipMalloc((void **)&d_n, n * sizeof(T)) != hipSuccess)) {
            rc = fail(PN_ERR_NOMEM, "hipMalloc pairwise failed");
            break;
        }
        hipError_t e = Ops<T>::pack(d_x, n, cols, cols ? cols : 1, d_p, n_pad, ld, nullptr);
        if (e == hipSuccess && cosine)
            e = (sizeof(T) == 4) ? launch_cosine_pairwise_f32((const float *)d_p, n, (int)cols, ld, (float *)d_n,
                                                              (float *)d_o, nullptr)
                                 : launch_cosine_pairwise_f64((const double *)d_p, n, (int)cols, ld, (double *)d_n,
                                                              (double *)d_o, nullptr);
        else if (e == hipSuccess)
            e = (sizeof(T) == 4) ? launch_exact_pairwise_f32((const float *)d_p, n, (int)cols, ld, (float *)d_o, nullptr)
                                 : launch_exact_pairwise_f64((const double *)d_p, n, (int)cols, ld, (double *)d_o, nullptr);
        if (e == hipSuccess) e = hipStreamSynchronize(nullptr);  // (the kernels above ran on the default stream)
        if (e == hipSuccess) e = hipMemcpy(out, d_o, n * n * sizeof(T), hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail(PN_ERR_DEVICE, "pairwise: %s", hipGetErrorString(e));
    } while (0);
    if (d_x) (void)hipFree(d_x);
    if (d_p) (void)hipFree(d_p);
    if (d_o) (void)hipFree(d_o);
    if (d_n) (void)hipFree(d_n);
    return rc;
}
extern "C" int pn_pairwise_cosine_f32(const float *x, size_t n, size_t cols, ptrdiff_t row_stride, int device,
                                      float *out) {
    return pairwise_impl<float>(x, n, cols, row_stride, device, out, true);
}
extern "C" int pn_pairwise_cosine_f64(const double *x, size_t n, size_t cols, ptrdiff_t row_stride, int device,
                                      double *out) {
    return pairwise_impl<double>(x, n, cols, row_stride, device, out, true);
}
extern "C" int pn_pairwise_f32(const float *x, size_t n, size_t cols, ptrdiff_t row_stride, int device, float *out) {
    return pairwise_impl<float>(x, n, cols, row_stride, device, out);
}
extern "C" int pn_pairwise_f64(const double *x, size_t n, size_t cols, ptrdiff_t row_stride, int device, double *out) {
    return pairwise_impl<double>(x, n, cols, row_stride, device, out);
}

// ---------------------------------------------------------------------------
// tree introspection (src/ball_tree.rs:296-353): a host-side ball tree identical to the reference's, built lazily
// ---------------------------------------------------------------------------
static int tree_of(const pn_index *ix, const HostTree **out) {
    if (!ix) return fail(PN_ERR_INVALID, "index is NULL");
    HostTree *t = ix->sh.tree.load(std::memory_order_acquire);
    if (!t) {
        std::lock_guard<std::mutex> lk(ix->sh.tree_mu);  // builders only: queries never take this mutex
        t = ix->sh.tree.load(std::memory_order_acquire);
        if (t) {
            *out = t;
            return PN_OK;
        }
        DeviceGuard g(ix->device);
        if (!g.ok) return fail(PN_ERR_DEVICE, "hipSetDevice(%d) failed", ix->device);
        const size_t eb = (size_t)ix->elem_bytes, cols = ix->dim ? ix->dim : 1;
        std::vector<unsigned char> host;
        try {
            host.resize(ix->n * cols * eb);
        } catch (const std::bad_alloc &) {
            return fail(PN_ERR_NOMEM, "host allocation of %zu rows failed", ix->n);
        }
        if (ix->dim)  // the index's own zero-padded copy of the points, rows back to their unpadded length
            HIPCHK(hipMemcpy2D(host.data(), ix->dim * eb, ix->d_pts, ix->ld * eb, ix->dim * eb, ix->n, hipMemcpyDeviceToHost));
        // BallTree::new(points, metric): the tree's radii and node bounds are the index's metric's (Cosine indexes
        // included -- src/ball_tree.rs:445-461 passes `metric` to Node::init)
        t = host_tree_build(host.data(), ix->n, ix->dim, ix->elem_bytes, ix->metric);
        if (!t) return fail(PN_ERR_NOMEM, "ball tree of %zu points: out of memory", ix->n);
        ix->sh.tree.store(t, std::memory_order_release);
    }
    *out = t;
    return PN_OK;
}
static int tree_node(const pn_index *ix, uint64_t node, const HostTree **t) {
    PNCHK(tree_of(ix, t));
    if (node >= host_tree_num_nodes(*t))  // the reference panics (slice index out of bounds)
        return fail(PN_ERR_INVALID, "node %llu out of range (the tree has %zu nodes)", (unsigned long long)node,
                    host_tree_num_nodes(*t));
    return PN_OK;
}
extern "C" int pn_tree_num_nodes(const pn_index *ix, uint64_t *out) {
    if (!out) return fail(PN_ERR_INVALID, "out is NULL");
    const HostTree *t = nullptr;
    PNCHK(tree_of(ix, &t));
    *out = host_tree_num_nodes(t);
    return PN_OK;
}
extern "C" int pn_tree_children_of(const pn_index *ix, uint64_t node, int *is_some, uint64_t *left, uint64_t *right) {
    if (!is_some || !left || !right) return fail(PN_ERR_INVALID, "NULL argument");
    const HostTree *t = nullptr;
    PNCHK(tree_node(ix, node, &t));
    uint64_t s, e;
    int leaf;
    host_tree_node(t, node, &s, &e, &leaf);
    *is_some = leaf ? 0 : 1;
    *left = 2 * node + 1;
    *right = 2 * node + 2;
    return PN_OK;
}
extern "C" int pn_tree_points_of(const pn_index *ix, uint64_t node, const uint64_t **idx, uint64_t *count) {
    if (!idx || !count) return fail(PN_ERR_INVALID, "NULL argument");
    const HostTree *t = nullptr;
    PNCHK(tree_node(ix, node, &t));
    uint64_t s, e;
    int leaf;
    host_tree_node(t, node, &s, &e, &leaf);
    *idx = host_tree_idx(t) + s;
    *count = e - s;
    return PN_OK;
}
extern "C" int pn_tree_radius_of_f32(const pn_index *ix, uint64_t node, float *out) {
    if (!out) return fail(PN_ERR_INVALID, "out is NULL");
    if (ix && ix->elem_bytes != 4) return fail(PN_ERR_INVALID, "index element type mismatch");
    const HostTree *t = nullptr;
    PNCHK(tree_node(ix, node, &t));
    *out = (float)host_tree_radius(t, node);
    return PN_OK;
}
extern "C" int pn_tree_radius_of_f64(const pn_index *ix, uint64_t node, double *out) {
    if (!out) return fail(PN_ERR_INVALID, "out is NULL");
    if (ix && ix->elem_bytes != 8) return fail(PN_ERR_INVALID, "index element type mismatch");
    const HostTree *t = nullptr;
    PNCHK(tree_node(ix, node, &t));
    *out = host_tree_radius(t, node);
    return PN_OK;
}
extern "C" int pn_tree_compare_nodes(const pn_index *ix, uint64_t x, uint64_t y, int *ordering) {
    if (!ordering) return fail(PN_ERR_INVALID, "ordering is NULL");
    const HostTree *t = nullptr;
    PNCHK(tree_node(ix, x, &t));
    PNCHK(tree_node(ix, y, &t));
    *ordering = host_tree_compare(t, x, y);
    return PN_OK;
}
extern "C" int pn_tree_node_distance_lower_bound_f32(const pn_index *ix, uint64_t n1, uint64_t n2, float *out) {
    if (!out) return fail(PN_ERR_INVALID, "out is NULL");
    if (ix && ix->elem_bytes != 4) return fail(PN_ERR_INVALID, "index element type mismatch");
    const HostTree *t = nullptr;
    PNCHK(tree_node(ix, n1, &t));
    PNCHK(tree_node(ix, n2, &t));
    *out = (float)host_tree_lower_bound(t, n1, n2);
    return PN_OK;
}
extern "C" int pn_tree_node_distance_lower_bound_f64(const pn_index *ix, uint64_t n1, uint64_t n2, double *out) {
    if (!out) return fail(PN_ERR_INVALID, "out is NULL");
    if (ix && ix->elem_bytes != 8) return fail(PN_ERR_INVALID, "index element type mismatch");
    const HostTree *t = nullptr;
    PNCHK(tree_node(ix, n1, &t));
    PNCHK(tree_node(ix, n2, &t));
    *out = host_tree_lower_bound(t, n1, n2);
    return PN_OK;
}
// test hook of the above: the centroid of a node (the reference keeps it private; tests compare it with the oracle's)
extern "C" int pn_tree_centroid_of(const pn_index *ix, uint64_t node, void *out_dim_elems) {
    if (!out_dim_elems) return fail(PN_ERR_INVALID, "out is NULL");
    const HostTree *t = nullptr;
    PNCHK(tree_node(ix, node, &t));
    memcpy(out_dim_elems, host_tree_centroid(t, node), ix->dim * (size_t)ix->elem_bytes);
    return PN_OK;
}

// ---------------------------------------------------------------------------
// shard merge + generator
// ---------------------------------------------------------------------------
template <typename T>
static int merge_topk_device_impl(const uint64_t *d_idx_parts, const T *d_dist_parts, size_t n_parts, size_t idx_part_stride,
                                  size_t dist_part_stride, size_t nq, size_t k_part, size_t k_out, uint64_t *d_idx_out,
                                  T *d_dist_out, int device, void *stream, bool signed_keys = false) {
    if (nq == 0 || k_out == 0) return PN_OK;
    if (!d_idx_parts || !d_dist_parts || !d_idx_out || !d_dist_out) return fail(PN_ERR_INVALID, "NULL argument");
    if (n_parts == 0 || k_part == 0) return fail(PN_ERR_INVALID, "empty parts");
    if (n_parts > 0x7FFFFFFFull || k_part > 0x7FFFFFFFull || k_out > 0x7FFFFFFFull || nq > 0x7FFFFFFFull)
        return fail(PN_ERR_UNSUPPORTED, "merge sizes beyond 2^31");
    // (any n_parts x k_part: beyond 64 KiB of LDS the merge ranks by binary search over the sorted parts, select.hip)
    PNCHK(check_device(device));
    DeviceGuard g(device);
    if (!g.ok) return fail(PN_ERR_DEVICE, "hipSetDevice(%d) failed", device);
    if constexpr (sizeof(T) == 4)
        HIPCHK(launch_merge_topk_f32(d_idx_parts, d_dist_parts, (int)n_parts, idx_part_stride, dist_part_stride, (int)nq,
                                     (int)k_part, (int)k_out, d_idx_out, d_dist_out, (hipStream_t)stream, nullptr, nullptr, 0,
                                     nullptr, signed_keys));
    else
        HIPCHK(launch_merge_topk_f64(d_idx_parts, d_dist_parts, (int)n_parts, idx_part_stride, dist_part_stride, (int)nq,
                                     (int)k_part, (int)k_out, d_idx_out, d_dist_out, (hipStream_t)stream, nullptr, nullptr, 0,
                                     nullptr, signed_keys));
    return PN_OK;
}
namespace pn {
// the shard merge with the key order named (signed_keys: a Cosine handle's distances, sharded.hip)
int merge_topk_device_keys_f32(const uint64_t *pi, const float *pd, size_t np, size_t is, size_t ds, size_t nq, size_t kp,
                               size_t ko, uint64_t *oi, float *od, int device, void *stream, bool signed_keys) {
    return merge_topk_device_impl<float>(pi, pd, np, is, ds, nq, kp, ko, oi, od, device, stream, signed_keys);
}
int merge_topk_device_keys_f64(const uint64_t *pi, const double *pd, size_t np, size_t is, size_t ds, size_t nq, size_t kp,
                               size_t ko, uint64_t *oi, double *od, int device, void *stream, bool signed_keys) {
    return merge_topk_device_impl<double>(pi, pd, np, is, ds, nq, kp, ko, oi, od, device, stream, signed_keys);
}
}  // namespace pn
extern "C" int pn_merge_topk_device_f32(const uint64_t *d_idx_parts, const float *d_dist_parts, size_t n_parts,
                                        size_t idx_part_stride, size_t dist_part_stride, size_t nq, size_t k_part,
                                        size_t k_out, uint64_t *d_idx_out, float *d_dist_out, int device,
                                        void *stream) {
    return merge_topk_device_impl<float>(d_idx_parts, d_dist_parts, n_parts, idx_part_stride, dist_part_stride, nq, k_part,
                                         k_out, d_idx_out, d_dist_out, device, stream);
}
extern "C" int pn_merge_topk_device_f64(const uint64_t *d_idx_parts, const double *d_dist_parts, size_t n_parts,
                                        size_t idx_part_stride, size_t dist_part_stride, size_t nq, size_t k_part,
                                        size_t k_out, uint64_t *d_idx_out, double *d_dist_out, int device,
                                        void *stream) {
    return merge_topk_device_impl<double>(d_idx_parts, d_dist_parts, n_parts, idx_part_stride, dist_part_stride, nq, k_part,
                                          k_out, d_idx_out, d_dist_out, device, stream);
}

extern "C" int pn_fill_uniform_device_f32(float *d_out, uint64_t count, uint64_t seed, uint64_t first_counter,
                                          int device, void *stream) {
    if (count == 0) return PN_OK;
    if (!d_out) return fail(PN_ERR_INVALID, "d_out is NULL");
    PNCHK(check_device(device));
    DeviceGuard g(device);
    if (!g.ok) return fail(PN_ERR_DEVICE, "hipSetDevice(%d) failed", device);
    HIPCHK(launch_fill_uniform_f32(d_out, count, seed, first_counter, (hipStream_t)stream));
    return PN_OK;
}
