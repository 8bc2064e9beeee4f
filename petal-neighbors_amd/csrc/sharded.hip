// sharded.hip -- row-sharded corpora behind the C ABI (include/petal_mi355x.h, pn_sharded_*): SURVEY.md 8(e).
//
// The corpus is split into contiguous row ranges (shard g = rows [g ceil(N/G), ...)), every shard is an ordinary
// pn_index with PN_OPT_INDEX_BASE = its first row, queries are replicated, and the per-shard exact top-k -- written
// straight into ONE packed buffer {idx[nq][k'] | dist[nq][k']} per GPU -- are exchanged by ONE RCCL all-gather per
// query batch and merged by merge_topk_kernel under the (distance, index) order.  Exact top-k is decomposable and the
// order is total, so the answer does not depend on the number of shards.
//
// Two ways to hold the shards:
//   * pn_sharded_create_f32: ONE process drives n_devices GPUs (ncclCommInitAll over the distinct devices; a device
//     named several times holds several shards, merged locally before the exchange -- this is also how the path is
//     exercised on a single GPU);
//   * pn_sharded_create_rank_device_f32: ONE process per GPU (the launch model of bench.py / torch.distributed.run);
//     rank 0 obtains a communicator id from pn_comm_unique_id, the host program carries it to the other ranks by
//     whatever means it has, every rank hands it in (ncclCommInitRank).
// RCCL is loaded at run time (dlopen: the copy already in the process if there is one, e.g. PyTorch's), so the
// library itself has no link-time dependency on it and CPU-only hosts can still load the ABI.
//
// Query batches above kShardChunk queries are cut into chunks whose exchange + merge run on a second stream while
// the next chunk's filter runs on the caller's (two packed-buffer sets, events in both directions).
#include <dlfcn.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include <rccl/rccl.h>

#include "../../include/petal_mi355x.h"
#include "pn_internal.h"

using namespace pn;

#define SHIP(expr)                                                                                          \
    do {                                                                                                    \
        hipError_t e_ = (expr);                                                                             \
        if (e_ != hipSuccess)                                                                               \
            return set_error(e_ == hipErrorOutOfMemory ? PN_ERR_NOMEM : PN_ERR_DEVICE, "%s: %s", #expr,     \
                             hipGetErrorString(e_));                                                        \
    } while (0)
#define SPN(expr)                     \
    do {                              \
        int rc_ = (expr);             \
        if (rc_ != PN_OK) return rc_; \
    } while (0)

// ---------------------------------------------------------------------------
// RCCL, resolved at run time
// ---------------------------------------------------------------------------
namespace {
struct Rccl {
    void *handle = nullptr;
    std::string err;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool ok() const { return handle != nullptr && err.empty(); }
};

Rccl load_rccl() {
    Rccl r;
    // the copy already mapped into the process first (PyTorch-ROCm ships its own): two RCCLs in one process would
    // each keep their own view of the devices
    const char *names[] = {"librccl.so.1", "librccl.so"};
    for (const char *n : names)
        if (!r.handle) r.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
    const char *paths[] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
    for (const char *n : paths)
        if (!r.handle) r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (!r.handle) {
        const char *e = dlerror();
        r.err = std::string("RCCL is not loadable: ") + (e ? e : "unknown dlopen failure");
        return r;
    }
    auto sym = [&](const char *name) -> void * {
        void *p = dlsym(r.handle, name);
        if (!p && r.err.empty()) r.err = std::string("RCCL does not export ") + name;
        return p;
    };
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
    r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(sym("ncclCommInitAll"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    return r;
}
Rccl &rccl() {
    static Rccl r = load_rccl();
    return r;
}
int need_rccl() {
    Rccl &r = rccl();
    if (!r.ok()) return set_error(PN_ERR_COMM, "%s", r.err.c_str());
    return PN_OK;
}
}  // namespace

#define SNCCL(expr)                                                                                   \
    do {                                                                                              \
        ncclResult_t r_ = (expr);                                                                     \
        if (r_ != ncclSuccess) return set_error(PN_ERR_COMM, "%s: %s", #expr, rccl().GetErrorString(r_)); \
    } while (0)

static_assert(sizeof(ncclUniqueId) == PN_COMM_ID_BYTES, "PN_COMM_ID_BYTES must be sizeof(ncclUniqueId)");

// ---------------------------------------------------------------------------
// the sharded handle
// ---------------------------------------------------------------------------
namespace {
constexpr size_t kShardChunk = 131072;  // queries per exchange when a batch is cut (two chunks in flight)

// grows only; an outgrown allocation may still be read by kernels of an asynchronous call that has already returned,
// so it is retired to its GPU's list and freed once that GPU's end-of-use event has passed (acquire_dev), never with a
// hipFree (a device-wide wait) inside an "enqueue and return" call
struct Buf {
    void *p = nullptr;
    size_t bytes = 0;
    std::vector<void *> *retired = nullptr;
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
        const size_t want = need + need / 8 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) return set_error(PN_ERR_NOMEM, "hipMalloc(%zu): %s", want, hipGetErrorString(e));
        bytes = want;
        return PN_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
};

struct Part {  // one row shard held by this process
    pn_index *ix = nullptr;  // nullptr: the shard has no rows (N < number of shards)
    uint64_t lo = 0, hi = 0;
    int dev_slot = 0;
};
struct Dev {  // one GPU this process drives = one rank of the communicator
    int device = 0;
    int comm_rank = 0;
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;     // host entry points
    hipStream_t comm_stream = nullptr;  // exchange + merge of a chunk while the next chunk is filtered
    hipEvent_t ev_local[2] = {nullptr, nullptr}, ev_merged[2] = {nullptr, nullptr}, ev_done = nullptr;
    std::vector<int> parts;
    Buf q, out_idx, out_dist, lparts, pack[2], gathered[2], rad_a, rad_b;
    // End of the last call that used this GPU's exchange buffers (pack / gathered / lparts) and the stream it was
    // enqueued on: the device entry point returns while its kernels run, so a later call on ANOTHER stream is ordered
    // behind this event before it touches the buffers (`mu` only serialises the enqueueing).  Same scheme as
    // Workspace::done in index.hip.
    hipEvent_t ev_use = nullptr;
    hipStream_t last_stream = nullptr;
    bool in_flight = false;
    std::vector<void *> retired;
    // PN_OPT_PROFILE: per call three timing events (start, local half done, merged) in a ring, resolved when a slot
    // is reused and by pn_sharded_get_stats -- never inside a call
    static constexpr int kProf = 32;
    hipEvent_t ev_prof[kProf][3] = {};
    bool prof_pending[kProf] = {};
    unsigned prof_next = 0;
    double shard_ms = 0.0, exchange_ms = 0.0;
    // (Bufs point at this Dev's retired list: adopt() after the vector of Devs has its final size)
    void adopt() {
        Buf *bs[] = {&q, &out_idx, &out_dist, &lparts, &pack[0], &pack[1], &gathered[0], &gathered[1], &rad_a, &rad_b};
        for (Buf *b : bs) b->retired = &retired;
    }
    void free_retired() {
        for (void *r : retired) (void)hipFree(r);
        retired.clear();
    }
};
struct SetGuard {
    int prev = -1;
    bool ok = true;
    explicit SetGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) ok = hipSetDevice(dev) == hipSuccess;
    }
    ~SetGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};
}  // namespace

struct pn_sharded {
    int elem_bytes = 4;  // 4: f32 corpus and queries (the *_f32 entry points), 8: f64
    int metric = 0;      // 0 Euclidean, 1 Cosine (every shard a pn_index_create_cosine_* index; the merge orders signed keys)
    int profile = 0;  // PN_OPT_PROFILE (also forwarded to the shards)
    uint64_t n_total = 0;
    size_t dim = 0;
    uint64_t per = 0;  // rows per shard = ceil(n_total / n_shards)
    int n_shards = 0;  // over all processes
    int world = 1;     // ranks of the communicator (GPUs)
    int max_parts_per_dev = 1;  // most shards any GPU holds (1 with one process per GPU)
    bool rank_mode = false;
    bool exchange_always = false;  // PN_OPT_EXCHANGE_ALWAYS
    std::vector<Part> parts;  // local ones
    // queries take a `const pn_sharded *` (BallTree queries take &self); what they touch below -- streams, exchange
    // buffers, the communicator -- is shared per-handle state, internally serialised by `mu`: one query at a time
    mutable std::vector<Dev> devs;  // local ones
    mutable std::mutex mu;
};

// 8-byte words of one packed part: [nq][kp] indices, then [nq][kp] distances of the handle's element type
template <typename T>
static size_t packed_words(size_t nq, size_t kp) {
    return sizeof(T) == 4 ? nq * kp + (nq * kp + 1) / 2 : 2 * nq * kp;
}
// the single-index entry points of either element type (the sharded handle is generic over A like the reference,
// src/ball_tree.rs:26-30)
template <typename T> struct ShT;
template <> struct ShT<float> {
    static int create(const float *p, size_t n, size_t c, ptrdiff_t rs, ptrdiff_t cs, int dev, pn_index **o, int metric) {
        return metric == 1 ? pn_index_create_cosine_f32(p, n, c, rs, cs, dev, o) : pn_index_create_f32(p, n, c, rs, cs, dev, o);
    }
    static int create_device(const float *p, size_t n, size_t c, size_t rs, int dev, void *st, pn_index **o) {
        return pn_index_create_device_f32(p, n, c, rs, dev, st, o);
    }
    static int query_strided(const pn_index *ix, const float *q, size_t nq, size_t qc, size_t qs, size_t k, uint64_t *oi,
                             float *od, size_t os, hipStream_t s) {
        return query_device_strided_f32(ix, q, nq, qc, qs, k, oi, od, os, s);
    }
    static int merge(const uint64_t *pi, const float *pd, size_t np, size_t is, size_t ds, size_t nq, size_t kp, size_t ko,
                     uint64_t *oi, float *od, int dev, void *s, bool signed_keys) {
        return merge_topk_device_keys_f32(pi, pd, np, is, ds, nq, kp, ko, oi, od, dev, s, signed_keys);
    }
    static int radius_device(const pn_index *ix, const float *q, size_t nq, size_t qc, size_t qs, float r, uint64_t *off,
                             uint64_t *idx, size_t cap, uint64_t *tot, void *st) {
        return pn_query_radius_device_f32(ix, q, nq, qc, qs, r, off, idx, cap, tot, st);
    }
    static int radius(const pn_index *ix, const float *q, size_t nq, size_t qc, ptrdiff_t qs, float r, uint64_t *off,
                      uint64_t **out) {
        return pn_query_radius_f32(ix, q, nq, qc, qs, r, off, out);
    }
};
template <> struct ShT<double> {
    static int create(const double *p, size_t n, size_t c, ptrdiff_t rs, ptrdiff_t cs, int dev, pn_index **o, int metric) {
        return metric == 1 ? pn_index_create_cosine_f64(p, n, c, rs, cs, dev, o) : pn_index_create_f64(p, n, c, rs, cs, dev, o);
    }
    static int create_device(const double *p, size_t n, size_t c, size_t rs, int dev, void *st, pn_index **o) {
        return pn_index_create_device_f64(p, n, c, rs, dev, st, o);
    }
    static int query_strided(const pn_index *ix, const double *q, size_t nq, size_t qc, size_t qs, size_t k, uint64_t *oi,
                             double *od, size_t os, hipStream_t s) {
        return query_device_strided_f64(ix, q, nq, qc, qs, k, oi, od, os, s);
    }
    static int merge(const uint64_t *pi, const double *pd, size_t np, size_t is, size_t ds, size_t nq, size_t kp, size_t ko,
                     uint64_t *oi, double *od, int dev, void *s, bool signed_keys) {
        return merge_topk_device_keys_f64(pi, pd, np, is, ds, nq, kp, ko, oi, od, dev, s, signed_keys);
    }
    static int radius_device(const pn_index *ix, const double *q, size_t nq, size_t qc, size_t qs, double r, uint64_t *off,
                             uint64_t *idx, size_t cap, uint64_t *tot, void *st) {
        return pn_query_radius_device_f64(ix, q, nq, qc, qs, r, off, idx, cap, tot, st);
    }
    static int radius(const pn_index *ix, const double *q, size_t nq, size_t qc, ptrdiff_t qs, double r, uint64_t *off,
                      uint64_t **out) {
        return pn_query_radius_f64(ix, q, nq, qc, qs, r, off, out);
    }
};
static int check_elem(const pn_sharded *sh, size_t bytes) {
    if (sh->elem_bytes != (int)bytes) return set_error(PN_ERR_INVALID, "handle element type mismatch (f%d handle)", sh->elem_bytes * 8);
    return PN_OK;
}

extern "C" int pn_comm_unique_id(void *id_out) {
    if (!id_out) return set_error(PN_ERR_INVALID, "id_out is NULL");
    SPN(need_rccl());
    ncclUniqueId id;
    SNCCL(rccl().GetUniqueId(&id));
    memcpy(id_out, &id, sizeof id);
    return PN_OK;
}

static void destroy_dev(Dev &d) {
    SetGuard g(d.device);
    // what this handle enqueued on d -- on its own streams or a caller's -- has finished: the end-of-use event of the last
    // call (DevUse) and the handle's own streams; not a device-wide wait
    if (d.in_flight && d.ev_use) (void)hipEventSynchronize(d.ev_use);
    if (d.stream) (void)hipStreamSynchronize(d.stream);
    if (d.comm_stream) (void)hipStreamSynchronize(d.comm_stream);
    if (d.comm && rccl().ok()) (void)rccl().CommDestroy(d.comm);
    Buf *bufs[] = {&d.q, &d.out_idx, &d.out_dist, &d.lparts, &d.pack[0], &d.pack[1], &d.gathered[0], &d.gathered[1],
                   &d.rad_a, &d.rad_b};
    for (Buf *b : bufs) b->release();
    d.free_retired();
    for (auto &tr : d.ev_prof)
        for (hipEvent_t e : tr)
            if (e) (void)hipEventDestroy(e);
    hipEvent_t evs[] = {d.ev_local[0], d.ev_local[1], d.ev_merged[0], d.ev_merged[1], d.ev_done, d.ev_use};
    for (hipEvent_t e : evs)
        if (e) (void)hipEventDestroy(e);
    if (d.stream) (void)hipStreamDestroy(d.stream);
    if (d.comm_stream) (void)hipStreamDestroy(d.comm_stream);
}

extern "C" void pn_sharded_destroy(pn_sharded *sh) {
    if (!sh) return;
    for (Part &p : sh->parts)
        if (p.ix) pn_index_destroy(p.ix);
    for (Dev &d : sh->devs) destroy_dev(d);
    delete sh;
}

static int init_dev_resources(Dev &d) {
    SetGuard g(d.device);
    if (!g.ok) return set_error(PN_ERR_DEVICE, "hipSetDevice(%d) failed", d.device);
    SHIP(hipStreamCreateWithFlags(&d.stream, hipStreamNonBlocking));
    SHIP(hipStreamCreateWithFlags(&d.comm_stream, hipStreamNonBlocking));
    for (int i = 0; i < 2; ++i) {
        SHIP(hipEventCreateWithFlags(&d.ev_local[i], hipEventDisableTiming));
        SHIP(hipEventCreateWithFlags(&d.ev_merged[i], hipEventDisableTiming));
    }
    SHIP(hipEventCreateWithFlags(&d.ev_done, hipEventDisableTiming));
    SHIP(hipEventCreateWithFlags(&d.ev_use, hipEventDisableTiming));
    return PN_OK;
}

// BallTree::new over n_devices row shards, one process.  Validation is the reference's
// (src/ball_tree.rs:44-49), done on the whole array before anything is split.
template <typename T>
static int sharded_create(const T *points, size_t n_rows, size_t n_cols, ptrdiff_t row_stride, ptrdiff_t col_stride,
                          const int *devices, int n_devices, pn_sharded **out, int metric = 0) {
    if (!out) return set_error(PN_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (n_rows == 0) return set_error(PN_ERR_EMPTY, "array is empty");
    if (n_cols > 1 && col_stride != 1) return set_error(PN_ERR_NOT_CONTIGUOUS, "array is not contiguous in memory");
    if (n_cols == 0 && n_rows >= 2) return set_error(PN_ERR_EMPTY_MATRIX, "empty matrix");
    if (!devices || n_devices < 1 || n_devices > 64) return set_error(PN_ERR_INVALID, "devices: 1..64 entries");
    if (!points && n_cols) return set_error(PN_ERR_INVALID, "points is NULL");
    if (row_stride < 0) return set_error(PN_ERR_UNSUPPORTED, "negative row stride");
    int n_gpu = 0;
    SPN(pn_device_count(&n_gpu));
    if (n_gpu <= 0) return set_error(PN_ERR_DEVICE, "no usable GPU; this library has no CPU path");
    for (int i = 0; i < n_devices; ++i)
        if (devices[i] < 0 || devices[i] >= n_gpu)
            return set_error(PN_ERR_INVALID, "device %d out of range [0,%d)", devices[i], n_gpu);
    SPN(need_rccl());
    pn_sharded *sh = new (std::nothrow) pn_sharded();
    if (!sh) return set_error(PN_ERR_NOMEM, "host allocation failed");
    sh->elem_bytes = (int)sizeof(T);
    sh->metric = metric;
    sh->n_total = n_rows;
    sh->dim = n_cols;
    sh->n_shards = n_devices;
    sh->per = (n_rows + (size_t)n_devices - 1) / (size_t)n_devices;
    // distinct devices in order of first appearance = ranks of the communicator
    std::vector<int> distinct;
    for (int i = 0; i < n_devices; ++i)
        if (std::find(distinct.begin(), distinct.end(), devices[i]) == distinct.end()) distinct.push_back(devices[i]);
    sh->world = (int)distinct.size();
    sh->devs.resize(distinct.size());
    for (Dev &d : sh->devs) d.adopt();
    int rc = PN_OK;
    for (size_t r = 0; r < distinct.size() && rc == PN_OK; ++r) {
        sh->devs[r].device = distinct[r];
        sh->devs[r].comm_rank = (int)r;
        rc = init_dev_resources(sh->devs[r]);
    }
    for (int g = 0; g < n_devices && rc == PN_OK; ++g) {
        Part p;
        p.lo = std::min<uint64_t>(n_rows, (uint64_t)g * sh->per);
        p.hi = std::min<uint64_t>(n_rows, p.lo + sh->per);
        p.dev_slot = (int)(std::find(distinct.begin(), distinct.end(), devices[g]) - distinct.begin());
        if (p.hi > p.lo) {
            rc = ShT<T>::create(points + (size_t)p.lo * (size_t)row_stride, (size_t)(p.hi - p.lo), n_cols, row_stride,
                                col_stride, devices[g], &p.ix, metric);
            if (rc == PN_OK) rc = pn_index_set_option(p.ix, PN_OPT_INDEX_BASE, (int64_t)p.lo);
        }
        sh->parts.push_back(p);
        sh->devs[p.dev_slot].parts.push_back(g);
    }
    for (const Dev &d : sh->devs) sh->max_parts_per_dev = std::max(sh->max_parts_per_dev, (int)d.parts.size());
    if (rc == PN_OK) {
        std::vector<ncclComm_t> comms(distinct.size(), nullptr);
        ncclResult_t r = rccl().CommInitAll(comms.data(), (int)distinct.size(), distinct.data());
        if (r != ncclSuccess)
            rc = set_error(PN_ERR_COMM, "ncclCommInitAll over %zu GPUs: %s", distinct.size(), rccl().GetErrorString(r));
        for (size_t i = 0; i < distinct.size(); ++i) sh->devs[i].comm = comms[i];
    }
    if (rc != PN_OK) {
        const std::string keep = pn_last_error();
        pn_sharded_destroy(sh);
        return set_error(rc, "%s", keep.c_str());
    }
    *out = sh;
    return PN_OK;
}

// One process per GPU: this rank's rows (already on `device`) of an n_total-row corpus.  Shard bounds are the
// library's: rank r holds rows [r ceil(N/W), min(N, (r+1) ceil(N/W))) -- n_local must be exactly that many (a rank
// beyond the corpus passes n_local = 0 and still takes part in every exchange).
template <typename T>
static int sharded_create_rank_device(const T *d_rows, size_t n_local, size_t n_cols, size_t row_stride, uint64_t n_total,
                                      int rank, int world, const void *comm_id, int device, void *stream,
                                      pn_sharded **out) {
    if (!out) return set_error(PN_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (n_total == 0) return set_error(PN_ERR_EMPTY, "array is empty");
    if (world < 1 || rank < 0 || rank >= world) return set_error(PN_ERR_INVALID, "rank %d of %d", rank, world);
    if (!comm_id) return set_error(PN_ERR_INVALID, "comm_id is NULL");
    const uint64_t per = (n_total + (uint64_t)world - 1) / (uint64_t)world;
    const uint64_t lo = std::min<uint64_t>(n_total, (uint64_t)rank * per), hi = std::min<uint64_t>(n_total, lo + per);
    if ((uint64_t)n_local != hi - lo)
        return set_error(PN_ERR_INVALID, "rank %d of %d holds rows [%llu, %llu) of %llu: n_local must be %llu, got %zu", rank,
                         world, (unsigned long long)lo, (unsigned long long)hi, (unsigned long long)n_total,
                         (unsigned long long)(hi - lo), n_local);
    int n_gpu = 0;
    SPN(pn_device_count(&n_gpu));
    if (device < 0 || device >= n_gpu) return set_error(PN_ERR_INVALID, "device %d out of range [0,%d)", device, n_gpu);
    SPN(need_rccl());
    pn_sharded *sh = new (std::nothrow) pn_sharded();
    if (!sh) return set_error(PN_ERR_NOMEM, "host allocation failed");
    sh->elem_bytes = (int)sizeof(T);
    sh->n_total = n_total;
    sh->dim = n_cols;
    sh->n_shards = world;
    sh->world = world;
    sh->per = per;
    sh->rank_mode = true;
    sh->devs.resize(1);
    sh->devs[0].adopt();
    sh->devs[0].device = device;
    sh->devs[0].comm_rank = rank;
    int rc = init_dev_resources(sh->devs[0]);
    Part p;
    p.lo = lo;
    p.hi = hi;
    if (rc == PN_OK && hi > lo) {
        rc = ShT<T>::create_device(d_rows, n_local, n_cols, row_stride, device, stream, &p.ix);
        if (rc == PN_OK) rc = pn_index_set_option(p.ix, PN_OPT_INDEX_BASE, (int64_t)lo);
    }
    sh->parts.push_back(p);
    sh->devs[0].parts.push_back(0);
    if (rc == PN_OK) {
        SetGuard g(device);
        ncclUniqueId id;
        memcpy(&id, comm_id, sizeof id);
        ncclResult_t r = rccl().CommInitRank(&sh->devs[0].comm, world, id, rank);
        if (r != ncclSuccess)
            rc = set_error(PN_ERR_COMM, "ncclCommInitRank(rank %d of %d): %s", rank, world, rccl().GetErrorString(r));
    }
    if (rc != PN_OK) {
        const std::string keep = pn_last_error();
        pn_sharded_destroy(sh);
        return set_error(rc, "%s", keep.c_str());
    }
    *out = sh;
    return PN_OK;
}

extern "C" int pn_sharded_create_f32(const float *points, size_t n_rows, size_t n_cols, ptrdiff_t row_stride,
                                     ptrdiff_t col_stride, const int *devices, int n_devices, pn_sharded **out) {
    return sharded_create<float>(points, n_rows, n_cols, row_stride, col_stride, devices, n_devices, out);
}
extern "C" int pn_sharded_create_f64(const double *points, size_t n_rows, size_t n_cols, ptrdiff_t row_stride,
                                     ptrdiff_t col_stride, const int *devices, int n_devices, pn_sharded **out) {
    return sharded_create<double>(points, n_rows, n_cols, row_stride, col_stride, devices, n_devices, out);
}
// BallTree::new(points, Cosine) over row shards driven by this process (one process per GPU: not offered -- there is no
// Cosine index from rows in HBM).  Every shard is a pn_index_create_cosine_* index (exact scan under Cosine::distance); the
// part merge orders the order-preserving keys of ALL floats, as the single index does (distances a few ulp below zero).
extern "C" int pn_sharded_create_cosine_f32(const float *points, size_t n_rows, size_t n_cols, ptrdiff_t row_stride,
                                            ptrdiff_t col_stride, const int *devices, int n_devices, pn_sharded **out) {
    return sharded_create<float>(points, n_rows, n_cols, row_stride, col_stride, devices, n_devices, out, 1);
}
extern "C" int pn_sharded_create_cosine_f64(const double *points, size_t n_rows, size_t n_cols, ptrdiff_t row_stride,
                                            ptrdiff_t col_stride, const int *devices, int n_devices, pn_sharded **out) {
    return sharded_create<double>(points, n_rows, n_cols, row_stride, col_stride, devices, n_devices, out, 1);
}
extern "C" int pn_sharded_create_rank_device_f32(const float *d_rows, size_t n_local, size_t n_cols, size_t row_stride,
                                                 uint64_t n_total, int rank, int world, const void *comm_id, int device,
                                                 void *stream, pn_sharded **out) {
    return sharded_create_rank_device<float>(d_rows, n_local, n_cols, row_stride, n_total, rank, world, comm_id, device,
                                             stream, out);
}
extern "C" int pn_sharded_create_rank_device_f64(const double *d_rows, size_t n_local, size_t n_cols, size_t row_stride,
                                                 uint64_t n_total, int rank, int world, const void *comm_id, int device,
                                                 void *stream, pn_sharded **out) {
    return sharded_create_rank_device<double>(d_rows, n_local, n_cols, row_stride, n_total, rank, world, comm_id, device,
                                              stream, out);
}

extern "C" int pn_sharded_info(const pn_sharded *sh, pn_sharded_info_t *out) {
    if (!sh || !out) return set_error(PN_ERR_INVALID, "NULL argument");
    out->n_points = sh->n_total;
    out->dim = sh->dim;
    out->n_shards = sh->n_shards;
    out->world = sh->world;
    out->local_shards = (int32_t)sh->parts.size();
    out->rank = sh->rank_mode ? sh->devs[0].comm_rank : 0;
    out->local_first_row = sh->parts.empty() ? 0 : sh->parts.front().lo;
    out->local_rows = 0;
    out->mfma_eligible = out->bf16_eligible = 1;
    for (const Part &p : sh->parts) {
        out->local_rows += p.hi - p.lo;
        pn_info pi{};
        if (p.ix && pn_index_info(p.ix, &pi) == PN_OK) {
            out->mfma_eligible &= pi.mfma_eligible;
            out->bf16_eligible &= pi.bf16_eligible;
        }
    }
    return PN_OK;
}

extern "C" int pn_sharded_set_option(pn_sharded *sh, int option, int64_t value) {
    if (!sh) return set_error(PN_ERR_INVALID, "handle is NULL");
    if (option == PN_OPT_INDEX_BASE) return set_error(PN_ERR_INVALID, "the index base of a shard is its first row");
    if (option == PN_OPT_EXCHANGE_ALWAYS) {
        sh->exchange_always = value != 0;
        return PN_OK;
    }
    if (option == PN_OPT_PROFILE) sh->profile = value != 0 ? 1 : 0;  // (and on to the shards below)
    for (Part &p : sh->parts)
        if (p.ix) SPN(pn_index_set_option(p.ix, option, value));
    return PN_OK;
}

static void prof_resolve(Dev &d, int slot, bool wait) {
    if (!d.prof_pending[slot]) return;
    if (wait) (void)hipEventSynchronize(d.ev_prof[slot][2]);
    float a = 0.0f, b = 0.0f;
    if (hipEventElapsedTime(&a, d.ev_prof[slot][0], d.ev_prof[slot][1]) == hipSuccess &&
        hipEventElapsedTime(&b, d.ev_prof[slot][1], d.ev_prof[slot][2]) == hipSuccess) {
        d.shard_ms += a;
        d.exchange_ms += b;
        d.prof_pending[slot] = false;
    } else if (wait) {
        d.prof_pending[slot] = false;
    }
    (void)hipGetLastError();
}

extern "C" int pn_sharded_get_stats(const pn_sharded *sh, pn_stats *out, int reset) {
    if (!sh || !out) return set_error(PN_ERR_INVALID, "NULL argument");
    pn_stats acc{};
    {
        std::lock_guard<std::mutex> lk(sh->mu);
        for (Dev &d : sh->devs) {
            SetGuard g(d.device);
            for (int i = 0; i < Dev::kProf; ++i) prof_resolve(d, i, true);
            acc.shard_ms += d.shard_ms;
            acc.exchange_ms += d.exchange_ms;
            if (reset) d.shard_ms = d.exchange_ms = 0.0;
        }
    }
    for (const Part &p : sh->parts) {
        if (!p.ix) continue;
        pn_stats s{};
        SPN(pn_index_get_stats(p.ix, &s, reset));
        acc.queries += s.queries;
        acc.fallback_queries += s.fallback_queries;
        acc.candidates += s.candidates;
        acc.hot_launches += s.hot_launches;
        acc.hot_ms += s.hot_ms;
        acc.last_call_ms += s.last_call_ms;
        acc.radius_results += s.radius_results;
        acc.evaluations += s.evaluations;
    }
    *out = acc;
    return PN_OK;
}

// ---------------------------------------------------------------------------
// k-NN
// ---------------------------------------------------------------------------
// The local half of one chunk on one GPU: every local shard's top-kp, merged when the GPU holds several shards, into
// the packed buffer `set` with kd slots per query (kd = what one GPU can contribute at most).  Enqueued on `s`.
template <typename T>
static int enqueue_local(const pn_sharded *sh, Dev &d, const T *d_q, size_t nq, size_t q_cols, size_t q_stride,
                         size_t k, size_t kp, size_t kd, int set, hipStream_t s) {
    const size_t words = packed_words<T>(nq, kd), pwords = packed_words<T>(nq, kp);
    constexpr size_t kDW = 8 / sizeof(T);  // distances per 8-byte word
    SPN(d.pack[set].ensure(words * 8));
    const size_t np = d.parts.size();
    uint64_t *oi = (uint64_t *)d.pack[set].p;
    T *od = reinterpret_cast<T *>(oi + nq * kd);
    if (np == 1) {  // straight into the buffer the all-gather sends
        const Part &p = sh->parts[d.parts[0]];
        if (p.hi - p.lo < kd)  // absent slots: index ~0 (and a NaN distance) -- the merge kernel skips them
            SHIP(hipMemsetAsync(oi, 0xFF, words * 8, s));
        if (p.ix) SPN(ShT<T>::query_strided(p.ix, d_q, nq, q_cols, q_stride, k < kd ? k : kd, oi, od, kd, s));
        return PN_OK;
    }
    SPN(d.lparts.ensure(np * pwords * 8));
    uint64_t *stage = (uint64_t *)d.lparts.p;
    for (size_t i = 0; i < np; ++i) {
        const Part &p = sh->parts[d.parts[i]];
        uint64_t *pi = stage + i * pwords;
        T *pd = reinterpret_cast<T *>(pi + nq * kp);
        if (p.hi - p.lo < kp) SHIP(hipMemsetAsync(pi, 0xFF, pwords * 8, s));
        if (p.ix) SPN(ShT<T>::query_strided(p.ix, d_q, nq, q_cols, q_stride, k < kp ? k : kp, pi, pd, kp, s));
    }
    return ShT<T>::merge(stage, reinterpret_cast<const T *>(stage + nq * kp), np, pwords, kDW * pwords, nq, kp, kd, oi, od,
                         d.device, s, sh->metric == 1);
}

// exchange + final merge of one chunk on one GPU, enqueued on `s`: one all-gather of the packed buffer, then the
// (distance, index) merge of the world's parts into d_idx/d_dist [nq][k_out]
template <typename T>
static int enqueue_merge(const pn_sharded *sh, Dev &d, size_t nq, size_t kd, size_t k_out, int set, uint64_t *d_idx,
                         T *d_dist, hipStream_t s) {
    const size_t words = packed_words<T>(nq, kd);
    const uint64_t *g = (const uint64_t *)d.gathered[set].p;
    return ShT<T>::merge(g, reinterpret_cast<const T *>(g + nq * kd), (size_t)sh->world, words, (8 / sizeof(T)) * words, nq,
                         kd, k_out, d_idx, d_dist, d.device, s, sh->metric == 1);
}

static size_t k_part_of(const pn_sharded *sh, size_t k) {  // slots per query a SHARD fills
    const uint64_t largest = std::min<uint64_t>(sh->per, sh->n_total);  // shard 0 is never smaller than another
    return (size_t)std::min<uint64_t>(k, largest ? largest : 1);
}
static size_t k_dev_of(const pn_sharded *sh, size_t k) {  // slots per query a GPU sends: the same on every rank
    uint64_t most = (uint64_t)sh->max_parts_per_dev * sh->per;
    if (most > sh->n_total) most = sh->n_total;
    return (size_t)std::min<uint64_t>(k, most ? most : 1);
}

// A call that is about to use d's exchange buffers on stream `s`: ordered behind the previous call's end when that
// ran on another stream; allocations outgrown by earlier calls are freed once that end has passed.
static int acquire_dev(Dev &d, hipStream_t s) {
    if (!d.retired.empty() && (!d.in_flight || hipEventQuery(d.ev_use) == hipSuccess)) d.free_retired();
    if (d.in_flight && d.last_stream != s) SHIP(hipStreamWaitEvent(s, d.ev_use, 0));
    return PN_OK;
}
struct DevUse {
    Dev &d;
    hipStream_t s;
    ~DevUse() {
        d.in_flight = hipEventRecord(d.ev_use, s) == hipSuccess;
        if (!d.in_flight) (void)hipStreamSynchronize(s);  // cannot mark the end of the call: wait for it instead
        d.last_stream = s;
    }
};

// rank mode (or one GPU): queries and results on this process's GPU, everything enqueued on `s`
template <typename T>
static int query_device_one(const pn_sharded *sh, Dev &d, const T *d_q, size_t nq, size_t q_cols, size_t q_stride,
                            size_t k, uint64_t *d_idx, T *d_dist, hipStream_t s) {
    const size_t k_out = (size_t)std::min<uint64_t>(k, sh->n_total), kp = k_part_of(sh, k), kd = k_dev_of(sh, k);
    SetGuard g(d.device);
    if (!g.ok) return set_error(PN_ERR_DEVICE, "hipSetDevice(%d) failed", d.device);
    if (sh->n_shards == 1 && !sh->exchange_always)  // one shard: nothing to exchange, straight into the caller's buffers
        return ShT<T>::query_strided(sh->parts[0].ix, d_q, nq, q_cols, q_stride, k, d_idx, d_dist, k_out, s);
    // the merges' LDS-free fallback serves any world x k (select.hip), so nothing here is bounded by k
    const size_t chunk = nq > kShardChunk ? kShardChunk : nq;
    const bool overlapped = nq > chunk;  // two chunks in flight: exchange + merge on the second stream
    SPN(acquire_dev(d, s));
    DevUse in_use{d, s};  // records the end-of-use event on every return path
    // PN_OPT_PROFILE, batches of one chunk: local half and exchange half between three events on the caller's stream
    int pslot = -1;
    if (sh->profile && !overlapped) {
        pslot = (int)(d.prof_next++ % Dev::kProf);
        prof_resolve(d, pslot, true);  // (a slot comes round again after kProf calls: long finished)
        for (hipEvent_t &e : d.ev_prof[pslot])
            if (!e) SHIP(hipEventCreate(&e));
        SHIP(hipEventRecord(d.ev_prof[pslot][0], s));
    }
    int set = 0;
    size_t n_chunks = 0;
    for (size_t q0 = 0; q0 < nq; q0 += chunk, set ^= 1, ++n_chunks) {
        const size_t nqc = nq - q0 < chunk ? nq - q0 : chunk;
        const size_t words = packed_words<T>(nqc, kd);
        SPN(d.gathered[set].ensure((size_t)sh->world * words * 8));
        // the packed buffer of this set is free again once the exchange that read it (two chunks ago) has been merged
        if (overlapped && n_chunks >= 2) SHIP(hipStreamWaitEvent(s, d.ev_merged[set], 0));
        SPN(enqueue_local<T>(sh, d, d_q + q0 * q_stride, nqc, q_cols, q_stride, k, kp, kd, set, s));
        hipStream_t xs = s;
        if (overlapped) {
            SHIP(hipEventRecord(d.ev_local[set], s));
            SHIP(hipStreamWaitEvent(d.comm_stream, d.ev_local[set], 0));
            xs = d.comm_stream;
        }
        if (pslot >= 0) SHIP(hipEventRecord(d.ev_prof[pslot][1], s));
        SNCCL(rccl().AllGather(d.pack[set].p, d.gathered[set].p, words, ncclUint64, d.comm, xs));
        SPN(enqueue_merge<T>(sh, d, nqc, kd, k_out, set, d_idx + q0 * k_out, d_dist + q0 * k_out, xs));
        if (pslot >= 0) {
            SHIP(hipEventRecord(d.ev_prof[pslot][2], s));
            d.prof_pending[pslot] = true;
        }
        if (overlapped) SHIP(hipEventRecord(d.ev_merged[set], xs));
    }
    if (overlapped) {  // results are ready in the order of the caller's stream
        SHIP(hipEventRecord(d.ev_done, d.comm_stream));
        SHIP(hipStreamWaitEvent(s, d.ev_done, 0));
    }
    return PN_OK;
}

template <typename T>
static int sharded_query_device(const pn_sharded *sh, const T *d_queries, size_t nq, size_t q_cols, size_t q_row_stride,
                                size_t k, uint64_t *d_idx_out, T *d_dist_out, void *stream) {
    if (!sh) return set_error(PN_ERR_INVALID, "handle is NULL");
    SPN(check_elem(sh, sizeof(T)));
    const size_t k_out = (size_t)std::min<uint64_t>(k, sh->n_total);
    if (nq == 0 || k_out == 0) return PN_OK;  // k == 0 -> empty result (src/ball_tree.rs:106-108)
    if (!d_queries && q_cols) return set_error(PN_ERR_INVALID, "queries is NULL");
    if (!d_idx_out || !d_dist_out) return set_error(PN_ERR_INVALID, "output buffer is NULL");
    if (sh->devs.size() != 1)
        return set_error(PN_ERR_UNSUPPORTED, "device-resident queries need a handle that drives ONE GPU (one process per "
                                             "GPU, or all shards on one device); use pn_sharded_query_f32");
    std::lock_guard<std::mutex> lk(sh->mu);
    return query_device_one<T>(sh, sh->devs[0], d_queries, nq, q_cols, q_row_stride, k, d_idx_out,
                               d_dist_out, (hipStream_t)stream);
}
extern "C" int pn_sharded_query_device_f32(const pn_sharded *sh, const float *d_queries, size_t nq, size_t q_cols,
                                           size_t q_row_stride, size_t k, uint64_t *d_idx_out, float *d_dist_out,
                                           void *stream) {
    return sharded_query_device<float>(sh, d_queries, nq, q_cols, q_row_stride, k, d_idx_out, d_dist_out, stream);
}
extern "C" int pn_sharded_query_device_f64(const pn_sharded *sh, const double *d_queries, size_t nq, size_t q_cols,
                                           size_t q_row_stride, size_t k, uint64_t *d_idx_out, double *d_dist_out,
                                           void *stream) {
    return sharded_query_device<double>(sh, d_queries, nq, q_cols, q_row_stride, k, d_idx_out, d_dist_out, stream);
}

// Host entry point: BallTree::query over the sharded corpus for a batch of host queries.
template <typename T>
static int sharded_query_host(const pn_sharded *sh, const T *queries, size_t nq, size_t q_cols, ptrdiff_t q_row_stride,
                              size_t k, uint64_t *idx_out, T *dist_out) {
    if (!sh) return set_error(PN_ERR_INVALID, "handle is NULL");
    SPN(check_elem(sh, sizeof(T)));
    const size_t k_out = (size_t)std::min<uint64_t>(k, sh->n_total), kp = k_part_of(sh, k), kd = k_dev_of(sh, k);
    if (nq == 0 || k_out == 0) return PN_OK;
    if (!queries && q_cols) return set_error(PN_ERR_INVALID, "queries is NULL");
    if (!idx_out || !dist_out) return set_error(PN_ERR_INVALID, "output buffer is NULL");
    if (q_row_stride < 0) return set_error(PN_ERR_UNSUPPORTED, "negative row stride");
    std::lock_guard<std::mutex> lk(sh->mu);
    std::vector<Dev> &devs = sh->devs;
    const size_t qc = q_cols ? q_cols : 1;
    // queries to every GPU this process drives (replicated)
    for (Dev &d : devs) {
        SetGuard g(d.device);
        if (!g.ok) return set_error(PN_ERR_DEVICE, "hipSetDevice(%d) failed", d.device);
        // behind whatever a device call on another stream still runs on these buffers; allocations outgrown by earlier
        // calls go back now (the previous host call ended with a stream sync: without this the multi-GPU path and the
        // one-shard early return, which never reach acquire_dev, kept every outgrown buffer until destroy -- ADVICE r3)
        SPN(acquire_dev(d, d.stream));
        SPN(d.q.ensure(nq * qc * sizeof(T)));
        if (q_cols) {
            if (nq == 1 || (size_t)q_row_stride == q_cols)
                SHIP(hipMemcpyAsync(d.q.p, queries, nq * q_cols * sizeof(T), hipMemcpyHostToDevice, d.stream));
            else
                SHIP(hipMemcpy2DAsync(d.q.p, q_cols * sizeof(T), queries, (size_t)q_row_stride * sizeof(T),
                                      q_cols * sizeof(T), nq, hipMemcpyHostToDevice, d.stream));
        }
    }
    Dev &d0 = devs[0];
    {
        SetGuard g(d0.device);
        SPN(d0.out_idx.ensure(nq * k_out * sizeof(uint64_t)));
        SPN(d0.out_dist.ensure(nq * k_out * sizeof(T)));
    }
    if (devs.size() == 1) {
        SPN(query_device_one<T>(sh, d0, (const T *)d0.q.p, nq, q_cols, qc, k, (uint64_t *)d0.out_idx.p,
                                (T *)d0.out_dist.p, d0.stream));
    } else {
        // one process, several GPUs: local work on every GPU, ONE grouped all-gather, merge on the first GPU
        const size_t words = packed_words<T>(nq, kd);
        for (Dev &d : devs) {
            SetGuard g(d.device);
            SPN(d.gathered[0].ensure((size_t)sh->world * words * 8));
            SPN(enqueue_local<T>(sh, d, (const T *)d.q.p, nq, q_cols, qc, k, kp, kd, 0, d.stream));
        }
        SNCCL(rccl().GroupStart());
        for (Dev &d : devs) {
            ncclResult_t r = rccl().AllGather(d.pack[0].p, d.gathered[0].p, words, ncclUint64, d.comm, d.stream);
            if (r != ncclSuccess) {
                (void)rccl().GroupEnd();
                return set_error(PN_ERR_COMM, "ncclAllGather: %s", rccl().GetErrorString(r));
            }
        }
        SNCCL(rccl().GroupEnd());
        SetGuard g(d0.device);
        SPN(enqueue_merge<T>(sh, d0, nq, kd, k_out, 0, (uint64_t *)d0.out_idx.p, (T *)d0.out_dist.p, d0.stream));
    }
    {
        SetGuard g(d0.device);
        SHIP(hipMemcpyAsync(idx_out, d0.out_idx.p, nq * k_out * sizeof(uint64_t), hipMemcpyDeviceToHost, d0.stream));
        SHIP(hipMemcpyAsync(dist_out, d0.out_dist.p, nq * k_out * sizeof(T), hipMemcpyDeviceToHost, d0.stream));
    }
    for (Dev &d : devs) {  // every GPU has left the collective before the call returns
        SetGuard g(d.device);
        SHIP(hipStreamSynchronize(d.stream));
    }
    return PN_OK;
}
extern "C" int pn_sharded_query_f32(const pn_sharded *sh, const float *queries, size_t nq, size_t q_cols,
                                    ptrdiff_t q_row_stride, size_t k, uint64_t *idx_out, float *dist_out) {
    return sharded_query_host<float>(sh, queries, nq, q_cols, q_row_stride, k, idx_out, dist_out);
}
extern "C" int pn_sharded_query_f64(const pn_sharded *sh, const double *queries, size_t nq, size_t q_cols,
                                    ptrdiff_t q_row_stride, size_t k, uint64_t *idx_out, double *dist_out) {
    return sharded_query_host<double>(sh, queries, nq, q_cols, q_row_stride, k, idx_out, dist_out);
}

// ---------------------------------------------------------------------------
// radius
// ---------------------------------------------------------------------------
// BallTree::query_radius over the sharded corpus: per query the ascending global rows of every shard, shard after
// shard -- shards are ascending row ranges, so the concatenation is ascending.  Local shards answer through
// pn_query_radius_f32; with one process per GPU the variable-length lists travel by two all-gathers (counts, then the
// lists padded to the longest) and are spliced on the host.
template <typename T>
static int sharded_query_radius(const pn_sharded *sh, const T *queries, size_t nq, size_t q_cols, ptrdiff_t q_row_stride,
                                T radius, uint64_t *offsets, uint64_t **idx_out) {
    if (!sh) return set_error(PN_ERR_INVALID, "handle is NULL");
    SPN(check_elem(sh, sizeof(T)));
    if (!offsets || !idx_out) return set_error(PN_ERR_INVALID, "output pointer is NULL");
    *idx_out = nullptr;
    offsets[0] = 0;
    if (nq == 0) return PN_OK;
    if (!queries && q_cols) return set_error(PN_ERR_INVALID, "queries is NULL");
    std::lock_guard<std::mutex> lk(sh->mu);
    if (sh->rank_mode && (sh->world > 1 || sh->exchange_always)) {
        // One process per GPU (round 4): the local shard answers through the DEVICE entry point (pn_query_radius_device_*:
        // counts, scan and fill in HBM), its CSR offsets are all-gathered as they lie in HBM, and the lists travel device
        // to device -- the host sees the gathered offsets once (to size the exchange) and the gathered lists once (to
        // splice).  Round 3 built a host CSR on every rank first (two stream synchronisations inside pn_query_radius_*)
        // and uploaded counts and lists again.
        Dev &d = sh->devs[0];
        SetGuard g(d.device);
        if (!g.ok) return set_error(PN_ERR_DEVICE, "hipSetDevice(%d) failed", d.device);
        SPN(acquire_dev(d, d.stream));
        const size_t W = (size_t)sh->world, qc = q_cols ? q_cols : 1;
        const pn_index *lix = sh->parts.empty() ? nullptr : sh->parts[0].ix;
        if (q_row_stride < 0) return set_error(PN_ERR_UNSUPPORTED, "negative row stride");
        SPN(d.q.ensure(nq * qc * sizeof(T)));
        if (q_cols) {
            if (nq == 1 || (size_t)q_row_stride == q_cols)
                SHIP(hipMemcpyAsync(d.q.p, queries, nq * q_cols * sizeof(T), hipMemcpyHostToDevice, d.stream));
            else
                SHIP(hipMemcpy2DAsync(d.q.p, q_cols * sizeof(T), queries, (size_t)q_row_stride * sizeof(T),
                                      q_cols * sizeof(T), nq, hipMemcpyHostToDevice, d.stream));
        }
        // rad_a: [local offsets nq + 1 | total 1 | pad 1 | gathered offsets W (nq + 1)]
        const size_t no = nq + 1;
        SPN(d.rad_a.ensure((no + 2 + W * no) * 8));
        uint64_t *d_off = (uint64_t *)d.rad_a.p, *d_tot = d_off + no, *d_all = d_off + no + 2;
        size_t cap = d.rad_b.bytes / 8 / (W + 1);      // this rank's share of what rad_b holds already
        if (cap < nq * 4 + 1024) cap = nq * 4 + 1024;
        std::vector<uint64_t> all_off(W * no);
        uint64_t longest = 1;
        for (int attempt = 0;; ++attempt) {
            SPN(d.rad_b.ensure((cap + W * cap) * 8));
            uint64_t *d_mine = (uint64_t *)d.rad_b.p;
            if (lix)
                SPN(ShT<T>::radius_device(lix, (const T *)d.q.p, nq, q_cols, qc, radius, d_off, d_mine, cap, d_tot, d.stream));
            else
                SHIP(hipMemsetAsync(d_off, 0, (no + 1) * 8, d.stream));  // a rank without rows: every list empty
            SNCCL(rccl().AllGather(d_off, d_all, no, ncclUint64, d.comm, d.stream));
            SHIP(hipMemcpyAsync(all_off.data(), d_all, W * no * 8, hipMemcpyDeviceToHost, d.stream));
            SHIP(hipStreamSynchronize(d.stream));
            longest = 1;
            for (size_t r = 0; r < W; ++r) longest = std::max<uint64_t>(longest, all_off[r * no + nq]);
            if (longest <= cap) break;  // every rank's list fits its buffer (all ranks see the same numbers: same decision)
            if (attempt) return set_error(PN_ERR_DEVICE, "radius lists changed size between two passes");
            cap = (size_t)longest;      // some rank overflowed: everybody re-runs with room for the longest list
        }
        uint64_t *d_mine = (uint64_t *)d.rad_b.p, *d_lists = d_mine + cap;
        SNCCL(rccl().AllGather(d_mine, d_lists, (size_t)longest, ncclUint64, d.comm, d.stream));
        std::vector<uint64_t> all_ids((size_t)(W * longest));
        SHIP(hipMemcpyAsync(all_ids.data(), d_lists, (size_t)(W * longest) * 8, hipMemcpyDeviceToHost, d.stream));
        SHIP(hipStreamSynchronize(d.stream));
        uint64_t total = 0;
        for (size_t r = 0; r < W; ++r) total += all_off[r * no + nq];
        uint64_t *res = (uint64_t *)malloc((total ? total : 1) * sizeof(uint64_t));
        if (!res) return set_error(PN_ERR_NOMEM, "malloc(%llu results) failed", (unsigned long long)total);
        uint64_t w = 0;
        for (size_t a = 0; a < nq; ++a) {  // per query, the ranks' (ascending) lists in rank order: shards are ascending row ranges
            offsets[a] = w;
            for (size_t r = 0; r < W; ++r) {
                const uint64_t lo = all_off[r * no + a], c = all_off[r * no + a + 1] - lo;
                if (c) memcpy(res + w, all_ids.data() + r * longest + lo, (size_t)c * sizeof(uint64_t));
                w += c;
            }
        }
        offsets[nq] = w;
        *idx_out = res;
        return PN_OK;
    }
    const size_t np = sh->parts.size();
    std::vector<std::vector<uint64_t>> offs(np, std::vector<uint64_t>(nq + 1, 0));
    std::vector<uint64_t *> lists(np, nullptr);
    struct Free {
        std::vector<uint64_t *> &l;
        ~Free() {
            for (uint64_t *p : l) pn_free(p);
        }
    } free_lists{lists};
    for (size_t i = 0; i < np; ++i)
        if (sh->parts[i].ix)
            SPN(ShT<T>::radius(sh->parts[i].ix, queries, nq, q_cols, q_row_stride, radius, offs[i].data(), &lists[i]));
    // local splice: per query, local shards in order
    std::vector<uint64_t> l_cnt(nq, 0);
    for (size_t i = 0; i < np; ++i)
        for (size_t a = 0; a < nq; ++a) l_cnt[a] += offs[i][a + 1] - offs[i][a];
    uint64_t l_total = 0;
    for (size_t a = 0; a < nq; ++a) l_total += l_cnt[a];
    std::vector<uint64_t> l_ids((size_t)l_total);
    {
        uint64_t w = 0;
        for (size_t a = 0; a < nq; ++a)
            for (size_t i = 0; i < np; ++i) {
                const uint64_t c = offs[i][a + 1] - offs[i][a];
                if (c) memcpy(l_ids.data() + w, lists[i] + offs[i][a], c * sizeof(uint64_t));
                w += c;
            }
    }
    {
        uint64_t *res = (uint64_t *)malloc((l_total ? l_total : 1) * sizeof(uint64_t));
        if (!res) return set_error(PN_ERR_NOMEM, "malloc(%llu results) failed", (unsigned long long)l_total);
        if (l_total) memcpy(res, l_ids.data(), (size_t)l_total * sizeof(uint64_t));
        uint64_t run = 0;
        for (size_t a = 0; a < nq; ++a) {
            offsets[a] = run;
            run += l_cnt[a];
        }
        offsets[nq] = run;
        *idx_out = res;
        return PN_OK;
    }
}
// query_radius with queries and CSR in HBM (pn_query_radius_device_*): a handle with ONE shard forwards to it (global row
// numbers through the shard's index base).  Several shards would need the ragged exchange on the device as well: the
// host entry point above serves them.
template <typename T>
static int sharded_query_radius_device(const pn_sharded *sh, const T *d_q, size_t nq, size_t q_cols, size_t q_stride, T radius,
                                       uint64_t *d_offsets, uint64_t *d_idx, size_t capacity, uint64_t *d_total, void *stream) {
    if (!sh) return set_error(PN_ERR_INVALID, "handle is NULL");
    SPN(check_elem(sh, sizeof(T)));
    if (sh->n_shards != 1 || sh->parts.size() != 1 || !sh->parts[0].ix)
        return set_error(PN_ERR_UNSUPPORTED, "the device-resident query_radius serves handles with one shard");
    return ShT<T>::radius_device(sh->parts[0].ix, d_q, nq, q_cols, q_stride, radius, d_offsets, d_idx, capacity, d_total, stream);
}
extern "C" int pn_sharded_query_radius_device_f32(const pn_sharded *sh, const float *d_q, size_t nq, size_t q_cols,
                                                  size_t q_stride, float radius, uint64_t *d_offsets, uint64_t *d_idx,
                                                  size_t capacity, uint64_t *d_total, void *stream) {
    return sharded_query_radius_device<float>(sh, d_q, nq, q_cols, q_stride, radius, d_offsets, d_idx, capacity, d_total, stream);
}
extern "C" int pn_sharded_query_radius_device_f64(const pn_sharded *sh, const double *d_q, size_t nq, size_t q_cols,
                                                  size_t q_stride, double radius, uint64_t *d_offsets, uint64_t *d_idx,
                                                  size_t capacity, uint64_t *d_total, void *stream) {
    return sharded_query_radius_device<double>(sh, d_q, nq, q_cols, q_stride, radius, d_offsets, d_idx, capacity, d_total, stream);
}
extern "C" int pn_sharded_query_radius_f32(const pn_sharded *sh, const float *queries, size_t nq, size_t q_cols,
                                           ptrdiff_t q_row_stride, float radius, uint64_t *offsets, uint64_t **idx_out) {
    return sharded_query_radius<float>(sh, queries, nq, q_cols, q_row_stride, radius, offsets, idx_out);
}
extern "C" int pn_sharded_query_radius_f64(const pn_sharded *sh, const double *queries, size_t nq, size_t q_cols,
                                           ptrdiff_t q_row_stride, double radius, uint64_t *offsets, uint64_t **idx_out) {
    return sharded_query_radius<double>(sh, queries, nq, q_cols, q_row_stride, radius, offsets, idx_out);
}
