// petal_neighbors.hpp -- header-only C++17 mirror of petal-neighbors' public API for the
// hot path (reference src/lib.rs:1-16, src/ball_tree.rs:15-374, src/distance.rs:9-74) over
// the C ABI in petal_mi355x.h.  Same names, argument meaning and error behaviour as the
// Rust types; results as std::vector like the Rust `Vec`s.  Link with -lpetal_mi355x.
#pragma once
#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "petal_mi355x.h"

namespace petal {

// ArrayError (src/lib.rs:9-16)
struct ArrayError : std::runtime_error {
    enum Kind { Empty, NotContiguous } kind;
    ArrayError(Kind k) : std::runtime_error(k == Empty ? "array is empty" : "array is not contiguous in memory"), kind(k) {}
};
struct DeviceError : std::runtime_error {
    int code;
    DeviceError(int c, const char *m) : std::runtime_error(m), code(c) {}
};

inline void check(int rc) {
    if (rc == PN_OK) return;
    if (rc == PN_ERR_EMPTY) throw ArrayError(ArrayError::Empty);
    if (rc == PN_ERR_NOT_CONTIGUOUS) throw ArrayError(ArrayError::NotContiguous);
    throw DeviceError(rc, pn_last_error());
}

namespace distance {
// Euclidean (src/distance.rs:16-55): zero-sized tag + Metric<A>
struct Euclidean {
    bool operator==(const Euclidean &) const { return true; }
    float distance(const float *a, const float *b, size_t len) const { return pn_euclidean_f32(a, b, len); }
    double distance(const double *a, const double *b, size_t len) const { return pn_euclidean_f64(a, b, len); }
    float rdistance(const float *a, const float *b, size_t len) const { return pn_reuclidean_f32(a, b, len); }
    double rdistance(const double *a, const double *b, size_t len) const { return pn_reuclidean_f64(a, b, len); }
    float rdistance_to_distance(float d) const { return pn_rdistance_to_distance_f32(d); }
    double rdistance_to_distance(double d) const { return pn_rdistance_to_distance_f64(d); }
    float distance_to_rdistance(float d) const { return pn_distance_to_rdistance_f32(d); }
    double distance_to_rdistance(double d) const { return pn_distance_to_rdistance_f64(d); }
};
// Cosine (src/distance.rs:76-122): 1 - dot / (|a| |b|); rdistance and the conversions are the identity
struct Cosine {
    bool operator==(const Cosine &) const { return true; }
    float distance(const float *a, const float *b, size_t len) const { return pn_cosine_f32(a, len, b, len); }
    double distance(const double *a, const double *b, size_t len) const { return pn_cosine_f64(a, len, b, len); }
    float rdistance(const float *a, const float *b, size_t len) const { return distance(a, b, len); }
    double rdistance(const double *a, const double *b, size_t len) const { return distance(a, b, len); }
    float rdistance_to_distance(float d) const { return d; }
    double rdistance_to_distance(double d) const { return d; }
    float distance_to_rdistance(float d) const { return d; }
    double distance_to_rdistance(double d) const { return d; }
};
// pairwise(x, &Cosine)
inline std::vector<float> pairwise(const float *x, size_t n, size_t d, const Cosine &, int device = 0) {
    std::vector<float> out(n * n);
    check(pn_pairwise_cosine_f32(x, n, d, (ptrdiff_t)d, device, out.data()));
    return out;
}
inline std::vector<double> pairwise(const double *x, size_t n, size_t d, const Cosine &, int device = 0) {
    std::vector<double> out(n * n);
    check(pn_pairwise_cosine_f64(x, n, d, (ptrdiff_t)d, device, out.data()));
    return out;
}
// pairwise (src/distance.rs:58-74): n x n row-major
inline std::vector<float> pairwise(const float *x, size_t n, size_t d, int device = 0) {
    std::vector<float> out(n * n);
    check(pn_pairwise_f32(x, n, d, (ptrdiff_t)d, device, out.data()));
    return out;
}
inline std::vector<double> pairwise(const double *x, size_t n, size_t d, int device = 0) {
    std::vector<double> out(n * n);
    check(pn_pairwise_f64(x, n, d, (ptrdiff_t)d, device, out.data()));
    return out;
}
}  // namespace distance

// BallTree<'a, A, M> (src/ball_tree.rs:15-24): M = distance::Euclidean (default) or distance::Cosine.  Under Cosine
// every query is an exact scan (cosine distance is not a metric; see pn_index_create_cosine_* in petal_mi355x.h).
template <typename A, typename M = distance::Euclidean>
class BallTree {
    static_assert(std::is_same<A, float>::value || std::is_same<A, double>::value, "A is f32 or f64");
    static_assert(std::is_same<M, distance::Euclidean>::value || std::is_same<M, distance::Cosine>::value,
                  "M is Euclidean or Cosine");
    static constexpr bool kF32 = std::is_same<A, float>::value;
    pn_index *h_ = nullptr;
    size_t n_ = 0, dim_ = 0;
    explicit BallTree(pn_index *h, size_t n, size_t d) : h_(h), n_(n), dim_(d) {}

  public:
    M metric;
    BallTree(BallTree &&o) noexcept : h_(o.h_), n_(o.n_), dim_(o.dim_) { o.h_ = nullptr; }
    BallTree(const BallTree &) = delete;
    ~BallTree() { pn_index_destroy(h_); }

    // BallTree::new(points, metric) (src/ball_tree.rs:38-63); strides in elements
    static BallTree create(const A *points, size_t rows, size_t cols, M = M{}, ptrdiff_t row_stride = -1,
                           ptrdiff_t col_stride = 1, int device = 0) {
        pn_index *h = nullptr;
        if (row_stride < 0) row_stride = (ptrdiff_t)cols;
        if constexpr (std::is_same<M, distance::Cosine>::value) {
            if constexpr (kF32)
                check(pn_index_create_cosine_f32(points, rows, cols, row_stride, col_stride, device, &h));
            else
                check(pn_index_create_cosine_f64(points, rows, cols, row_stride, col_stride, device, &h));
        } else {
            if constexpr (kF32)
                check(pn_index_create_f32(points, rows, cols, row_stride, col_stride, device, &h));
            else
                check(pn_index_create_f64(points, rows, cols, row_stride, col_stride, device, &h));
        }
        return BallTree(h, rows, cols);
    }

    // ---- tree introspection (src/ball_tree.rs:296-353); the tree is built on first use (petal_mi355x.h, pn_tree_*)
    size_t num_nodes() const {
        uint64_t v = 0;
        check(pn_tree_num_nodes(h_, &v));
        return (size_t)v;
    }
    // children_of(n) -> Option<(usize, usize)>: {false, ...} for a leaf
    struct Children { bool some; size_t left, right; };
    Children children_of(size_t n) const {
        int some = 0;
        uint64_t l = 0, r = 0;
        check(pn_tree_children_of(h_, n, &some, &l, &r));
        return Children{some != 0, (size_t)l, (size_t)r};
    }
    std::vector<size_t> points_of(size_t n) const {
        const uint64_t *p = nullptr;
        uint64_t c = 0;
        check(pn_tree_points_of(h_, n, &p, &c));
        return std::vector<size_t>(p, p + c);
    }
    A radius_of(size_t n) const {
        A v = 0;
        if constexpr (kF32) check(pn_tree_radius_of_f32(h_, n, &v)); else check(pn_tree_radius_of_f64(h_, n, &v));
        return v;
    }
    // compare_nodes(x, y) -> Option<Ordering>: -1 / 0 / 1, or 2 for None (a NaN radius)
    int compare_nodes(size_t x, size_t y) const {
        int o = 0;
        check(pn_tree_compare_nodes(h_, x, y, &o));
        return o;
    }
    A node_distance_lower_bound(size_t n1, size_t n2) const {
        A v = 0;
        if constexpr (kF32)
            check(pn_tree_node_distance_lower_bound_f32(h_, n1, n2, &v));
        else
            check(pn_tree_node_distance_lower_bound_f64(h_, n1, n2, &v));
        return v;
    }

    // BallTree::euclidean / ::new (src/ball_tree.rs:38-63, 367-373); strides in elements
    static BallTree euclidean(const A *points, size_t rows, size_t cols, ptrdiff_t row_stride = -1,
                              ptrdiff_t col_stride = 1, int device = 0) {
        static_assert(std::is_same<M, distance::Euclidean>::value, "euclidean() builds a BallTree<A, Euclidean>");
        return create(points, rows, cols, M{}, row_stride, col_stride, device);
    }
    size_t num_points() const { return n_; }  // src/ball_tree.rs:351

    // query (src/ball_tree.rs:102): ascending, min(k, n) results, k == 0 -> empty
    std::pair<std::vector<size_t>, std::vector<A>> query(const A *point, size_t len, size_t k) const {
        const size_t kout = k < n_ ? k : n_;
        std::vector<uint64_t> idx(kout);
        std::vector<A> dist(kout);
        if constexpr (std::is_same<A, float>::value)
            check(pn_query_f32(h_, point, 1, len, (ptrdiff_t)len, k, idx.data(), dist.data()));
        else
            check(pn_query_f64(h_, point, 1, len, (ptrdiff_t)len, k, idx.data(), dist.data()));
        return {std::vector<size_t>(idx.begin(), idx.end()), std::move(dist)};
    }
    // query_nearest (src/ball_tree.rs:80)
    std::pair<size_t, A> query_nearest(const A *point, size_t len) const {
        uint64_t i = 0;
        A d = 0;
        if constexpr (std::is_same<A, float>::value)
            check(pn_query_nearest_f32(h_, point, 1, len, (ptrdiff_t)len, &i, &d));
        else
            check(pn_query_nearest_f64(h_, point, 1, len, (ptrdiff_t)len, &i, &d));
        return {(size_t)i, d};
    }
    // query_radius (src/ball_tree.rs:137); ascending indices
    std::vector<size_t> query_radius(const A *point, size_t len, A distance) const {
        uint64_t off[2] = {0, 0};
        uint64_t *out = nullptr;
        if constexpr (std::is_same<A, float>::value)
            check(pn_query_radius_f32(h_, point, 1, len, (ptrdiff_t)len, distance, off, &out));
        else
            check(pn_query_radius_f64(h_, point, 1, len, (ptrdiff_t)len, distance, off, &out));
        std::vector<size_t> v(out, out + off[1]);
        pn_free(out);
        return v;
    }
    // extension: a batch of points per call (row-major queries), results nq x min(k, n)
    void query_batch(const A *queries, size_t nq, size_t len, size_t k, uint64_t *idx_out, A *dist_out) const {
        if constexpr (std::is_same<A, float>::value)
            check(pn_query_f32(h_, queries, nq, len, (ptrdiff_t)len, k, idx_out, dist_out));
        else
            check(pn_query_f64(h_, queries, nq, len, (ptrdiff_t)len, k, idx_out, dist_out));
    }
    pn_index *handle() const { return h_; }
};

// VantagePointTree (src/vantage_point_tree.rs:13-98): the reference's second index answers 1-NN only and returns the
// neighbour BallTree::query_nearest returns; on this engine both are the k = 1 case of the same exact scan.
template <typename A>
class VantagePointTree {
    BallTree<A> tree_;
    explicit VantagePointTree(BallTree<A> &&t) : tree_(std::move(t)) {}

  public:
    // VantagePointTree::euclidean / ::new (src/vantage_point_tree.rs:31-72): the same ArrayError cases as BallTree::new
    static VantagePointTree euclidean(const A *points, size_t rows, size_t cols, ptrdiff_t row_stride = -1,
                                      ptrdiff_t col_stride = 1, int device = 0) {
        return VantagePointTree(BallTree<A>::euclidean(points, rows, cols, row_stride, col_stride, device));
    }
    // query_nearest (src/vantage_point_tree.rs:88-98)
    std::pair<size_t, A> query_nearest(const A *point, size_t len) const { return tree_.query_nearest(point, len); }
};

// Row-sharded BallTree<A, Euclidean> over several GPUs driven by this process (petal_mi355x.h, pn_sharded_*): shard g of
// devices.size() lives on devices[g]; one RCCL all-gather per query batch; answers equal the single-index answers.
// A = float | double (the reference is generic over A, src/ball_tree.rs:26-30).
namespace detail {
inline int sh_create(const float *p, size_t r, size_t c, ptrdiff_t rs, ptrdiff_t cs, const int *d, int n, pn_sharded **o) {
    return pn_sharded_create_f32(p, r, c, rs, cs, d, n, o);
}
inline int sh_create(const double *p, size_t r, size_t c, ptrdiff_t rs, ptrdiff_t cs, const int *d, int n, pn_sharded **o) {
    return pn_sharded_create_f64(p, r, c, rs, cs, d, n, o);
}
inline int sh_query(const pn_sharded *h, const float *q, size_t nq, size_t len, size_t k, uint64_t *i, float *d) {
    return pn_sharded_query_f32(h, q, nq, len, (ptrdiff_t)len, k, i, d);
}
inline int sh_query(const pn_sharded *h, const double *q, size_t nq, size_t len, size_t k, uint64_t *i, double *d) {
    return pn_sharded_query_f64(h, q, nq, len, (ptrdiff_t)len, k, i, d);
}
inline int sh_radius(const pn_sharded *h, const float *q, size_t len, float r, uint64_t *off, uint64_t **out) {
    return pn_sharded_query_radius_f32(h, q, 1, len, (ptrdiff_t)len, r, off, out);
}
inline int sh_radius(const pn_sharded *h, const double *q, size_t len, double r, uint64_t *off, uint64_t **out) {
    return pn_sharded_query_radius_f64(h, q, 1, len, (ptrdiff_t)len, r, off, out);
}
}  // namespace detail
template <typename A>
class ShardedBallTreeT {
    pn_sharded *h_ = nullptr;
    size_t n_ = 0;

  public:
    ShardedBallTreeT(const A *points, size_t rows, size_t cols, const std::vector<int> &devices,
                     ptrdiff_t row_stride = -1, ptrdiff_t col_stride = 1)
        : n_(rows) {
        if (row_stride < 0) row_stride = (ptrdiff_t)cols;
        check(detail::sh_create(points, rows, cols, row_stride, col_stride, devices.data(), (int)devices.size(), &h_));
    }
    ShardedBallTreeT(const ShardedBallTreeT &) = delete;
    ~ShardedBallTreeT() { pn_sharded_destroy(h_); }
    size_t num_points() const { return n_; }
    std::pair<std::vector<size_t>, std::vector<A>> query(const A *point, size_t len, size_t k) const {
        const size_t kout = k < n_ ? k : n_;
        std::vector<uint64_t> idx(kout);
        std::vector<A> dist(kout);
        check(detail::sh_query(h_, point, 1, len, k, idx.data(), dist.data()));
        return {std::vector<size_t>(idx.begin(), idx.end()), std::move(dist)};
    }
    void query_batch(const A *queries, size_t nq, size_t len, size_t k, uint64_t *idx_out, A *dist_out) const {
        check(detail::sh_query(h_, queries, nq, len, k, idx_out, dist_out));
    }
    std::vector<size_t> query_radius(const A *point, size_t len, A distance) const {
        uint64_t off[2] = {0, 0};
        uint64_t *out = nullptr;
        check(detail::sh_radius(h_, point, len, distance, off, &out));
        std::vector<size_t> v(out, out + off[1]);
        pn_free(out);
        return v;
    }
};
using ShardedBallTree = ShardedBallTreeT<float>;

}  // namespace petal
