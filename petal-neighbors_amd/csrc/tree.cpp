// tree.cpp -- the ball tree itself, for the reference's tree-introspection API only
// (BallTree::{num_nodes, children_of, points_of, radius_of, compare_nodes, node_distance_lower_bound},
// reference src/ball_tree.rs:296-353, made public for downstream dual-tree algorithms).
//
// Queries never touch this: the MI355X path answers them by batched scans and no tree exists until one of the
// accessors is called.  Then the index builds -- once, on the host, from its own copy of the points -- the same
// implicit complete binary tree the reference builds (src/ball_tree.rs:38-63, 445-461, 504-613): children of node i
// are 2i+1 and 2i+2, every node holds a range of the permutation `idx`, its centroid (sequential mean) and radius
// (largest distance to the centroid under the tree's OWN metric -- Euclidean or Cosine, Node::init takes `metric`), split at the median of the column of maximum spread by a Lomuto quick-select
// whose pivot is the range's last element.  Same arithmetic, same order, same tie rules, so node for node the
// answers equal the reference's.  Compiled with -ffp-contract=off (no fused multiply-add anywhere: the reference's
// folds are separately rounded, src/distance.rs:26-35).
//
// This is product code and shares nothing with oracle/ (test infrastructure); tests compare the two node by node.
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <new>
#include <vector>

#include "host_tree.h"

namespace pn {

namespace {

template <typename T> inline T t_sqrt(T x);
template <> inline float t_sqrt<float>(float x) { return sqrtf(x); }
template <> inline double t_sqrt<double>(double x) { return sqrt(x); }
// num-traits FloatCore::{min, max}: a NaN operand yields the other operand (fmin/fmax semantics)
template <typename T> inline T t_min(T a, T b);
template <> inline float t_min<float>(float a, float b) { return fminf(a, b); }
template <> inline double t_min<double>(double a, double b) { return fmin(a, b); }
template <typename T> inline T t_max(T a, T b);
template <> inline float t_max<float>(float a, float b) { return fmaxf(a, b); }
template <> inline double t_max<double>(double a, double b) { return fmax(a, b); }

// Euclidean::distance (src/distance.rs:26-35): sub, mul, add separately rounded, ascending coordinate, then sqrt
template <typename T>
inline T euclid(const T *a, const T *b, size_t dim) {
    T s = (T)0;
    for (size_t k = 0; k < dim; ++k) {
        const T d = a[k] - b[k];
        const T sq = d * d;
        s = s + sq;
    }
    return t_sqrt<T>(s);
}

// Cosine::distance (src/distance.rs:86-107): three sequential sums (products and additions separately rounded),
// 1 - dot / (sqrt(sum a^2) * sqrt(sum b^2)); both vectors have the tree's dimension here
template <typename T>
inline T cosine(const T *a, const T *b, size_t dim) {
    T dot = (T)0, n1 = (T)0, n2 = (T)0;
    for (size_t k = 0; k < dim; ++k) dot = dot + a[k] * b[k];
    for (size_t k = 0; k < dim; ++k) n1 = n1 + a[k] * a[k];
    for (size_t k = 0; k < dim; ++k) n2 = n2 + b[k] * b[k];
    return (T)1 - dot / (t_sqrt<T>(n1) * t_sqrt<T>(n2));
}

template <typename T>
struct Tree {
    size_t n = 0, dim = 0;
    int metric = 0;  // BallTree::metric (src/ball_tree.rs:23): 0 Euclidean, 1 Cosine
    // metric.distance: Node::init and node_distance_lower_bound go through it (src/ball_tree.rs:309, 459)
    T dist(const T *a, const T *b) const { return metric == 1 ? cosine<T>(a, b, dim) : euclid<T>(a, b, dim); }
    const T *pts = nullptr;  // [n][dim], owned by `store`
    std::vector<T> store;
    std::vector<uint64_t> idx;                // the permutation (BallTree::idx, src/ball_tree.rs:21)
    std::vector<uint64_t> start, end;         // Node::range
    std::vector<T> radius;                    // Node::radius
    std::vector<unsigned char> leaf;          // Node::is_leaf
    std::vector<T> centroid;                  // Node::centroid, [nodes][dim]

    const T *row(uint64_t i) const { return pts + (size_t)i * dim; }

    // Node::init (src/ball_tree.rs:445-461)
    void init_node(size_t node, size_t s, size_t e) {
        T *c = centroid.data() + node * dim;
        for (size_t k = 0; k < dim; ++k) c[k] = (T)0;
        for (size_t j = s; j < e; ++j) {
            const T *r = row(idx[j]);
            for (size_t k = 0; k < dim; ++k) c[k] += r[k];
        }
        const T len = (T)(e - s);
        for (size_t k = 0; k < dim; ++k) c[k] /= len;
        T mx = (T)0;
        for (size_t j = s; j < e; ++j) mx = t_max<T>(dist(c, row(idx[j])), mx);
        radius[node] = mx;
        start[node] = s;
        end[node] = e;
    }

    // max_spread_column (src/ball_tree.rs:577-613): the first column wins ties, a NaN spread never wins
    size_t max_spread_column(size_t s, size_t e) const {
        size_t best = 0;
        T best_spread = (T)0;
        for (size_t col = 0; col < dim; ++col) {
            T mn = row(idx[s])[col], mx = mn;
            for (size_t j = s + 1; j < e; ++j) {
                const T v = row(idx[j])[col];
                mn = t_min<T>(mn, v);
                mx = t_max<T>(mx, v);
            }
            const T spread = mx - mn;
            if (col == 0) {
                best_spread = spread;
            } else if (spread > best_spread) {  // partial_cmp == Some(Greater)
                best = col;
                best_spread = spread;
            }
        }
        return best;
    }

    // halve_node_indices (src/ball_tree.rs:545-569): quick-select of the median position, pivot = last element
    void halve(size_t s, size_t e, size_t col) {
        uint64_t *ix = idx.data() + s;
        const size_t len = e - s;
        size_t first = 0, last = len - 1;
        const size_t mid = len / 2;
        for (;;) {
            size_t cur = first;
            const T pivot = row(ix[last])[col];
            for (size_t i = first; i < last; ++i) {
                if (row(ix[i])[col] < pivot) {
                    const uint64_t t = ix[i];
                    ix[i] = ix[cur];
                    ix[cur] = t;
                    ++cur;
                }
            }
            const uint64_t t = ix[cur];
            ix[cur] = ix[last];
            ix[last] = t;
            if (cur == mid) break;
            if (cur < mid)
                first = cur + 1;
            else
                last = cur - 1;
        }
    }

    // build_subtree (src/ball_tree.rs:504-538); depth = height <= 64
    void build(size_t node, size_t s, size_t e) {
        init_node(node, s, e);
        const size_t left = node * 2 + 1;
        if (left >= start.size()) {
            leaf[node] = 1;
            return;
        }
        halve(s, e, max_spread_column(s, e));
        const size_t mid = (s + e) / 2;
        build(left, s, mid);
        build(left + 1, mid, e);
    }
};

}  // namespace

struct HostTree {
    int elem_bytes;
    Tree<float> f;
    Tree<double> d;
};

template <typename T>
static bool build_into(Tree<T> &t, const void *pts, size_t n, size_t dim, int metric) {
    t.n = n;
    t.dim = dim;
    t.metric = metric;
    const T *p = static_cast<const T *>(pts);
    t.store.assign(p, p + n * dim);
    t.pts = t.store.data();
    // height = BITS - leading_zeros(n); size = 2^height - 1 (src/ball_tree.rs:51-52)
    size_t height = 0;
    for (size_t v = n; v; v >>= 1) ++height;
    const size_t size = ((size_t)1 << height) - 1;
    t.idx.resize(n);
    for (size_t i = 0; i < n; ++i) t.idx[i] = i;
    t.start.assign(size, 0);
    t.end.assign(size, 0);
    t.radius.assign(size, (T)0);
    t.leaf.assign(size, 0);
    t.centroid.assign(size * dim, (T)0);
    t.build(0, 0, n);
    return true;
}

HostTree *host_tree_build(const void *pts, size_t n, size_t dim, int elem_bytes, int metric) {
    if (n == 0 || (dim == 0 && n >= 2)) return nullptr;  // rejected at index creation already
    HostTree *h = new (std::nothrow) HostTree();
    if (!h) return nullptr;
    h->elem_bytes = elem_bytes;
    try {
        if (elem_bytes == 4)
            build_into<float>(h->f, pts, n, dim, metric);
        else
            build_into<double>(h->d, pts, n, dim, metric);
    } catch (const std::bad_alloc &) {
        delete h;
        return nullptr;
    }
    return h;
}
void host_tree_free(HostTree *h) { delete h; }

#define PN_TREE_DISPATCH(expr_f, expr_d) (h->elem_bytes == 4 ? (expr_f) : (expr_d))

size_t host_tree_num_nodes(const HostTree *h) { return PN_TREE_DISPATCH(h->f.start.size(), h->d.start.size()); }
const uint64_t *host_tree_idx(const HostTree *h) { return PN_TREE_DISPATCH(h->f.idx.data(), h->d.idx.data()); }
void host_tree_node(const HostTree *h, size_t node, uint64_t *start, uint64_t *end, int *is_leaf) {
    if (h->elem_bytes == 4) {
        *start = h->f.start[node];
        *end = h->f.end[node];
        *is_leaf = h->f.leaf[node];
    } else {
        *start = h->d.start[node];
        *end = h->d.end[node];
        *is_leaf = h->d.leaf[node];
    }
}
// the f32 values are returned widened to double (exact); the caller narrows them back (exact)
double host_tree_radius(const HostTree *h, size_t node) {
    return PN_TREE_DISPATCH((double)h->f.radius[node], h->d.radius[node]);
}
const void *host_tree_centroid(const HostTree *h, size_t node) {
    return h->elem_bytes == 4 ? static_cast<const void *>(h->f.centroid.data() + node * h->f.dim)
                              : static_cast<const void *>(h->d.centroid.data() + node * h->d.dim);
}
// compare_nodes (src/ball_tree.rs:340-343): radius.partial_cmp -> -1 Less, 0 Equal, 1 Greater, 2 None (a NaN radius)
int host_tree_compare(const HostTree *h, size_t x, size_t y) {
    const double a = host_tree_radius(h, x), b = host_tree_radius(h, y);
    if (a < b) return -1;
    if (a > b) return 1;
    if (a == b) return 0;
    return 2;
}
// node_distance_lower_bound (src/ball_tree.rs:303-318): max(metric.distance(c1, c2) - R1 - R2, 0), evaluated left to right
// in T (`lb < 0 ? 0 : lb`: a NaN stays NaN, as in the reference)
double host_tree_lower_bound(const HostTree *h, size_t n1, size_t n2) {
    if (h->elem_bytes == 4) {
        const Tree<float> &t = h->f;
        const float lb = t.dist(t.centroid.data() + n1 * t.dim, t.centroid.data() + n2 * t.dim) - t.radius[n1] - t.radius[n2];
        return (double)(lb < 0.0f ? 0.0f : lb);
    }
    const Tree<double> &t = h->d;
    const double lb = t.dist(t.centroid.data() + n1 * t.dim, t.centroid.data() + n2 * t.dim) - t.radius[n1] - t.radius[n2];
    return lb < 0.0 ? 0.0 : lb;
}

}  // namespace pn
