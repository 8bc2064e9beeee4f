/*
 * oracle_impl.h -- type-generic body of the CPU oracle (TEST INFRASTRUCTURE ONLY).
 *
 * Included twice by oracle.c, once with T=float / SFX=f32 and once with
 * T=double / SFX=f64.  Every function is a restatement, in plain C, of the
 * petal-neighbors 0.18.0 algorithm it cites (paths relative to /root/reference).
 * Arithmetic is never contracted or re-associated: the translation unit is
 * built with -O2 -ffp-contract=off and no fast-math (see Makefile).
 *
 * Nothing outside tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may call into this file.
 */

#define CAT_(a, b) a##_##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SFX)

/* ---- ordered-float total order (src/ball_tree.rs:396-421 compares
 * OrderedFloat<A>; ordered-float 5: NaN == NaN, NaN > everything else). ---- */
static inline int FN(of_cmp)(T a, T b)
{
    if (a != a) return (b != b) ? 0 : 1;
    if (b != b) return -1;
    return (a < b) ? -1 : (a > b) ? 1 : 0;
}

/* ---- src/distance.rs:37-45  Euclidean::rdistance: sequential fold
 * sum += (v1 - v2) * (v1 - v2), sum0 = 0, index order, mul and add rounded
 * separately (Rust never contracts). ---- */
T FN(oracle_reuclidean)(const T *x1, const T *x2, size_t dim)
{
    T sum = (T)0;
    for (size_t i = 0; i < dim; ++i) {
        T diff = x1[i] - x2[i];
        sum += diff * diff;
    }
    return sum;
}

/* ---- src/distance.rs:26-35  Euclidean::distance = sqrt(fold). ---- */
T FN(oracle_euclidean)(const T *x1, const T *x2, size_t dim)
{
    return SQRT(FN(oracle_reuclidean)(x1, x2, dim));
}

/* src/distance.rs:47-49 */
T FN(oracle_rdistance_to_distance)(T d) { return SQRT(d); }
/* src/distance.rs:52-54  d.powi(2) == d*d (one rounding) */
T FN(oracle_distance_to_rdistance)(T d) { return d * d; }

/* ---- src/distance.rs:85-107  Cosine::distance = 1 - dot / (sqrt(sum x1^2) * sqrt(sum x2^2)); each of the
 * three sums is a sequential iterator sum in index order (products and additions rounded separately).
 * rdistance and both conversions are the identity (src/distance.rs:109-121). ---- */
T FN(oracle_cosine)(const T *x1, size_t len1, const T *x2, size_t len2)
{
    /* the dot product zips the two vectors (shorter length); each norm zips a vector with itself (its own length) */
    const size_t lz = len1 < len2 ? len1 : len2;
    T dot = (T)0, n1 = (T)0, n2 = (T)0;
    for (size_t i = 0; i < lz; ++i) dot += x1[i] * x2[i];
    for (size_t i = 0; i < len1; ++i) n1 += x1[i] * x1[i];
    for (size_t i = 0; i < len2; ++i) n2 += x2[i] * x2[i];
    return (T)1 - dot / (SQRT(n1) * SQRT(n2));
}

/* src/distance.rs:58-74 with metric = Cosine */
void FN(oracle_pairwise_cosine)(const T *x, size_t n, size_t dim, size_t ld, T *out)
{
    for (size_t i = 0; i < n * n; ++i) out[i] = (T)0;
    if (n < 2) return;
    for (size_t i = 0; i < n; ++i)
        for (size_t j = i + 1; j < n; ++j) {
            T d = FN(oracle_cosine)(x + i * ld, dim, x + j * ld, dim);
            out[i * n + j] = d;
            out[j * n + i] = d;
        }
}

/* ---- src/distance.rs:58-74  pairwise: zero matrix, i<j filled + mirrored;
 * n < 2 -> zeros. ---- */
void FN(oracle_pairwise)(const T *x, size_t n, size_t dim, size_t ld, T *out)
{
    for (size_t i = 0; i < n * n; ++i) out[i] = (T)0;
    if (n < 2) return;
    for (size_t i = 0; i < n; ++i)
        for (size_t j = i + 1; j < n; ++j) {
            T d = FN(oracle_euclidean)(x + i * ld, x + j * ld, dim);
            out[i * n + j] = d;
            out[j * n + i] = d;
        }
}

/* ======================================================================
 * Canonical brute force: the SPECIFICATION of the result:
 * the k smallest of {distance(q, p_i)} under the OrderedFloat total order,
 * ties inside equal-distance groups ordered by ascending index.
 * Same shape as the reference's own test oracle naive_k_nearest_neighbors
 * (src/ball_tree.rs:873-894: distance to every row, sort, take k).
 * ====================================================================== */
typedef struct { T d; size_t i; } FN(pair_t);

static int FN(pair_cmp)(const void *a, const void *b)
{
    const FN(pair_t) *x = (const FN(pair_t) *)a, *y = (const FN(pair_t) *)b;
    int c = FN(of_cmp)(x->d, y->d);
    if (c) return c;
    return (x->i < y->i) ? -1 : (x->i > y->i) ? 1 : 0;
}

/* returns number of results per query = min(k, n); outputs are nq x kout */
size_t FN(oracle_brute_knn)(const T *pts, size_t n, size_t dim, size_t ld,
                            const T *q, size_t nq, size_t qld, size_t k,
                            uint64_t *idx_out, T *dist_out)
{
    size_t kout = k < n ? k : n;
    if (kout == 0) return 0;
    FN(pair_t) *buf = (FN(pair_t) *)malloc(n * sizeof(*buf));
    for (size_t a = 0; a < nq; ++a) {
        for (size_t i = 0; i < n; ++i) {
            buf[i].d = FN(oracle_euclidean)(q + a * qld, pts + i * ld, dim);
            buf[i].i = i;
        }
        qsort(buf, n, sizeof(*buf), FN(pair_cmp));
        for (size_t j = 0; j < kout; ++j) {
            idx_out[a * kout + j] = buf[j].i;
            dist_out[a * kout + j] = buf[j].d;
        }
    }
    free(buf);
    return kout;
}

/* Canonical radius set { i : distance(q, p_i) < r }, ascending index
 * (leaf test of src/ball_tree.rs:275-282 is strict '<').
 * Two-call protocol: counts first (out == NULL), then fill. */
size_t FN(oracle_brute_radius)(const T *pts, size_t n, size_t dim, size_t ld,
                               const T *q, T r, uint64_t *out)
{
    size_t c = 0;
    for (size_t i = 0; i < n; ++i) {
        T d = FN(oracle_euclidean)(q, pts + i * ld, dim);
        if (d < r) {
            if (out) out[c] = i;
            ++c;
        }
    }
    return c;
}

/* ======================================================================
 * Faithful ball tree (src/ball_tree.rs).  Used (a) to cross-check the
 * canonical brute force, (b) as the CPU baseline in bench.py.
 * ====================================================================== */
typedef struct {
    size_t start, end; /* Node::range  (src/ball_tree.rs:428) */
    T *centroid;       /* one allocation per node, as Array1 (:429) */
    T radius;
    int is_leaf;
} FN(node_t);

typedef struct {
    const T *pts;
    size_t n, dim, ld;
    size_t *idx;      /* BallTree::idx (:21) */
    FN(node_t) *nodes; /* BallTree::nodes (:22) */
    size_t n_nodes;
    /* instrumentation (not in the reference) */
    uint64_t n_centroid_evals, n_point_evals;
    int par_depth; /* 0 = build on one thread like the reference */
    int metric;    /* BallTree::metric (:23): 0 = Euclidean, 1 = Cosine */
} FN(tree_t);

/* metric.distance(a, b) for the tree's metric tag (src/ball_tree.rs:165,218,276,309,459,464,474 all go through
 * Metric::distance); both vectors have the tree's dimension */
static inline T FN(metric_distance)(int metric, const T *a, const T *b, size_t dim)
{
    return metric == 1 ? FN(oracle_cosine)(a, dim, b, dim) : FN(oracle_euclidean)(a, b, dim);
}

/* src/ball_tree.rs:445-461  Node::init */
void FN(oracle_node_init_metric)(const T *pts, size_t dim, size_t ld, const size_t *idx,
                                 size_t len, T *centroid, T *radius, int metric)
{
    for (size_t c = 0; c < dim; ++c) centroid[c] = (T)0;
    for (size_t a = 0; a < len; ++a) { /* :446-453 row-by-row accumulation */
        const T *row = pts + idx[a] * ld;
        for (size_t c = 0; c < dim; ++c) centroid[c] += row[c];
    }
    T flen = (T)len; /* A::from_usize (:454) */
    for (size_t c = 0; c < dim; ++c) centroid[c] /= flen;
    T mx = (T)0; /* :458-460 fold(0, |max, i| A::max(dist, max)) */
    for (size_t a = 0; a < len; ++a) {
        T d = FN(metric_distance)(metric, centroid, pts + idx[a] * ld, dim); /* :459 metric.distance(centroid, row) */
        mx = FMAX(d, mx); /* Float::max ignores NaN like fmax */
    }
    *radius = mx;
}
void FN(oracle_node_init)(const T *pts, size_t dim, size_t ld, const size_t *idx,
                          size_t len, T *centroid, T *radius)
{
    FN(oracle_node_init_metric)(pts, dim, ld, idx, len, centroid, radius, 0);
}

/* src/ball_tree.rs:577-613  max_spread_column; returns (size_t)-1 on the
 * reference's "empty matrix" panic condition. */
size_t FN(oracle_max_spread_column)(const T *pts, size_t dim, size_t ld,
                                    const size_t *idx, size_t len)
{
    if (dim == 0 || len == 0) return (size_t)-1; /* assert :582 */
    size_t best = 0;
    T best_spread = (T)0;
    for (size_t c = 0; c < dim; ++c) {
        T mn = pts[idx[0] * ld + c], mx = mn; /* :595 */
        for (size_t a = 1; a < len; ++a) {
            T v = pts[idx[a] * ld + c];
            mn = FMIN(mn, v); /* A::min / A::max (:596) */
            mx = FMAX(mx, v);
        }
        T spread = mx - mn;
        if (c == 0) {
            best_spread = spread; /* :601 */
        } else if (spread > best_spread) {
            /* partial_cmp == Some(Greater) (:605): false for NaN on either side */
            best = c;
            best_spread = spread;
        }
    }
    return best;
}

/* src/ball_tree.rs:545-569  halve_node_indices (Lomuto quick-select, last
 * element is the pivot).  len must be >= 1 (the reference underflows on 0). */
void FN(oracle_halve_node_indices)(size_t *idx, size_t len, const T *col, size_t cstride)
{
    size_t first = 0, last = len - 1;
    size_t mid = len / 2;
    for (;;) {
        size_t cur = first;
        for (size_t i = first; i < last; ++i) {
            if (col[idx[i] * cstride] < col[idx[last] * cstride]) {
                size_t t = idx[i]; idx[i] = idx[cur]; idx[cur] = t;
                ++cur;
            }
        }
        { size_t t = idx[cur]; idx[cur] = idx[last]; idx[last] = t; }
        if (cur == mid) break;
        if (cur < mid) first = cur + 1;
        else last = cur - 1;
    }
}

typedef struct { FN(tree_t) *t; size_t root, start, end; } FN(bjob_t);
static void *FN(bjob_run)(void *p);

/* src/ball_tree.rs:504-538  build_subtree */
static void FN(build_subtree)(FN(tree_t) *t, size_t root, size_t start, size_t end)
{
    FN(node_t) *nd = &t->nodes[root];
    nd->centroid = (T *)malloc((t->dim ? t->dim : 1) * sizeof(T));
    FN(oracle_node_init_metric)(t->pts, t->dim, t->ld, t->idx + start, end - start,
                                nd->centroid, &nd->radius, t->metric);
    nd->start = start;
    nd->end = end;
    size_t left = root * 2 + 1;
    if (left >= t->n_nodes) { /* :524-527 */
        nd->is_leaf = 1;
        return;
    }
    size_t col = FN(oracle_max_spread_column)(t->pts, t->dim, t->ld, t->idx + start, end - start);
    FN(oracle_halve_node_indices)(t->idx + start, end - start, t->pts + col, t->ld);
    size_t mid = (start + end) / 2; /* :535 */
    /* The two recursive calls touch disjoint node and idx ranges, so the top
     * levels may run on separate threads without changing any result (test
     * infrastructure speed-up only; the reference builds on one thread). */
    if (t->par_depth > 0 && root < ((size_t)1 << t->par_depth) - 1) {
        FN(bjob_t) job = { t, left, start, mid };
        pthread_t th;
        if (pthread_create(&th, NULL, FN(bjob_run), &job) == 0) {
            FN(build_subtree)(t, left + 1, mid, end);
            pthread_join(th, NULL);
            return;
        }
    }
    FN(build_subtree)(t, left, start, mid);
    FN(build_subtree)(t, left + 1, mid, end);
}
static void *FN(bjob_run)(void *p)
{
    FN(bjob_t) *j = (FN(bjob_t) *)p;
    FN(build_subtree)(j->t, j->root, j->start, j->end);
    return NULL;
}

/* src/ball_tree.rs:38-63  BallTree::new.  Returns NULL with *err = 1 (Empty)
 * or 2 (NotContiguous: inner stride != 1), mirroring ArrayError (src/lib.rs:9-16).
 * dim == 0 with n >= 2 is the reference's "empty matrix" panic -> *err = 3. */
FN(tree_t) *FN(oracle_tree_build_mt)(const T *pts, size_t n, size_t dim, size_t ld,
                                     ptrdiff_t col_stride, int par_depth, int *err);
FN(tree_t) *FN(oracle_tree_build)(const T *pts, size_t n, size_t dim, size_t ld,
                                  ptrdiff_t col_stride, int *err)
{
    return FN(oracle_tree_build_mt)(pts, n, dim, ld, col_stride, 0, err);
}
FN(tree_t) *FN(oracle_tree_build_metric)(const T *pts, size_t n, size_t dim, size_t ld,
                                         ptrdiff_t col_stride, int par_depth, int metric, int *err);
FN(tree_t) *FN(oracle_tree_build_mt)(const T *pts, size_t n, size_t dim, size_t ld,
                                     ptrdiff_t col_stride, int par_depth, int *err)
{
    return FN(oracle_tree_build_metric)(pts, n, dim, ld, col_stride, par_depth, 0, err);
}
/* BallTree::new(points, metric) (src/ball_tree.rs:38-63): metric 0 = Euclidean, 1 = Cosine */
FN(tree_t) *FN(oracle_tree_build_metric)(const T *pts, size_t n, size_t dim, size_t ld,
                                         ptrdiff_t col_stride, int par_depth, int metric, int *err)
{
    *err = 0;
    if (n == 0) { *err = 1; return NULL; }                       /* :44-46 */
    if (dim > 1 && col_stride != 1) { *err = 2; return NULL; }   /* :47-49 */
    if (dim == 0 && n >= 2) { *err = 3; return NULL; }           /* :582 panic */
    FN(tree_t) *t = (FN(tree_t) *)calloc(1, sizeof(*t));
    t->pts = pts; t->n = n; t->dim = dim; t->ld = ld;
    t->par_depth = par_depth;
    t->metric = metric;
    unsigned height = 0; /* usize::BITS - leading_zeros(n) (:51) */
    for (size_t v = n; v; v >>= 1) ++height;
    t->n_nodes = ((size_t)1 << height) - 1; /* :52 */
    t->idx = (size_t *)malloc(n * sizeof(size_t));
    for (size_t i = 0; i < n; ++i) t->idx[i] = i; /* :54 */
    t->nodes = (FN(node_t) *)calloc(t->n_nodes, sizeof(FN(node_t))); /* Node::default :484-497 */
    FN(build_subtree)(t, 0, 0, n);
    return t;
}

void FN(oracle_tree_free)(FN(tree_t) *t)
{
    if (!t) return;
    for (size_t i = 0; i < t->n_nodes; ++i) free(t->nodes[i].centroid);
    free(t->nodes);
    free(t->idx);
    free(t);
}

size_t FN(oracle_tree_num_nodes)(const FN(tree_t) *t) { return t->n_nodes; }
const size_t *FN(oracle_tree_idx)(const FN(tree_t) *t) { return t->idx; }
void FN(oracle_tree_node)(const FN(tree_t) *t, size_t i, size_t *start, size_t *end,
                          T *radius, int *is_leaf, T *centroid_out)
{
    const FN(node_t) *nd = &t->nodes[i];
    *start = nd->start; *end = nd->end; *radius = nd->radius; *is_leaf = nd->is_leaf;
    if (centroid_out && nd->centroid)
        for (size_t c = 0; c < t->dim; ++c) centroid_out[c] = nd->centroid[c];
}
void FN(oracle_tree_eval_counts)(FN(tree_t) *t, uint64_t *centroid, uint64_t *point, int reset)
{
    *centroid = t->n_centroid_evals; *point = t->n_point_evals;
    if (reset) t->n_centroid_evals = t->n_point_evals = 0;
}

/* src/ball_tree.rs:473-481  Node::distance_lower_bound */
static inline T FN(node_lb)(const FN(tree_t) *t, const FN(node_t) *nd, const T *q, uint64_t *cnt)
{
    T cd = FN(metric_distance)(t->metric, q, nd->centroid, t->dim);
    ++*cnt;
    T lb = cd - nd->radius;
    return (lb < (T)0) ? (T)0 : lb;
}

/* ---- Rust std BinaryHeap<Neighbor> emulation [recalled: std source is not
 * under /root/reference; tie order between equal distances is therefore
 * UNPINNED -- SURVEY.md A.3]. Max-heap on distance only. ---- */
typedef struct { size_t idx; T d; } FN(nb_t);
typedef struct { FN(nb_t) *v; size_t len; } FN(heap_t);

#define NB_LE(a, b) (FN(of_cmp)((a).d, (b).d) <= 0)
#define NB_LT(a, b) (FN(of_cmp)((a).d, (b).d) < 0)
#define NB_GE(a, b) (FN(of_cmp)((a).d, (b).d) >= 0)

static void FN(heap_sift_up)(FN(heap_t) *h, size_t start, size_t pos)
{
    FN(nb_t) e = h->v[pos];
    while (pos > start) {
        size_t parent = (pos - 1) / 2;
        if (NB_LE(e, h->v[parent])) break;
        h->v[pos] = h->v[parent];
        pos = parent;
    }
    h->v[pos] = e;
}
static void FN(heap_push)(FN(heap_t) *h, FN(nb_t) e)
{
    h->v[h->len] = e;
    ++h->len;
    FN(heap_sift_up)(h, 0, h->len - 1);
}
static void FN(heap_sift_down_to_bottom)(FN(heap_t) *h, size_t pos)
{
    size_t end = h->len, start = pos;
    FN(nb_t) e = h->v[pos];
    size_t child = 2 * pos + 1;
    while (end >= 2 && child <= end - 2) {
        if (NB_LE(h->v[child], h->v[child + 1])) ++child;
        h->v[pos] = h->v[child];
        pos = child;
        child = 2 * pos + 1;
    }
    if (child == end - 1) {
        h->v[pos] = h->v[child];
        pos = child;
    }
    h->v[pos] = e;
    FN(heap_sift_up)(h, start, pos);
}
static void FN(heap_pop)(FN(heap_t) *h)
{
    FN(nb_t) item = h->v[h->len - 1];
    --h->len;
    if (h->len) {
        h->v[0] = item; /* swap(item, data[0]); the old root is dropped */
        FN(heap_sift_down_to_bottom)(h, 0);
    }
}
static void FN(heap_sift_down_range)(FN(nb_t) *v, size_t pos, size_t end)
{
    FN(nb_t) e = v[pos];
    size_t child = 2 * pos + 1;
    while (end >= 2 && child <= end - 2) {
        if (NB_LE(v[child], v[child + 1])) ++child;
        if (NB_GE(e, v[child])) { v[pos] = e; return; }
        v[pos] = v[child];
        pos = child;
        child = 2 * pos + 1;
    }
    if (child == end - 1 && NB_LT(e, v[child])) {
        v[pos] = v[child];
        pos = child;
    }
    v[pos] = e;
}
static void FN(heap_into_sorted)(FN(heap_t) *h)
{
    size_t end = h->len;
    while (end > 1) {
        --end;
        FN(nb_t) t = h->v[0]; h->v[0] = h->v[end]; h->v[end] = t;
        FN(heap_sift_down_range)(h->v, 0, end);
    }
}

/* src/ball_tree.rs:203-243  nearest_k_neighbors_in_subtree */
static void FN(knn_subtree)(FN(tree_t) *t, const T *q, size_t root, T *radius, size_t k,
                            FN(heap_t) *h)
{
    FN(node_t) *nd = &t->nodes[root];
    if (FN(node_lb)(t, nd, q, &t->n_centroid_evals) > *radius) return; /* :212 */
    if (nd->is_leaf) {
        for (size_t a = nd->start; a < nd->end; ++a) { /* :217-226 */
            size_t i = t->idx[a];
            FN(nb_t) nb = { i, FN(metric_distance)(t->metric, q, t->pts + i * t->ld, t->dim) };
            ++t->n_point_evals;
            if (h->len < k) {
                FN(heap_push)(h, nb);
            } else if (NB_LT(nb, h->v[0])) {
                FN(heap_pop)(h);
                FN(heap_push)(h, nb);
            }
        }
    } else {
        size_t c1 = root * 2 + 1, c2 = c1 + 1;
        T lb1 = FN(node_lb)(t, &t->nodes[c1], q, &t->n_centroid_evals);
        T lb2 = FN(node_lb)(t, &t->nodes[c2], q, &t->n_centroid_evals);
        if (!(lb1 < lb2)) { size_t s = c1; c1 = c2; c2 = s; } /* :232-236 */
        FN(knn_subtree)(t, q, c1, radius, k, h);
        FN(knn_subtree)(t, q, c2, radius, k, h);
    }
    if (h->len == k) *radius = h->v[0].d; /* :240-242 */
}

/* src/ball_tree.rs:102-121  BallTree::query.  Returns number of results. */
size_t FN(oracle_tree_query)(FN(tree_t) *t, const T *q, size_t k, uint64_t *idx_out, T *dist_out)
{
    if (k == 0) return 0; /* :106-108 */
    /* BinaryHeap::with_capacity(k) only reserves; at most min(k, n)+1 live */
    size_t cap = (k < t->n ? k : t->n) + 1;
    FN(heap_t) h = { (FN(nb_t) *)malloc(cap * sizeof(FN(nb_t))), 0 };
    T radius = (T)INFINITY;
    FN(knn_subtree)(t, q, 0, &radius, k, &h);
    FN(heap_into_sorted)(&h); /* :117 */
    for (size_t j = 0; j < h.len; ++j) {
        idx_out[j] = h.v[j].idx;
        dist_out[j] = h.v[j].d;
    }
    size_t r = h.len;
    free(h.v);
    return r;
}

/* src/ball_tree.rs:149-196  nearest_neighbor_in_subtree; returns 1 = Some */
static int FN(nn_subtree)(FN(tree_t) *t, const T *q, size_t root, T radius, size_t *oi, T *od)
{
    FN(node_t) *nd = &t->nodes[root];
    T lb = FN(node_lb)(t, nd, q, &t->n_centroid_evals);
    if (lb > radius) return 0; /* :157-159 */
    if (nd->is_leaf) {
        size_t min_i = 0;
        T min_d = (T)INFINITY; /* fold seed (0, inf) :162-163 */
        for (size_t a = nd->start; a < nd->end; ++a) {
            size_t i = t->idx[a];
            T d = FN(metric_distance)(t->metric, q, t->pts + i * t->ld, t->dim);
            ++t->n_point_evals;
            if (d < min_d) { min_i = i; min_d = d; } /* :167 */
        }
        if (min_d <= radius) { *oi = min_i; *od = min_d; return 1; } /* :174 */
        return 0;
    }
    size_t c1 = root * 2 + 1, c2 = c1 + 1;
    T lb1 = FN(node_lb)(t, &t->nodes[c1], q, &t->n_centroid_evals);
    T lb2 = FN(node_lb)(t, &t->nodes[c2], q, &t->n_centroid_evals);
    if (!(lb1 < lb2)) { size_t s = c1; c1 = c2; c2 = s; }
    size_t i1; T d1;
    if (FN(nn_subtree)(t, q, c1, radius, &i1, &d1)) { /* :189-194 */
        size_t i2; T d2;
        if (FN(nn_subtree)(t, q, c2, d1, &i2, &d2)) { *oi = i2; *od = d2; }
        else { *oi = i1; *od = d1; }
        return 1;
    }
    return FN(nn_subtree)(t, q, c2, radius, oi, od);
}

/* src/ball_tree.rs:80-86 BallTree::query_nearest, and the private
 * nearest_neighbor_in_subtree(point, root, radius) used by the test at :670. */
int FN(oracle_tree_nearest_in_subtree)(FN(tree_t) *t, const T *q, size_t root, T radius,
                                       uint64_t *idx_out, T *dist_out)
{
    size_t i = 0; T d = (T)0;
    int some = FN(nn_subtree)(t, q, root, radius, &i, &d);
    if (some) { *idx_out = i; *dist_out = d; }
    return some;
}

/* src/ball_tree.rs:137-142, 250-294  query_radius: explicit stack, right
 * child popped first, whole-node accept on ub <= r, leaf test strict '<'.
 * Returns the count; writes traversal order into out (capacity n). */
size_t FN(oracle_tree_query_radius)(FN(tree_t) *t, const T *q, T radius, uint64_t *out)
{
    size_t cnt = 0;
    size_t cap = 64, sp = 0;
    size_t *stack = (size_t *)malloc(cap * sizeof(size_t));
    stack[sp++] = 0;
    for (;;) {
        size_t sub = stack[--sp];
        FN(node_t) *nd = &t->nodes[sub];
        T cd = FN(metric_distance)(t->metric, q, nd->centroid, t->dim); /* :463-471 */
        ++t->n_centroid_evals;
        T lb = cd - nd->radius;
        if (lb < (T)0) lb = (T)0;
        T ub = cd + nd->radius;
        if (lb > radius) {
            if (sp == 0) break;
            continue;
        }
        if (ub <= radius) { /* :271-273 */
            for (size_t a = nd->start; a < nd->end; ++a) out[cnt++] = t->idx[a];
        } else if (nd->is_leaf) { /* :274-282 */
            for (size_t a = nd->start; a < nd->end; ++a) {
                size_t i = t->idx[a];
                T d = FN(metric_distance)(t->metric, q, t->pts + i * t->ld, t->dim);
                ++t->n_point_evals;
                if (d < radius) out[cnt++] = i;
            }
        } else {
            if (sp + 2 > cap) { cap *= 2; stack = (size_t *)realloc(stack, cap * sizeof(size_t)); }
            stack[sp++] = sub * 2 + 1; /* :284-285 */
            stack[sp++] = sub * 2 + 2;
        }
        if (sp == 0) break;
    }
    free(stack);
    return cnt;
}

/* ---- batched / multi-threaded drivers for the CPU baseline (bench.py).
 * The reference is single-threaded; `Euclidean: Sync` (src/distance.rs:19)
 * lets a caller share one tree between threads, which is what nthreads > 1
 * models: one query per thread at a time, static partition. ---- */
typedef struct {
    FN(tree_t) *t; const T *q; size_t qld, k, lo, hi, kout;
    uint64_t *idx_out; T *dist_out;
} FN(qjob_t);

static void *FN(qjob_run)(void *p)
{
    FN(qjob_t) *j = (FN(qjob_t) *)p;
    /* private copy of the counters so threads do not race on instrumentation */
    FN(tree_t) local = *j->t;
    for (size_t a = j->lo; a < j->hi; ++a)
        FN(oracle_tree_query)(&local, j->q + a * j->qld, j->k,
                              j->idx_out + a * j->kout, j->dist_out + a * j->kout);
    return NULL;
}

size_t FN(oracle_tree_query_batch)(FN(tree_t) *t, const T *q, size_t nq, size_t qld, size_t k,
                                   int nthreads, uint64_t *idx_out, T *dist_out)
{
    size_t kout = k < t->n ? k : t->n;
    if (kout == 0 || nq == 0) return kout;
    if (nthreads < 1) nthreads = 1;
    if ((size_t)nthreads > nq) nthreads = (int)nq;
    pthread_t *th = (pthread_t *)malloc(nthreads * sizeof(pthread_t));
    FN(qjob_t) *jobs = (FN(qjob_t) *)malloc(nthreads * sizeof(FN(qjob_t)));
    for (int i = 0; i < nthreads; ++i) {
        FN(qjob_t) j = { t, q, qld, k, nq * i / nthreads, nq * (i + 1) / nthreads, kout,
                         idx_out, dist_out };
        jobs[i] = j;
        if (nthreads == 1) FN(qjob_run)(&jobs[i]);
        else pthread_create(&th[i], NULL, FN(qjob_run), &jobs[i]);
    }
    if (nthreads > 1)
        for (int i = 0; i < nthreads; ++i) pthread_join(th[i], NULL);
    free(th); free(jobs);
    return kout;
}

#undef NB_LE
#undef NB_LT
#undef NB_GE
#undef FN
#undef CAT
#undef CAT_
