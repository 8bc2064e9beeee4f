"""ctypes front-end of the CPU oracle (oracle/liboracle.so).  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module; the product package (``petal_neighbors_amd``) never
does.  Every function is a thin view of the C restatement in ``oracle.c`` /
``oracle_impl.h``, which cites the reference lines it follows.

Pinning status: checked against the reference's own golden vectors
(``tests/golden/reference_kats.json``); equal-distance index order is
"parity unpinned" (depends on Rust's std ``BinaryHeap``, restated from memory).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

ERR_EMPTY = 1           # ArrayError::Empty          (src/lib.rs:12-13)
ERR_NOT_CONTIGUOUS = 2  # ArrayError::NotContiguous  (src/lib.rs:14-15)
ERR_EMPTY_MATRIX_PANIC = 3  # max_spread_column assert (src/ball_tree.rs:582)


def build(force: bool = False) -> str:
    """Compile liboracle.so with the committed recipe (oracle/Makefile)."""
    srcs = [os.path.join(_HERE, f) for f in ("oracle.c", "oracle_impl.h", "Makefile")]
    stale = (not os.path.exists(_LIB_PATH)) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs)
    if force or stale:
        subprocess.run(["make", "-C", _HERE, "-B" if force else "-s", "liboracle.so"],
                       check=True, capture_output=True)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _declare(_lib)
    return _lib


_SZ, _U64P, _I, _VP = C.c_size_t, C.POINTER(C.c_uint64), C.c_int, C.c_void_p


def _declare(L):
    for sfx, ct in (("f32", C.c_float), ("f64", C.c_double)):
        P = C.POINTER(ct)
        f = getattr(L, f"oracle_euclidean_{sfx}"); f.restype = ct; f.argtypes = [P, P, _SZ]
        f = getattr(L, f"oracle_reuclidean_{sfx}"); f.restype = ct; f.argtypes = [P, P, _SZ]
        f = getattr(L, f"oracle_rdistance_to_distance_{sfx}"); f.restype = ct; f.argtypes = [ct]
        f = getattr(L, f"oracle_distance_to_rdistance_{sfx}"); f.restype = ct; f.argtypes = [ct]
        f = getattr(L, f"oracle_pairwise_{sfx}"); f.restype = None; f.argtypes = [P, _SZ, _SZ, _SZ, P]
        f = getattr(L, f"oracle_cosine_{sfx}"); f.restype = ct; f.argtypes = [P, _SZ, P, _SZ]
        f = getattr(L, f"oracle_pairwise_cosine_{sfx}"); f.restype = None; f.argtypes = [P, _SZ, _SZ, _SZ, P]
        f = getattr(L, f"oracle_brute_knn_{sfx}"); f.restype = _SZ
        f.argtypes = [P, _SZ, _SZ, _SZ, P, _SZ, _SZ, _SZ, _U64P, P]
        f = getattr(L, f"oracle_brute_radius_{sfx}"); f.restype = _SZ
        f.argtypes = [P, _SZ, _SZ, _SZ, P, ct, _U64P]
        f = getattr(L, f"oracle_node_init_{sfx}"); f.restype = None
        f.argtypes = [P, _SZ, _SZ, C.POINTER(_SZ), _SZ, P, P]
        f = getattr(L, f"oracle_max_spread_column_{sfx}"); f.restype = _SZ
        f.argtypes = [P, _SZ, _SZ, C.POINTER(_SZ), _SZ]
        f = getattr(L, f"oracle_halve_node_indices_{sfx}"); f.restype = None
        f.argtypes = [C.POINTER(_SZ), _SZ, P, _SZ]
        f = getattr(L, f"oracle_tree_build_{sfx}"); f.restype = _VP
        f.argtypes = [P, _SZ, _SZ, _SZ, C.c_ssize_t, C.POINTER(_I)]
        f = getattr(L, f"oracle_tree_build_mt_{sfx}"); f.restype = _VP
        f.argtypes = [P, _SZ, _SZ, _SZ, C.c_ssize_t, _I, C.POINTER(_I)]
        f = getattr(L, f"oracle_tree_build_metric_{sfx}"); f.restype = _VP
        f.argtypes = [P, _SZ, _SZ, _SZ, C.c_ssize_t, _I, _I, C.POINTER(_I)]
        f = getattr(L, f"oracle_tree_free_{sfx}"); f.restype = None; f.argtypes = [_VP]
        f = getattr(L, f"oracle_tree_num_nodes_{sfx}"); f.restype = _SZ; f.argtypes = [_VP]
        f = getattr(L, f"oracle_tree_idx_{sfx}"); f.restype = C.POINTER(_SZ); f.argtypes = [_VP]
        f = getattr(L, f"oracle_tree_node_{sfx}"); f.restype = None
        f.argtypes = [_VP, _SZ, C.POINTER(_SZ), C.POINTER(_SZ), P, C.POINTER(_I), P]
        f = getattr(L, f"oracle_tree_eval_counts_{sfx}"); f.restype = None
        f.argtypes = [_VP, _U64P, _U64P, _I]
        f = getattr(L, f"oracle_tree_query_{sfx}"); f.restype = _SZ
        f.argtypes = [_VP, P, _SZ, _U64P, P]
        f = getattr(L, f"oracle_tree_nearest_in_subtree_{sfx}"); f.restype = _I
        f.argtypes = [_VP, P, _SZ, ct, _U64P, P]
        f = getattr(L, f"oracle_tree_query_radius_{sfx}"); f.restype = _SZ
        f.argtypes = [_VP, P, ct, _U64P]
        f = getattr(L, f"oracle_tree_query_batch_{sfx}"); f.restype = _SZ
        f.argtypes = [_VP, P, _SZ, _SZ, _SZ, _I, _U64P, P]
        f = getattr(L, f"oracle_fill_uniform_{sfx}"); f.restype = None
        f.argtypes = [P, C.c_uint64, C.c_uint64, C.c_uint64]


def _sfx(dtype):
    dtype = np.dtype(dtype)
    if dtype == np.float32:
        return "f32", C.c_float
    if dtype == np.float64:
        return "f64", C.c_double
    raise TypeError(f"oracle supports float32/float64, got {dtype}")


def _ptr(a, ct):
    return a.ctypes.data_as(C.POINTER(ct))


def _rows(a):
    """C-contiguous 2-D view + leading dimension (elements)."""
    a = np.ascontiguousarray(a)
    return a, (a.shape[1] if a.ndim == 2 else a.shape[0])


class OracleArrayError(ValueError):
    def __init__(self, code):
        self.code = code
        super().__init__({1: "array is empty", 2: "array is not contiguous in memory",
                          3: "empty matrix"}.get(code, f"error {code}"))


# ---------------------------------------------------------------- metric
def euclidean(a, b):
    """src/distance.rs:26-35 (zip truncates to the shorter vector)."""
    a = np.ascontiguousarray(a); b = np.ascontiguousarray(b, dtype=a.dtype)
    s, ct = _sfx(a.dtype)
    n = min(a.shape[0], b.shape[0])
    return a.dtype.type(getattr(lib(), f"oracle_euclidean_{s}")(_ptr(a, ct), _ptr(b, ct), n))


def reuclidean(a, b):
    """src/distance.rs:37-45"""
    a = np.ascontiguousarray(a); b = np.ascontiguousarray(b, dtype=a.dtype)
    s, ct = _sfx(a.dtype)
    n = min(a.shape[0], b.shape[0])
    return a.dtype.type(getattr(lib(), f"oracle_reuclidean_{s}")(_ptr(a, ct), _ptr(b, ct), n))


def cosine(a, b):
    """Cosine::distance, src/distance.rs:85-107 (rdistance and both conversions are the identity, :109-121)."""
    a = np.ascontiguousarray(a); b = np.ascontiguousarray(b, dtype=a.dtype)
    s, ct = _sfx(a.dtype)
    return a.dtype.type(getattr(lib(), f"oracle_cosine_{s}")(_ptr(a, ct), a.shape[0], _ptr(b, ct), b.shape[0]))


def pairwise_cosine(x):
    """src/distance.rs:58-74 with the Cosine metric"""
    x, ld = _rows(np.atleast_2d(x))
    s, ct = _sfx(x.dtype)
    n, d = x.shape
    out = np.empty((n, n), dtype=x.dtype)
    getattr(lib(), f"oracle_pairwise_cosine_{s}")(_ptr(x, ct), n, d, ld, _ptr(out, ct))
    return out


def rdistance_to_distance(d, dtype=np.float64):
    s, ct = _sfx(dtype)
    return np.dtype(dtype).type(getattr(lib(), f"oracle_rdistance_to_distance_{s}")(ct(d)))


def distance_to_rdistance(d, dtype=np.float64):
    s, ct = _sfx(dtype)
    return np.dtype(dtype).type(getattr(lib(), f"oracle_distance_to_rdistance_{s}")(ct(d)))


def pairwise(x):
    """src/distance.rs:58-74"""
    x, ld = _rows(np.atleast_2d(x))
    s, ct = _sfx(x.dtype)
    n, d = x.shape
    out = np.empty((n, n), dtype=x.dtype)
    getattr(lib(), f"oracle_pairwise_{s}")(_ptr(x, ct), n, d, ld, _ptr(out, ct))
    return out


# ------------------------------------------------------- canonical brute force
def brute_knn(points, queries, k):
    """The specification: k smallest by (distance total order, index)."""
    p, ld = _rows(points)
    q = np.ascontiguousarray(np.atleast_2d(queries), dtype=p.dtype)
    s, ct = _sfx(p.dtype)
    n, d = p.shape
    nq = q.shape[0]
    kout = min(k, n)
    idx = np.empty((nq, kout), dtype=np.uint64)
    dist = np.empty((nq, kout), dtype=p.dtype)
    if kout:
        getattr(lib(), f"oracle_brute_knn_{s}")(_ptr(p, ct), n, d, ld, _ptr(q, ct), nq, q.shape[1],
                                                 k, _ptr(idx, C.c_uint64), _ptr(dist, ct))
    return idx, dist


def brute_radius(points, query, r):
    """{ i : distance(q, p_i) < r }, ascending."""
    p, ld = _rows(points)
    q = np.ascontiguousarray(query, dtype=p.dtype)
    s, ct = _sfx(p.dtype)
    n, d = p.shape
    out = np.empty(n, dtype=np.uint64)
    c = getattr(lib(), f"oracle_brute_radius_{s}")(_ptr(p, ct), n, d, ld, _ptr(q, ct), ct(r),
                                                    _ptr(out, C.c_uint64))
    return out[:c].copy()


# ------------------------------------------------------------ faithful tree
def node_init(points, idx):
    p, ld = _rows(points)
    s, ct = _sfx(p.dtype)
    ia = np.ascontiguousarray(idx, dtype=np.uintp)
    cen = np.empty(p.shape[1], dtype=p.dtype)
    rad = ct(0)
    getattr(lib(), f"oracle_node_init_{s}")(_ptr(p, ct), p.shape[1], ld, _ptr(ia, _SZ), len(ia),
                                             _ptr(cen, ct), C.byref(rad))
    return cen, p.dtype.type(rad.value)


def max_spread_column(points, idx):
    p, ld = _rows(points)
    s, ct = _sfx(p.dtype)
    ia = np.ascontiguousarray(idx, dtype=np.uintp)
    if len(ia) and p.shape[0] and int(ia.max()) >= p.shape[0]:
        raise IndexError("index out of bounds")  # src/ball_tree.rs:583-586
    r = getattr(lib(), f"oracle_max_spread_column_{s}")(_ptr(p, ct), p.shape[1] if p.ndim == 2 else 0,
                                                         ld, _ptr(ia, _SZ), len(ia))
    if r == C.c_size_t(-1).value:
        raise ValueError("empty matrix")  # src/ball_tree.rs:582
    return int(r)


def halve_node_indices(idx, col):
    col = np.ascontiguousarray(col)
    s, ct = _sfx(col.dtype)
    ia = np.ascontiguousarray(idx, dtype=np.uintp).copy()
    if len(ia) == 0:
        raise OverflowError("attempt to subtract with overflow")  # src/ball_tree.rs:549
    getattr(lib(), f"oracle_halve_node_indices_{s}")(_ptr(ia, _SZ), len(ia), _ptr(col, ct), 1)
    return ia


class Tree:
    """Faithful restatement of ``BallTree<A, M>`` (src/ball_tree.rs), M = Euclidean (default) or Cosine
    (``metric="cosine"``: BallTree::new(points, Cosine) -- every distance of the build and of the walks goes through
    Metric::distance, src/ball_tree.rs:165,218,276,309,459,464,474)."""

    def __init__(self, points, build_threads_log2=0, metric="euclidean"):
        a = np.asarray(points)
        if a.ndim != 2:
            raise ValueError("points must be 2-D")
        if a.dtype not in (np.float32, np.float64):
            a = a.astype(np.float64)
        self._s, self._ct = _sfx(a.dtype)
        n, d = a.shape
        # the reference checks only the INNER stride (src/ball_tree.rs:47)
        col_stride = (a.strides[1] // a.itemsize) if d > 0 else 1
        if n > 0 and d > 1 and col_stride != 1:
            raise OracleArrayError(ERR_NOT_CONTIGUOUS)
        self.points = np.ascontiguousarray(a)
        err = _I(0)
        L = lib()
        self.metric = {"euclidean": 0, "cosine": 1}[metric]
        self._h = getattr(L, f"oracle_tree_build_metric_{self._s}")(
            _ptr(self.points, self._ct), n, d, max(d, 1) if self.points.size else 0, 1,
            int(build_threads_log2), self.metric, C.byref(err))
        if not self._h:
            raise OracleArrayError(err.value)
        self.n, self.dim = n, d

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            getattr(lib(), f"oracle_tree_free_{self._s}")(h)
            self._h = None

    @property
    def num_nodes(self):
        return getattr(lib(), f"oracle_tree_num_nodes_{self._s}")(self._h)

    @property
    def idx(self):
        p = getattr(lib(), f"oracle_tree_idx_{self._s}")(self._h)
        return np.ctypeslib.as_array(p, shape=(self.n,)).copy()

    def node(self, i):
        st, en, leaf = _SZ(0), _SZ(0), _I(0)
        rad = self._ct(0)
        cen = np.empty(self.dim, dtype=self.points.dtype)
        getattr(lib(), f"oracle_tree_node_{self._s}")(self._h, i, C.byref(st), C.byref(en),
                                                       C.byref(rad), C.byref(leaf), _ptr(cen, self._ct))
        return dict(range=(st.value, en.value), radius=rad.value, is_leaf=bool(leaf.value), centroid=cen)

    def eval_counts(self, reset=True):
        a, b = C.c_uint64(0), C.c_uint64(0)
        getattr(lib(), f"oracle_tree_eval_counts_{self._s}")(self._h, C.byref(a), C.byref(b), int(reset))
        return a.value, b.value

    def _q(self, point):
        q = np.ascontiguousarray(point, dtype=self.points.dtype)
        if q.shape[0] < self.dim:  # zip truncation is not restated: callers pass full-length queries
            raise ValueError("query shorter than the tree dimension")
        return q

    def query(self, point, k):
        """BallTree::query (src/ball_tree.rs:102-121)."""
        q = self._q(point)
        kout = min(k, self.n)
        idx = np.empty(max(kout, 1), dtype=np.uint64)
        dist = np.empty(max(kout, 1), dtype=self.points.dtype)
        r = getattr(lib(), f"oracle_tree_query_{self._s}")(self._h, _ptr(q, self._ct), k,
                                                            _ptr(idx, C.c_uint64), _ptr(dist, self._ct))
        return idx[:r].copy(), dist[:r].copy()

    def query_batch(self, queries, k, nthreads=1):
        q = np.ascontiguousarray(np.atleast_2d(queries), dtype=self.points.dtype)
        nq = q.shape[0]
        kout = min(k, self.n)
        idx = np.empty((nq, kout), dtype=np.uint64)
        dist = np.empty((nq, kout), dtype=self.points.dtype)
        if kout and nq:
            getattr(lib(), f"oracle_tree_query_batch_{self._s}")(
                self._h, _ptr(q, self._ct), nq, q.shape[1], k, nthreads,
                _ptr(idx, C.c_uint64), _ptr(dist, self._ct))
        return idx, dist

    def query_nearest(self, point):
        """BallTree::query_nearest (src/ball_tree.rs:80-86)."""
        r = self.nearest_in_subtree(point, 0, np.inf)
        assert r is not None, "0 is a valid index"
        return r

    def nearest_in_subtree(self, point, root, radius):
        """nearest_neighbor_in_subtree (src/ball_tree.rs:149-196) -> (idx, dist) | None."""
        q = self._q(point)
        i = C.c_uint64(0)
        d = self._ct(0)
        some = getattr(lib(), f"oracle_tree_nearest_in_subtree_{self._s}")(
            self._h, _ptr(q, self._ct), root, self._ct(radius), C.byref(i), C.byref(d))
        return (int(i.value), self.points.dtype.type(d.value)) if some else None

    def query_radius(self, point, distance):
        """BallTree::query_radius (src/ball_tree.rs:137-142), traversal order."""
        q = self._q(point)
        out = np.empty(self.n, dtype=np.uint64)
        c = getattr(lib(), f"oracle_tree_query_radius_{self._s}")(self._h, _ptr(q, self._ct),
                                                                   self._ct(distance), _ptr(out, C.c_uint64))
        return out[:c].copy()


# -------------------------------------------------------------- synthetic data
def fill_uniform(count, seed, first_ctr=0, dtype=np.float32):
    """uniform [0,1) with exactly 24 random bits: (mix32(seed, ctr) >> 8) * 2^-24."""
    s, ct = _sfx(dtype)
    out = np.empty(count, dtype=dtype)
    getattr(lib(), f"oracle_fill_uniform_{s}")(_ptr(out, ct), count, seed, first_ctr)
    return out


# -------------------------------------------------------------- answer comparison
def compare_knn(idx_a, dist_a, idx_b, dist_b):
    """Two k-NN answers for the same queries (rows ascending by distance): the result contract of SURVEY.md App. A.2/A.3.

    Distances must be bit-identical.  Indices must be identical except INSIDE groups of exactly equal distances, where
    the reference's order is unspecified (tree shape + BinaryHeap internals): there the two index SETS must agree -- and
    in the group that reaches the k-th position even the sets may differ (either member of a tie at the cut is a valid
    k-th neighbour).  Returns a dict of counts (queries): ``dist_mismatch``, ``idx_mismatch`` (outside tie groups or a
    set difference inside a closed one), ``tie_reordered`` (same sets, different order), ``tie_at_cut`` (sets differ only
    in the group cut by k), and ``first_bad`` = the first offending query or -1."""
    ia, ib = np.asarray(idx_a).astype(np.uint64), np.asarray(idx_b).astype(np.uint64)
    da, db = np.ascontiguousarray(dist_a), np.ascontiguousarray(dist_b)
    assert ia.shape == ib.shape == da.shape == db.shape and da.dtype == db.dtype, "answers of different shape or type"
    nq, k = ia.shape
    ut = np.uint32 if da.dtype == np.float32 else np.uint64
    ba, bb = da.view(ut), db.view(ut)
    out = dict(queries=int(nq), dist_mismatch=0, idx_mismatch=0, tie_reordered=0, tie_at_cut=0, first_bad=-1)
    drow = (ba != bb).any(axis=1) if k else np.zeros(nq, bool)
    irow = (ia != ib).any(axis=1) if k else np.zeros(nq, bool)
    for q in np.nonzero(drow | irow)[0]:
        if drow[q]:
            out["dist_mismatch"] += 1
            out["first_bad"] = int(q) if out["first_bad"] < 0 else out["first_bad"]
            continue
        bad = reordered = cut = False
        j = 0
        while j < k:
            e = j + 1
            while e < k and ba[q, e] == ba[q, j]:
                e += 1
            if not np.array_equal(ia[q, j:e], ib[q, j:e]):
                if np.array_equal(np.sort(ia[q, j:e]), np.sort(ib[q, j:e])):
                    reordered = True
                elif e == k:
                    # the group that the cut at k goes through: members beyond k exist only if more rows tie at this
                    # distance, which the caller cannot see from k entries -- reported separately, not as a mismatch
                    cut = True
                else:
                    bad = True
            j = e
        if bad:
            out["idx_mismatch"] += 1
            out["first_bad"] = int(q) if out["first_bad"] < 0 else out["first_bad"]
        elif cut:
            out["tie_at_cut"] += 1
        elif reordered:
            out["tie_reordered"] += 1
    out["agree"] = out["dist_mismatch"] == 0 and out["idx_mismatch"] == 0
    return out
