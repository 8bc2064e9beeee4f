"""Host mirror of ``petal_neighbors::BallTree`` (reference src/ball_tree.rs:15-374).

Same constructor names, argument meaning, result shapes and error behaviour as
the Rust type, so tests read like the reference's own; the work is done by the
HIP kernels behind the C ABI (``include/petal_mi355x.h``).  The ball tree walk
is not reproduced: ``BallTree`` here owns a zero-padded copy of the points in
HBM and answers queries by batched exact scans.

Batch methods (``query_batch`` etc.) are extensions: the reference takes one
point per call (src/ball_tree.rs:102); ``nq = 1`` is that call.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .distance import Euclidean
from .errors import check

ENGINES = {"auto": _lib.PN_ENGINE_AUTO, "exact": _lib.PN_ENGINE_EXACT, "mfma": _lib.PN_ENGINE_MFMA,
           "bf16": _lib.PN_ENGINE_BF16}


def _float_array(a):
    a = np.asarray(a)
    if a.dtype not in (np.float32, np.float64):  # `A: FloatCore` is f32 or f64 in practice
        a = a.astype(np.float64)
    return a


class BallTree:
    """``BallTree<'a, A, Euclidean>``: ``points`` / ``metric`` stay readable like the pub fields
    at src/ball_tree.rs:20-23 (``idx`` and ``nodes`` do not exist: no tree is built)."""

    def __init__(self, handle, points, metric, dtype, device):
        self._h = C.c_void_p(handle)
        self.points = points
        self.metric = metric
        self.dtype = np.dtype(dtype)
        self.device = device
        self._sfx = "f32" if self.dtype == np.float32 else "f64"
        info = _lib.PnInfo()
        check(_lib.lib().pn_index_info(self._h, C.byref(info)))
        self._n, self._dim = int(info.n_points), int(info.dim)
        self.mfma_eligible = bool(info.mfma_eligible)
        self.bf16_eligible = bool(info.bf16_eligible)
        self.bf16_layout = int(info.bf16_layout)
        self.seed_model = bool(info.seed_model)

    # ------------------------------------------------------------ construction
    @classmethod
    def new(cls, points, metric, device: int = 0):
        """``BallTree::new(points, metric)`` (src/ball_tree.rs:38-63).

        Raises ``ArrayError.Empty`` for zero rows and ``ArrayError.NotContiguous`` when the
        inner stride is not 1 (only the inner stride is checked, as at :47)."""
        from .distance import Cosine
        if not isinstance(metric, (Euclidean, Cosine)):
            raise NotImplementedError("the MI355X path serves distance.Euclidean and distance.Cosine")
        cosine = isinstance(metric, Cosine)
        a = _float_array(points)
        if a.ndim != 2:
            raise ValueError("points must be a 2-D array (Ix2)")
        n, d = a.shape
        item = a.itemsize
        rs, cs = (a.strides[0] // item, a.strides[1] // item)
        if a.strides[0] % item or a.strides[1] % item or (n > 1 and rs < 0):
            a = np.ascontiguousarray(a)  # exotic views: ndarray would accept them; copy, same values
            rs, cs = d, 1
        if d <= 1:
            cs = 1
            if d == 1 and a.strides[1] != item:
                a = np.ascontiguousarray(a)
                rs = d
        L = _lib.lib()
        h = C.c_void_p(0)
        # Cosine: an exact scan under Cosine::distance (it is not a metric: the reference's pruned walk can miss true
        # neighbours under it; include/petal_mi355x.h, pn_index_create_cosine_*)
        sfx = "f32" if a.dtype == np.float32 else "f64"
        fn = getattr(L, f"pn_index_create_cosine_{sfx}" if cosine else f"pn_index_create_{sfx}")
        check(fn(a.ctypes.data if a.size else None, n, d, rs, cs, device, C.byref(h)))
        return cls(h.value, a, metric, a.dtype, device)

    @classmethod
    def euclidean(cls, points, device: int = 0):
        """``BallTree::euclidean(points)`` (src/ball_tree.rs:367-373)."""
        return cls.new(points, Euclidean(), device)

    @classmethod
    def from_device(cls, tensor, stream=None):
        """Build from a row-major float32 or float64 CUDA tensor already in HBM (extension)."""
        import torch
        if tensor.dtype not in (torch.float32, torch.float64) or tensor.dim() != 2 or not tensor.is_cuda:
            raise ValueError("from_device takes a 2-D float32 or float64 CUDA tensor")
        if tensor.shape[1] > 1 and tensor.stride(1) != 1:
            from .errors import NotContiguous
            raise NotContiguous()
        n, d = tensor.shape
        dev = tensor.device.index or 0
        h = C.c_void_p(0)
        st = stream if stream is not None else torch.cuda.current_stream(tensor.device).cuda_stream
        f64 = tensor.dtype == torch.float64
        create = _lib.lib().pn_index_create_device_f64 if f64 else _lib.lib().pn_index_create_device_f32
        check(create(tensor.data_ptr() if n * d else None, n, d, tensor.stride(0) if n > 1 else max(d, 1), dev,
                     C.c_void_p(st), C.byref(h)))
        return cls(h.value, None, Euclidean(), np.float64 if f64 else np.float32, dev)

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            _lib.lib().pn_index_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:  # interpreter shutdown
            pass

    # ------------------------------------------------------------------ options
    def set_engine(self, name: str):
        check(_lib.lib().pn_index_set_option(self._h, _lib.PN_OPT_ENGINE, ENGINES[name]))
        return self

    def set_option(self, opt: int, value: int):
        check(_lib.lib().pn_index_set_option(self._h, opt, int(value)))
        return self

    def stats(self, reset: bool = False):
        s = _lib.PnStats()
        check(_lib.lib().pn_index_get_stats(self._h, C.byref(s), int(reset)))
        return {k: getattr(s, k) for k, _ in s._fields_ if k != "reserved"}

    # ---------------------------------------------------------------- accessors
    def num_points(self) -> int:
        """``BallTree::num_points`` (src/ball_tree.rs:351-353)."""
        return self._n

    @property
    def dim(self) -> int:
        return self._dim

    # Tree introspection (src/ball_tree.rs:296-353).  No tree exists until one of these is called; the library then
    # builds the reference's implicit ball tree once, on the host (csrc/tree.cpp) -- queries never use it.
    def num_nodes(self) -> int:
        """``BallTree::num_nodes`` (src/ball_tree.rs:345-348)."""
        out = C.c_uint64(0)
        check(_lib.lib().pn_tree_num_nodes(self._h, C.byref(out)))
        return int(out.value)

    def _node(self, n):
        n = int(n)
        if n < 0 or n >= self.num_nodes():
            raise IndexError(f"node {n} out of range")  # the reference panics
        return n

    def children_of(self, n: int):
        """``BallTree::children_of(n) -> Option<(usize, usize)>`` (src/ball_tree.rs:320-329): None for a leaf."""
        some, left, right = C.c_int(0), C.c_uint64(0), C.c_uint64(0)
        check(_lib.lib().pn_tree_children_of(self._h, self._node(n), C.byref(some), C.byref(left), C.byref(right)))
        return (int(left.value), int(right.value)) if some.value else None

    def points_of(self, n: int):
        """``BallTree::points_of(n) -> &[usize]`` (src/ball_tree.rs:331-334): the node's slice of ``idx``."""
        ptr, cnt = C.POINTER(C.c_uint64)(), C.c_uint64(0)
        check(_lib.lib().pn_tree_points_of(self._h, self._node(n), C.byref(ptr), C.byref(cnt)))
        return np.ctypeslib.as_array(ptr, shape=(int(cnt.value),)).copy() if cnt.value else np.empty(0, dtype=np.uint64)

    def radius_of(self, n: int):
        """``BallTree::radius_of(n)`` (src/ball_tree.rs:336-339)."""
        out = C.c_float(0) if self._sfx == "f32" else C.c_double(0)
        check(getattr(_lib.lib(), f"pn_tree_radius_of_{self._sfx}")(self._h, self._node(n), C.byref(out)))
        return self.dtype.type(out.value)

    def compare_nodes(self, x: int, y: int):
        """``BallTree::compare_nodes(x, y) -> Option<Ordering>`` by radius (src/ball_tree.rs:340-343):
        -1 / 0 / 1 for Less / Equal / Greater, None when a radius is NaN."""
        out = C.c_int(0)
        check(_lib.lib().pn_tree_compare_nodes(self._h, self._node(x), self._node(y), C.byref(out)))
        return None if out.value == 2 else int(out.value)

    def node_distance_lower_bound(self, n1: int, n2: int):
        """``BallTree::node_distance_lower_bound(n1, n2)`` = max(|c1 - c2| - R1 - R2, 0) (src/ball_tree.rs:303-318)."""
        out = C.c_float(0) if self._sfx == "f32" else C.c_double(0)
        check(getattr(_lib.lib(), f"pn_tree_node_distance_lower_bound_{self._sfx}")(self._h, self._node(n1), self._node(n2),
                                                                                      C.byref(out)))
        return self.dtype.type(out.value)

    def _centroid_of(self, n: int):
        c = np.empty(self._dim, dtype=self.dtype)
        check(_lib.lib().pn_tree_centroid_of(self._h, self._node(n), c.ctypes.data))
        return c

    # ------------------------------------------------------------------ queries
    def _queries(self, q, one: bool):
        a = np.ascontiguousarray(q, dtype=self.dtype)
        if one:
            if a.ndim != 1:
                raise ValueError("point must be 1-D (Ix1)")
            a = a.reshape(1, -1)
        elif a.ndim != 2:
            raise ValueError("queries must be 2-D")
        return a

    def query_batch(self, queries, k: int):
        """k-NN for every row of ``queries``: (nq, min(k, n)) indices (uint64) and distances."""
        a = self._queries(queries, False)
        nq, qc = a.shape
        kout = min(int(k), self._n)
        idx = np.empty((nq, kout), dtype=np.uint64)
        dist = np.empty((nq, kout), dtype=self.dtype)
        if nq and kout:
            fn = getattr(_lib.lib(), f"pn_query_{self._sfx}")
            check(fn(self._h, a.ctypes.data, nq, qc, max(qc, 1), int(k), idx.ctypes.data, dist.ctypes.data))
        return idx, dist

    def query(self, point, k: int):
        """``BallTree::query(point, k) -> (Vec<usize>, Vec<A>)`` ascending (src/ball_tree.rs:102-121);
        ``k == 0`` returns two empty arrays (:106-108)."""
        idx, dist = self.query_batch(self._queries(point, True), k)
        return idx[0], dist[0]

    def query_nearest(self, point):
        """``BallTree::query_nearest(point) -> (usize, A)`` (src/ball_tree.rs:80-86)."""
        a = self._queries(point, True)
        idx = np.empty(1, dtype=np.uint64)
        dist = np.empty(1, dtype=self.dtype)
        fn = getattr(_lib.lib(), f"pn_query_nearest_{self._sfx}")
        check(fn(self._h, a.ctypes.data, 1, a.shape[1], max(a.shape[1], 1), idx.ctypes.data, dist.ctypes.data))
        return int(idx[0]), dist[0]

    def query_radius_batch(self, queries, distance):
        """CSR (offsets[nq+1], indices) of ``{ i : dist(q, p_i) < distance }``, ascending per query."""
        a = self._queries(queries, False)
        nq, qc = a.shape
        offsets = np.zeros(nq + 1, dtype=np.uint64)
        out = C.c_void_p(0)
        r = C.c_float(distance) if self._sfx == "f32" else C.c_double(distance)
        fn = getattr(_lib.lib(), f"pn_query_radius_{self._sfx}")
        check(fn(self._h, a.ctypes.data, nq, qc, max(qc, 1), r, offsets.ctypes.data, C.byref(out)))
        total = int(offsets[-1])
        try:
            if total:
                buf = (C.c_uint64 * total).from_address(out.value)
                idx = np.frombuffer(buf, dtype=np.uint64).copy()
            else:
                idx = np.empty(0, dtype=np.uint64)
        finally:
            if out.value:
                _lib.lib().pn_free(out)
        return offsets, idx

    def query_radius(self, point, distance):
        """``BallTree::query_radius(point, distance) -> Vec<usize>`` (src/ball_tree.rs:137-142).
        The reference's order is unspecified (its tests sort, :667/:777); here: ascending."""
        _, idx = self.query_radius_batch(self._queries(point, True), distance)
        return idx

    # --------------------------------------------------------- device-resident
    def query_radius_device(self, queries, distance, capacity: int, out_offsets=None, out_idx=None, out_total=None,
                            stream=None):
        """``query_radius`` with queries and results in HBM, nothing read back (``pn_query_radius_device_*``).

        Returns CUDA tensors ``(offsets int64 [nq+1], idx int64 [capacity], total int64 [1])``: the CSR offsets are always
        complete; rows are written where their position is below ``capacity``; ``total > capacity`` (whenever the caller
        looks) means the buffer was too small -- call again with ``capacity >= total``."""
        import torch
        tdt = torch.float32 if self._sfx == "f32" else torch.float64
        if queries.dtype != tdt or queries.dim() != 2 or not queries.is_cuda:
            raise ValueError("queries must be a 2-D CUDA tensor of the tree's element type")
        if queries.shape[1] > 1 and queries.stride(1) != 1:
            queries = queries.contiguous()
        nq, qc = queries.shape
        dev = queries.device
        offs = out_offsets if out_offsets is not None else torch.empty(nq + 1, dtype=torch.int64, device=dev)
        idx = out_idx if out_idx is not None else torch.empty(max(int(capacity), 1), dtype=torch.int64, device=dev)
        tot = out_total if out_total is not None else torch.empty(1, dtype=torch.int64, device=dev)
        st = stream if stream is not None else torch.cuda.current_stream(dev).cuda_stream
        r = C.c_float(distance) if self._sfx == "f32" else C.c_double(distance)
        fn = getattr(_lib.lib(), f"pn_query_radius_device_{self._sfx}")
        check(fn(self._h, queries.data_ptr() if nq * qc else None, nq, qc, queries.stride(0) if nq > 1 else max(qc, 1), r,
                 offs.data_ptr(), idx.data_ptr(), int(capacity), tot.data_ptr(), C.c_void_p(st)))
        return offs, idx, tot

    def query_device(self, queries, k: int, out_idx=None, out_dist=None, stream=None):
        """k-NN with queries and results in HBM (torch CUDA tensors of the tree's element type; indices as int64)."""
        import torch
        tdt = torch.float32 if self._sfx == "f32" else torch.float64
        if queries.dtype != tdt or queries.dim() != 2 or not queries.is_cuda:
            raise ValueError(f"queries must be a 2-D {tdt} CUDA tensor")
        if queries.shape[1] > 1 and queries.stride(1) != 1:
            queries = queries.contiguous()
        nq, qc = queries.shape
        kout = min(int(k), self._n)
        if out_idx is None:
            out_idx = torch.empty((nq, kout), dtype=torch.int64, device=queries.device)
        if out_dist is None:
            out_dist = torch.empty((nq, kout), dtype=tdt, device=queries.device)
        if nq and kout:
            st = stream if stream is not None else torch.cuda.current_stream(queries.device).cuda_stream
            check(getattr(_lib.lib(), f"pn_query_device_{self._sfx}")(self._h, queries.data_ptr(), nq, qc,
                                                 queries.stride(0) if nq > 1 else max(qc, 1), int(k),
                                                 out_idx.data_ptr(), out_dist.data_ptr(), C.c_void_p(st)))
        return out_idx, out_dist
