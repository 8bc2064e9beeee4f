"""Host mirror of petal-neighbors' ``distance`` module (reference src/distance.rs).

``Euclidean`` / ``Cosine`` are the zero-sized metric tags with the four ``Metric<A>`` methods
(src/distance.rs:9-14, 21-55, 76-122); ``pairwise`` is the batched n x n matrix
(src/distance.rs:58-74) and runs on the GPU.  Everything calls the C ABI
(libpetal_mi355x.so); nothing is computed in Python/NumPy.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .errors import check


def _as_float_array(a, dtype=None):
    a = np.asarray(a)
    if dtype is not None:
        return np.ascontiguousarray(a, dtype=dtype)
    if a.dtype not in (np.float32, np.float64):
        a = a.astype(np.float64)
    return np.ascontiguousarray(a)


class Euclidean:
    """``distance::Euclidean`` (src/distance.rs:16-55): sequential unfused fold + sqrt."""

    def __eq__(self, other):  # derive(Eq, PartialEq) on a unit struct
        return isinstance(other, Euclidean)

    def __hash__(self):
        return hash("Euclidean")

    def __repr__(self):
        return "Euclidean"

    @staticmethod
    def _pair(x1, x2):
        a = _as_float_array(x1)
        b = _as_float_array(x2, a.dtype)
        if a.ndim != 1 or b.ndim != 1:
            raise ValueError("Metric::distance takes two 1-D views")
        n = min(a.shape[0], b.shape[0])  # zip truncates (src/distance.rs:27-28)
        return a, b, n, ("f32" if a.dtype == np.float32 else "f64")

    def distance(self, x1, x2):
        a, b, n, sfx = self._pair(x1, x2)
        return a.dtype.type(getattr(_lib.lib(), f"pn_euclidean_{sfx}")(a.ctypes.data, b.ctypes.data, n))

    def rdistance(self, x1, x2):
        a, b, n, sfx = self._pair(x1, x2)
        return a.dtype.type(getattr(_lib.lib(), f"pn_reuclidean_{sfx}")(a.ctypes.data, b.ctypes.data, n))

    def rdistance_to_distance(self, d):
        if isinstance(d, np.float32):
            return np.float32(_lib.lib().pn_rdistance_to_distance_f32(float(d)))
        return np.float64(_lib.lib().pn_rdistance_to_distance_f64(float(d)))

    def distance_to_rdistance(self, d):
        if isinstance(d, np.float32):
            return np.float32(_lib.lib().pn_distance_to_rdistance_f32(float(d)))
        return np.float64(_lib.lib().pn_distance_to_rdistance_f64(float(d)))


class Cosine:
    """``distance::Cosine`` (src/distance.rs:76-122): ``1 - dot / (|x1| |x2|)`` with the reference's three
    sequential sums; ``rdistance`` and both conversions are the identity there.  Served as a pair metric, by
    ``pairwise`` and as the metric of ``BallTree.new(points, Cosine())`` -- there as an EXACT scan: cosine distance is
    not a metric, so the reference's ball-pruned walk may skip true neighbours under it; this engine never does."""

    def __eq__(self, other):
        return isinstance(other, Cosine)

    def __hash__(self):
        return hash("Cosine")

    def __repr__(self):
        return "Cosine"

    def distance(self, x1, x2):
        a = _as_float_array(x1)
        b = _as_float_array(x2, a.dtype)
        if a.ndim != 1 or b.ndim != 1:
            raise ValueError("Metric::distance takes two 1-D views")
        sfx = "f32" if a.dtype == np.float32 else "f64"
        return a.dtype.type(getattr(_lib.lib(), f"pn_cosine_{sfx}")(a.ctypes.data, a.shape[0], b.ctypes.data, b.shape[0]))

    def rdistance(self, x1, x2):  # src/distance.rs:109-112
        return self.distance(x1, x2)

    def rdistance_to_distance(self, d):  # src/distance.rs:113-116
        return d

    def distance_to_rdistance(self, d):  # src/distance.rs:118-121
        return d


def pairwise_device(x, out=None):
    """``distance::pairwise(x, &Euclidean)`` with rows and the n x n result in HBM (torch CUDA tensors; extension)."""
    import torch
    if not x.is_cuda or x.dim() != 2 or x.dtype not in (torch.float32, torch.float64) or (x.shape[1] > 1 and x.stride(1) != 1):
        raise ValueError("x must be a row-major 2-D float32/float64 CUDA tensor")
    n, d = x.shape
    if out is None:
        out = torch.empty((n, n), dtype=x.dtype, device=x.device)
    sfx = "f32" if x.dtype == torch.float32 else "f64"
    st = torch.cuda.current_stream(x.device).cuda_stream
    check(getattr(_lib.lib(), f"pn_pairwise_device_{sfx}")(x.data_ptr() if n * d else None, n, d,
                                                           x.stride(0) if n > 1 else max(d, 1), x.device.index or 0,
                                                           out.data_ptr(), C.c_void_p(st)))
    return out


def pairwise(x, metric=None, device: int = 0):
    """``distance::pairwise(x, &metric)`` (src/distance.rs:58-74) on the GPU."""
    if metric is not None and not isinstance(metric, (Euclidean, Cosine)):
        raise NotImplementedError("only the Euclidean and Cosine metrics are on the MI355X path")
    fn = "pn_pairwise_cosine" if isinstance(metric, Cosine) else "pn_pairwise"
    a = np.asarray(x)
    if a.ndim != 2:
        raise ValueError("pairwise takes a 2-D array")
    a = _as_float_array(a)
    n, d = a.shape
    sfx = "f32" if a.dtype == np.float32 else "f64"
    out = np.empty((n, n), dtype=a.dtype)
    rc = getattr(_lib.lib(), f"{fn}_{sfx}")(a.ctypes.data, n, d, d, device, out.ctypes.data)
    check(rc)
    return out
