"""Row-sharded exact k-NN over the GPUs of one node.

The exchange lives BEHIND the C ABI (``pn_sharded_*``, csrc/sharded.hip): row shards with global indices
(``PN_OPT_INDEX_BASE``), ONE ``ncclAllGather`` (RCCL over xGMI) of the packed per-GPU ``{idx | dist}`` top-k per
query batch, a HIP merge kernel ordered by (distance, index).  Exact top-k is decomposable -- top-k of a union is
top-k of the per-part top-k's -- and the order is total, so results are identical for any number of shards.

* ``ShardedIndex``     -- ctypes view of a ``pn_sharded`` handle: one process over several GPUs
                          (``from_host(points, devices)``) or one process per GPU (``from_rank_device``).
* ``ShardedBallTree``  -- ``BallTree``-shaped front for one-process-per-GPU programs (``torch.distributed`` carries
                          the 128-byte communicator id to the ranks, nothing else); its per-rank engine is
                          pluggable so the host logic is testable on CPU with ``gloo`` and an oracle-backed engine
                          (tests/test_sharded_gloo.py).  ``AbiShardEngine`` (default) hands the whole batch to
                          ``pn_sharded_query_device_f32``; ``HipShardEngine`` keeps the exchange in
                          ``torch.distributed`` (one ``all_gather_into_tensor``) for comparison.
There is no CPU engine in this package.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

ABSENT = np.uint64(0xFFFFFFFFFFFFFFFF)


def shard_bounds(n: int, world: int, rank: int):
    """Contiguous row range [lo, hi) of ``rank``: ceil(n / world) rows per shard; the same
    rule as csrc/sharded.hip."""
    per = (n + world - 1) // world
    lo = min(n, rank * per)
    hi = min(n, lo + per)
    return lo, hi


class ShardedIndex:
    """A ``pn_sharded`` handle (include/petal_mi355x.h): row shards + RCCL exchange behind the ABI."""

    def __init__(self, handle, keep=None, dtype=np.float32):
        from . import _lib
        from .errors import check
        self._h = C.c_void_p(handle)
        self._keep = keep
        self.dtype = np.dtype(dtype)   # float32 or float64: the element type of corpus, queries and distances
        self._sfx = "f64" if self.dtype == np.float64 else "f32"
        info = _lib.PnShardedInfo()
        check(_lib.lib().pn_sharded_info(self._h, C.byref(info)))
        self.n, self.dim = int(info.n_points), int(info.dim)
        self.n_shards, self.world, self.rank = int(info.n_shards), int(info.world), int(info.rank)
        self.local_first_row, self.local_rows = int(info.local_first_row), int(info.local_rows)
        self.mfma_eligible, self.bf16_eligible = bool(info.mfma_eligible), bool(info.bf16_eligible)

    @classmethod
    def from_host(cls, points, devices, metric=None):
        """``BallTree::new`` over ``len(devices)`` row shards driven by this process; shard g lives on
        ``devices[g]`` (a device may be named several times).  ``metric``: None / ``Euclidean()`` or ``Cosine()``."""
        from . import _lib
        from .errors import check
        a = np.asarray(points)
        if a.dtype not in (np.float32, np.float64):   # (an f64 array gives an f64 handle, like BallTree::new is generic over A)
            a = a.astype(np.float32)
        if a.ndim != 2:
            raise ValueError("points must be a 2-D array (Ix2)")
        n, d = a.shape
        item = a.itemsize
        rs, cs = (a.strides[0] // item, a.strides[1] // item) if n and d else (max(d, 1), 1)
        devs = (C.c_int * len(devices))(*[int(x) for x in devices])
        h = C.c_void_p(0)
        from .distance import Cosine
        sfx = "f64" if a.dtype == np.float64 else "f32"
        create = getattr(_lib.lib(), ("pn_sharded_create_cosine_" if isinstance(metric, Cosine) else "pn_sharded_create_") + sfx)
        check(create(a.ctypes.data if a.size else None, n, d, rs, cs if d > 1 else 1, devs, len(devices), C.byref(h)))
        return cls(h.value, keep=a, dtype=a.dtype)

    @staticmethod
    def unique_id() -> bytes:
        from . import _lib
        from .errors import check
        buf = C.create_string_buffer(_lib.PN_COMM_ID_BYTES)
        check(_lib.lib().pn_comm_unique_id(buf))
        return buf.raw

    @classmethod
    def from_rank_device(cls, rows, n_total: int, rank: int, world: int, comm_id: bytes, device: int, stream=None,
                         dtype=None):
        """One process per GPU: ``rows`` = this rank's shard (float32 or float64 CUDA tensor, or None for an empty
        shard).  ``dtype`` is the element type of the WHOLE job (every rank must pass the same one: the packed parts of
        the all-gather are sized by it); it defaults to the rows' type and must be given where a rank may hold no rows."""
        import torch
        from . import _lib
        from .errors import check
        h = C.c_void_p(0)
        if dtype is None:
            dtype = np.float64 if (rows is not None and rows.dtype == torch.float64) else np.float32
        dtype = np.dtype(dtype)
        if dtype not in (np.dtype(np.float32), np.dtype(np.float64)):
            raise ValueError("dtype must be float32 or float64")
        f64 = dtype == np.float64
        if rows is not None and rows.dtype != (torch.float64 if f64 else torch.float32):
            raise ValueError("rows are %s but the job's element type is %s" % (rows.dtype, dtype))
        if rows is not None and rows.numel():
            if rows.dtype not in (torch.float32, torch.float64) or rows.dim() != 2 or not rows.is_cuda or (rows.shape[1] > 1 and rows.stride(1) != 1):
                raise ValueError("rows must be a row-major 2-D float32 or float64 CUDA tensor")
            ptr, nl, d, ld = rows.data_ptr(), rows.shape[0], rows.shape[1], (rows.stride(0) if rows.shape[0] > 1 else max(rows.shape[1], 1))
            st = stream if stream is not None else torch.cuda.current_stream(rows.device).cuda_stream
        else:
            ptr, nl, d, ld, st = None, 0, (rows.shape[1] if rows is not None else 0), 1, 0
        idb = C.create_string_buffer(bytes(comm_id), _lib.PN_COMM_ID_BYTES)
        create = _lib.lib().pn_sharded_create_rank_device_f64 if f64 else _lib.lib().pn_sharded_create_rank_device_f32
        check(create(ptr, nl, d, ld, int(n_total), int(rank), int(world), idb, int(device), C.c_void_p(st), C.byref(h)))
        return cls(h.value, dtype=np.float64 if f64 else np.float32)

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            from . import _lib
            _lib.lib().pn_sharded_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, opt: int, value: int):
        from . import _lib
        from .errors import check
        check(_lib.lib().pn_sharded_set_option(self._h, opt, int(value)))
        return self

    def set_engine(self, name: str):
        from . import _lib
        from .ball_tree import ENGINES
        return self.set_option(_lib.PN_OPT_ENGINE, ENGINES[name])

    def stats(self, reset: bool = False):
        from . import _lib
        from .errors import check
        s = _lib.PnStats()
        check(_lib.lib().pn_sharded_get_stats(self._h, C.byref(s), int(reset)))
        return {k: getattr(s, k) for k, _ in s._fields_ if k != "reserved"}

    def num_points(self) -> int:
        return self.n

    def query_batch(self, queries, k: int):
        """host queries (nq, d) -> (idx uint64, dist of the handle's dtype) of shape (nq, min(k, n))"""
        from . import _lib
        from .errors import check
        a = np.ascontiguousarray(queries, dtype=self.dtype)
        if a.ndim != 2:
            raise ValueError("queries must be 2-D")
        nq, qc = a.shape
        kout = min(int(k), self.n)
        idx = np.empty((nq, kout), dtype=np.uint64)
        dist = np.empty((nq, kout), dtype=self.dtype)
        if nq and kout:
            check(getattr(_lib.lib(), "pn_sharded_query_" + self._sfx)(self._h, a.ctypes.data, nq, qc, max(qc, 1), int(k),
                                                                      idx.ctypes.data, dist.ctypes.data))
        return idx, dist

    def query(self, point, k: int):
        i, d = self.query_batch(np.asarray(point, dtype=self.dtype).reshape(1, -1), k)
        return i[0], d[0]

    def query_device(self, queries, k: int, out_idx=None, out_dist=None, stream=None):
        """queries / results in HBM (torch CUDA tensors), enqueued on the current stream"""
        import torch
        from . import _lib
        from .errors import check
        tdt = torch.float64 if self.dtype == np.float64 else torch.float32
        if queries.dtype != tdt or queries.dim() != 2 or not queries.is_cuda:
            raise ValueError("queries must be a 2-D CUDA tensor of the handle's dtype (%s)" % self.dtype)
        if queries.shape[1] > 1 and queries.stride(1) != 1:
            queries = queries.contiguous()
        nq, qc = queries.shape
        kout = min(int(k), self.n)
        if out_idx is None:
            out_idx = torch.empty((nq, kout), dtype=torch.int64, device=queries.device)
        if out_dist is None:
            out_dist = torch.empty((nq, kout), dtype=tdt, device=queries.device)
        if nq and kout:
            st = stream if stream is not None else torch.cuda.current_stream(queries.device).cuda_stream
            check(getattr(_lib.lib(), "pn_sharded_query_device_" + self._sfx)(
                self._h, queries.data_ptr(), nq, qc, queries.stride(0) if nq > 1 else max(qc, 1), int(k),
                out_idx.data_ptr(), out_dist.data_ptr(), C.c_void_p(st)))
        return out_idx, out_dist

    def query_radius_device(self, queries, distance, capacity: int, out_offsets=None, out_idx=None, out_total=None,
                            stream=None):
        """``pn_sharded_query_radius_device_*`` (one-shard handles): see ``BallTree.query_radius_device``."""
        import torch
        from . import _lib
        from .errors import check
        tdt = torch.float64 if self.dtype == np.float64 else torch.float32
        if queries.dtype != tdt or queries.dim() != 2 or not queries.is_cuda:
            raise ValueError("queries must be a 2-D CUDA tensor of the handle's dtype (%s)" % self.dtype)
        if queries.shape[1] > 1 and queries.stride(1) != 1:
            queries = queries.contiguous()
        nq, qc = queries.shape
        dev = queries.device
        offs = out_offsets if out_offsets is not None else torch.empty(nq + 1, dtype=torch.int64, device=dev)
        idx = out_idx if out_idx is not None else torch.empty(max(int(capacity), 1), dtype=torch.int64, device=dev)
        tot = out_total if out_total is not None else torch.empty(1, dtype=torch.int64, device=dev)
        st = stream if stream is not None else torch.cuda.current_stream(dev).cuda_stream
        r = C.c_double(distance) if self.dtype == np.float64 else C.c_float(distance)
        fn = getattr(_lib.lib(), f"pn_sharded_query_radius_device_{self._sfx}")
        check(fn(self._h, queries.data_ptr() if nq * qc else None, nq, qc, queries.stride(0) if nq > 1 else max(qc, 1), r,
                 offs.data_ptr(), idx.data_ptr(), int(capacity), tot.data_ptr(), C.c_void_p(st)))
        return offs, idx, tot

    def query_radius_batch(self, queries, distance):
        """CSR (offsets[nq+1], indices) of ``{ i : dist(q, p_i) < distance }`` over all shards, ascending per query."""
        from . import _lib
        from .errors import check
        a = np.ascontiguousarray(queries, dtype=self.dtype)
        nq, qc = a.shape
        offsets = np.zeros(nq + 1, dtype=np.uint64)
        out = C.c_void_p(0)
        r = C.c_double(distance) if self.dtype == np.float64 else C.c_float(distance)
        check(getattr(_lib.lib(), "pn_sharded_query_radius_" + self._sfx)(self._h, a.ctypes.data, nq, qc, max(qc, 1), r,
                                                                         offsets.ctypes.data, C.byref(out)))
        total = int(offsets[-1])
        try:
            idx = (np.frombuffer((C.c_uint64 * total).from_address(out.value), dtype=np.uint64).copy()
                   if total else np.empty(0, dtype=np.uint64))
        finally:
            if out.value:
                _lib.lib().pn_free(out)
        return offsets, idx


class AbiShardEngine:
    """Default engine of ``ShardedBallTree``: the whole batch -- local shard, all-gather, merge -- is ONE call into
    the C ABI (``pn_sharded_query_device_f32``); ``torch.distributed`` only carried the communicator id."""
    collective = True

    def __init__(self, device: int):
        self.device = device
        self.index = None

    def build_rank(self, rows, n_total, rank, world, comm_id, dtype=None):
        self.index = ShardedIndex.from_rank_device(rows, n_total, rank, world, comm_id, self.device, dtype=dtype)

    @property
    def tree(self):  # option / statistics target (bench.py)
        return self.index

    def query_all(self, queries, k: int):
        return self.index.query_device(queries, k)

    def radius_all(self, queries, r: float):
        q = queries.cpu().numpy() if hasattr(queries, "cpu") else np.asarray(queries)
        return self.index.query_radius_batch(q, r)

    def empty(self, shape, dtype):
        import torch
        return torch.empty(shape, dtype=dtype, device=f"cuda:{self.device}")


class HipShardEngine:
    """Local ``BallTree`` on the rank's GPU + the HIP merge kernel; the exchange stays in ``torch.distributed``."""

    def __init__(self, device: int):
        self.device = device
        self.tree = None

    def build(self, shard_rows, lo: int):
        from . import _lib
        from .ball_tree import BallTree
        import torch
        if isinstance(shard_rows, torch.Tensor):
            self.tree = BallTree.from_device(shard_rows)
        else:
            self.tree = BallTree.euclidean(np.asarray(shard_rows, dtype=np.float32), device=self.device)
        self.tree.set_option(_lib.PN_OPT_INDEX_BASE, lo)

    def query(self, queries, k: int):
        """queries: float32 CUDA tensor (nq, d) -> (idx int64, dist float32) CUDA tensors (nq, kout)."""
        return self.tree.query_device(queries, k)

    def packed_words(self, nq: int, kp: int) -> int:
        return nq * kp + (nq * kp + 1) // 2

    def query_packed(self, queries, k: int):
        """The shard's top-k written straight into the buffer the all-gather sends (layout of ``pack``): no staging
        tensors, no fill, no copies -- at 8 GPUs those small kernels were a tenth of a step.  Requires k <= rows."""
        import torch
        nq = queries.shape[0]
        buf = torch.empty(self.packed_words(nq, k), dtype=torch.int64, device=queries.device)
        idx = buf[: nq * k].view(nq, k)
        dst = buf[nq * k:].view(torch.float32)[: nq * k].view(nq, k)
        self.tree.query_device(queries, k, out_idx=idx, out_dist=dst)
        return buf

    def radius(self, queries, r: float):
        """the shard's CSR through the device entry point (pn_query_radius_device_*: counts, scan and fill in HBM); the
        host looks at the total once, to size the copy -- and calls again with room for it when the buffer was short"""
        nq = queries.shape[0]
        cap = max(4 * nq, 1024)
        while True:
            offs, idx, tot = self.tree.query_radius_device(queries, r, cap)
            total = int(tot.item())
            if total <= cap:
                break
            cap = total
        return offs.cpu().numpy().astype(np.uint64), idx[:total].cpu().numpy().astype(np.uint64)

    def merge(self, idx_parts, dist_parts, k_out: int):
        """parts: (G, nq, k_part) CUDA tensors (any part stride, rows contiguous) -> (nq, k_out)."""
        import torch
        from . import _lib
        from .errors import check
        g, nq, kp = idx_parts.shape
        assert idx_parts.stride(1) == kp and idx_parts.stride(2) == 1
        assert dist_parts.stride(1) == kp and dist_parts.stride(2) == 1
        out_i = torch.empty((nq, k_out), dtype=torch.int64, device=idx_parts.device)
        out_d = torch.empty((nq, k_out), dtype=torch.float32, device=idx_parts.device)
        st = torch.cuda.current_stream(idx_parts.device).cuda_stream
        check(_lib.lib().pn_merge_topk_device_f32(idx_parts.data_ptr(), dist_parts.data_ptr(), g,
                                                  idx_parts.stride(0) if g > 1 else nq * kp,
                                                  dist_parts.stride(0) if g > 1 else nq * kp, nq, kp, k_out,
                                                  out_i.data_ptr(), out_d.data_ptr(), self.device, C.c_void_p(st)))
        return out_i, out_d

    def pack(self, idx, dst):
        """One buffer per rank for ONE all-gather: int64 words [idx (nq*k) | dist bits (nq*k/2, padded)]."""
        import torch
        nq, kp = idx.shape
        words = nq * kp + (nq * kp + 1) // 2
        buf = torch.empty(words, dtype=torch.int64, device=idx.device)
        buf[: nq * kp] = idx.reshape(-1)
        buf[nq * kp:].view(torch.float32)[: nq * kp] = dst.reshape(-1)
        return buf

    def unpack(self, gathered, world, nq, kp):
        """gathered: (world * words,) int64 -> strided (world, nq, kp) views of indices and distances."""
        import torch
        words = nq * kp + (nq * kp + 1) // 2
        g = gathered.view(world, words)
        idx = g[:, : nq * kp].unflatten(1, (nq, kp))
        dst = g[:, nq * kp:].view(torch.float32)[:, : nq * kp].unflatten(1, (nq, kp))
        return idx, dst

    def to_backend(self, a):
        return a

    def empty(self, shape, dtype):
        import torch
        return torch.empty(shape, dtype=dtype, device=f"cuda:{self.device}")


class ShardedBallTree:
    """``BallTree``-shaped index over a row-sharded corpus.

    ``points_fn(lo, hi)`` returns this rank's rows (ndarray or CUDA tensor); the
    full corpus never has to exist in one place.
    """

    def __init__(self, n_points: int, points_fn, engine=None, group=None, dtype=np.float32):
        """``dtype``: element type of the whole job (float32 or float64), the same on every rank -- a rank without rows
        has nothing else to learn it from, and the exchange is sized by it; the ranks compare theirs before any
        communicator exists and all raise together on a mismatch."""
        import torch.distributed as dist
        self.dtype = np.dtype(dtype)
        self.dist = dist
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.n = int(n_points)
        self.lo, self.hi = shard_bounds(self.n, self.world, self.rank)
        if engine is None:
            import torch
            engine = AbiShardEngine(torch.cuda.current_device())
        self.engine = engine
        self.n_local = self.hi - self.lo
        if getattr(engine, "collective", False):
            # the ABI owns the exchange: all it needs from this program is the communicator id on every rank
            import torch
            rows = points_fn(self.lo, self.hi) if self.n_local > 0 else None
            tdt = torch.float64 if self.dtype == np.float64 else torch.float32
            if rows is not None and not isinstance(rows, torch.Tensor):
                rows = torch.from_numpy(np.ascontiguousarray(rows, dtype=self.dtype)).to(f"cuda:{engine.device}")
            mine = self.dtype.itemsize if rows is None or rows.dtype == tdt else -int(rows.element_size())
            # Failure-symmetric: rank 0 may fail to make the id (RCCL not loadable: PN_ERR_COMM) -- it then still takes
            # part in the broadcast, with a status byte in front of an all-zero id, and EVERY rank raises after it.  (A
            # rank 0 that raised before the broadcast left the others blocked in it.)
            cid = torch.zeros(129, dtype=torch.uint8)
            err = None
            if self.rank == 0:
                try:
                    cid[1:] = torch.frombuffer(bytearray(ShardedIndex.unique_id()), dtype=torch.uint8)
                    cid[0] = 1
                except Exception as e:  # noqa: BLE001
                    err = e
            if self.world > 1:
                on_gpu = dist.get_backend(group) == "nccl"
                t = cid.to(f"cuda:{engine.device}") if on_gpu else cid
                dist.broadcast(t, src=0, group=group)
                cid = t.cpu()
            if int(cid[0]) != 1:
                raise err if err is not None else RuntimeError("rank 0 could not create the RCCL communicator id")
            # element sizes agreed BEFORE the communicator is made: ranks that disagree would size the all-gather
            # differently (1.5 vs 2 words per entry) and hang or corrupt it.  Every rank sees every size and raises.
            sizes = [mine]
            if self.world > 1:
                on_gpu = dist.get_backend(group) == "nccl"
                t = torch.tensor([mine], dtype=torch.int64)
                t = t.to(f"cuda:{engine.device}") if on_gpu else t
                parts = [torch.empty_like(t) for _ in range(self.world)]
                dist.all_gather(parts, t, group=group)
                sizes = [int(x.cpu()[0]) for x in parts]
            if any(sz != self.dtype.itemsize for sz in sizes):
                raise ValueError("ShardedBallTree: the ranks do not agree on the element type (bytes per element by rank: "
                                 "%s, negative = rows of another type than dtype=%s)" % (sizes, self.dtype))
            engine.build_rank(rows, self.n, self.rank, self.world, bytes(cid[1:].numpy().tobytes()), dtype=self.dtype)
        elif self.n_local > 0:
            self.engine.build(points_fn(self.lo, self.hi), self.lo)

    def num_points(self) -> int:
        return self.n

    def query_batch(self, queries, k: int):
        """All ranks pass the SAME queries; every rank returns the full (nq, min(k, n)) answer."""
        import torch
        nq = queries.shape[0]
        k_out = min(int(k), self.n)
        if k_out == 0 or nq == 0:
            return (self.engine.empty((nq, 0), torch.int64), self.engine.empty((nq, 0), torch.float32))
        if getattr(self.engine, "collective", False):
            return self.engine.query_all(queries, k_out)
        if self.world == 1:
            return self.engine.query(queries, k_out)
        k_part = min(int(k), max(shard_bounds(self.n, self.world, 0)[1], 1))  # largest shard
        if self.n_local >= k_part and hasattr(self.engine, "query_packed"):
            mine = self.engine.query_packed(queries, k_part)
        else:  # a shard with fewer than k_part rows (or none): absent slots are marked
            idx = self.engine.empty((nq, k_part), torch.int64)
            dst = self.engine.empty((nq, k_part), torch.float32)
            idx.fill_(-1)  # 0xFFFF...: absent
            dst.fill_(float("nan"))
            if self.n_local > 0:
                li, ld = self.engine.query(queries, k_part)
                kl = li.shape[1]
                idx[:, :kl] = li
                dst[:, :kl] = ld
            mine = self.engine.pack(idx, dst)
        # the one exchange step of the path: ONE all-gather of the packed per-shard top-k (RCCL over xGMI)
        gathered = self.engine.empty((self.world * mine.numel(),), torch.int64)
        self.dist.all_gather_into_tensor(gathered, mine, group=self.group)
        g_idx, g_dst = self.engine.unpack(gathered, self.world, nq, k_part)
        return self.engine.merge(g_idx, g_dst, k_out)

    def query(self, point, k: int):
        i, d = self.query_batch(point.reshape(1, -1), k)
        return i[0], d[0]

    def query_radius_batch(self, queries, r: float):
        """CSR (offsets, indices) on every rank; shards are ascending row ranges, so
        concatenating per-shard ascending lists in rank order is globally ascending."""
        import torch
        nq = queries.shape[0]
        if getattr(self.engine, "collective", False):
            return self.engine.radius_all(queries, r)
        if self.n_local > 0:
            off, ids = self.engine.radius(queries, r)
        else:
            off, ids = np.zeros(nq + 1, dtype=np.uint64), np.empty(0, dtype=np.uint64)
        if self.world == 1:
            return off, ids
        counts = torch.from_numpy(np.diff(off.astype(np.int64)))
        all_counts = [torch.empty_like(counts) for _ in range(self.world)]
        dev = self.engine.empty((1,), torch.int64).device
        counts_d = counts.to(dev)
        all_counts = [torch.empty_like(counts_d) for _ in range(self.world)]
        self.dist.all_gather(all_counts, counts_d, group=self.group)
        totals = [int(c.sum().item()) for c in all_counts]
        mx = max(max(totals), 1)
        pad = torch.full((mx,), -1, dtype=torch.int64, device=dev)
        pad[: len(ids)] = torch.from_numpy(ids.astype(np.int64)).to(dev)
        all_ids = [torch.empty_like(pad) for _ in range(self.world)]
        self.dist.all_gather(all_ids, pad, group=self.group)
        cnt = torch.stack(all_counts).cpu().numpy()  # (G, nq)
        per_q = cnt.sum(axis=0)
        offsets = np.zeros(nq + 1, dtype=np.uint64)
        offsets[1:] = np.cumsum(per_q).astype(np.uint64)
        out = np.empty(int(offsets[-1]), dtype=np.uint64)
        starts = np.zeros((self.world, nq + 1), dtype=np.int64)
        starts[:, 1:] = np.cumsum(cnt, axis=1)
        host_ids = [t.cpu().numpy().astype(np.uint64) for t in all_ids]
        for a in range(nq):
            o = int(offsets[a])
            for g in range(self.world):
                c = int(cnt[g, a])
                if c:
                    out[o:o + c] = host_ids[g][starts[g, a]:starts[g, a] + c]
                    o += c
        return offsets, out
