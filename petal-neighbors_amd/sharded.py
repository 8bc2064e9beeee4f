"""Row-sharded exact k-NN over the GPUs of one node.

One process per GPU (``torch.distributed``; backend ``nccl`` = RCCL over xGMI on
ROCm, ``gloo`` in the CPU tests).  The corpus is split by contiguous row ranges;
every rank sees all queries, answers them on its shard with global indices
(``PN_OPT_INDEX_BASE``), and ONE all-gather of the per-shard ``(idx, dist)``
top-k per batch feeds a local merge ordered by (distance, index).  Exact top-k
is decomposable -- top-k of a union is top-k of the per-part top-k's -- and the
(distance, index) order is total, so results are identical at any world size.

The per-rank engine is pluggable so the host logic (bounds, padding, gather,
merge call pattern) is testable on CPU with ``gloo``; the default engine is the
HIP path and there is no CPU engine in this package.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

ABSENT = np.uint64(0xFFFFFFFFFFFFFFFF)


def shard_bounds(n: int, world: int, rank: int):
    """Contiguous row range [lo, hi) of ``rank``: ceil(n / world) rows per shard."""
    per = (n + world - 1) // world
    lo = min(n, rank * per)
    hi = min(n, lo + per)
    return lo, hi


class HipShardEngine:
    """Default engine: local ``BallTree`` on the rank's GPU + the HIP merge kernel."""

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
        return self.tree.query_radius_batch(queries.cpu().numpy(), r)

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

    def __init__(self, n_points: int, points_fn, engine=None, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.n = int(n_points)
        self.lo, self.hi = shard_bounds(self.n, self.world, self.rank)
        if engine is None:
            import torch
            engine = HipShardEngine(torch.cuda.current_device())
        self.engine = engine
        self.n_local = self.hi - self.lo
        if self.n_local > 0:
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
