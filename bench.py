#!/usr/bin/env python3
"""bench.py -- exact k-NN queries/sec on MI355X, with roofline and CPU baseline.

    python bench.py --gpus N --steps K --warmup W          (N > 1: starts its own N ranks, one per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[1], the configuration the metric is quoted on):
1 000 000 points x 128 dims f32, 10 000 queries, k = 10, synthetic uniform [0,1)
generated in HBM (counter hash, seeds 0x5EED0001 / 0x5EED0002).  A "step" is one
pass of the hot path over the whole query batch: filter/scan kernel, exact
re-rank + selection, and (N > 1) the all-gather + merge of per-shard top-k --
enqueued through the C ABI's asynchronous entry points (pn_query_device_f32, or
pn_sharded_query_device_f32, which owns the RCCL all-gather; --comm torch keeps
the exchange in torch.distributed instead).  Inputs are resident in HBM when the
timed region starts.  With N > 1 the corpus is row-sharded over the ranks (total
work fixed: strong scaling).  --config picks another shape (CONFIGS below; c1 = the
reference's own bench shapes, one point per call: the plumbing).

Prints ONE JSON line on rank 0 (contract in the task statement), including
  roofline     -- dominant kernel: algorithmic flops (2*N*D per query, SURVEY.md
                  8d) / its average hipEvent duration, against the MFMA peak of
                  the dtype the kernel contracts in (bf16 first tier: 2516.8 TFLOP/s);
                  traffic = HBM bytes per launch from the committed rocprofv3 --pmc
                  record of the same configuration (profiles/*pmc*.json), else null;
  verified     -- after the timed region the step's answers are checked (order,
                  range, duplicates, distances re-derived by the scalar metric, a
                  sample against the oracle's brute force and 2048 queries against
                  the exact engine); --no-verify skips it (profiling runs);
  cpu_baseline -- the CPU oracle's faithful ball tree ("port": the reference is
                  Rust and cannot be built here) timed on this host's cores.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

CONFIGS = {
    # name: (n_points, dim, n_queries, k)
    "c2": (1_000_000, 128, 10_000, 10),        # BASELINE.json configs[1] (headline)
    "c3": (10_000_000, 128, 100_000, 100),     # configs[2]
    "c4": (10_000_000, 768, 100_000, 10),      # configs[3] (8 GPUs in BASELINE; fits one here)
    "c5": (100_000_000, 96, 1_000_000, 10),    # configs[4]
    "tiny": (20_000, 128, 512, 10),
    # reduced-scale shapes of the other BASELINE configs (same D and k, 1M rows, 10k queries)
    "c3s": (1_000_000, 128, 10_000, 100),
    "c4s": (1_000_000, 768, 10_000, 10),
    "c5s": (1_000_000, 96, 10_000, 10),
    "c5shard": (12_500_000, 96, 1_000_000, 10),  # one of the 8 row shards of configs[4]
    "c2k1": (1_000_000, 128, 10_000, 1),         # query_nearest at the headline shape
    "c2s2": (500_000, 128, 10_000, 10),          # one shard of C2 at 2 / 4 / 8 GPUs
    "c2s4": (250_000, 128, 10_000, 10),
    "c2s8": (125_000, 128, 10_000, 10),
    "c5m": (1_000_000, 96, 200_000, 10),         # many queries: one workgroup per query tile
    "c3m": (1_000_000, 128, 200_000, 100),
    "c3k10": (10_000_000, 128, 100_000, 10),     # configs[2]'s corpus and batch at k = 10 (grid in rounds, short buffers)
    "c2k500": (1_000_000, 128, 10_000, 500),     # large k at the headline shape (the tier's limits)
    "c2k1000": (1_000_000, 128, 10_000, 1000),
    "d64": (1_000_000, 64, 10_000, 10),          # narrower rows (kernel experiments)
    "d64k100": (1_000_000, 64, 10_000, 100),
}
PEAK_F32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
PEAK_BF16_MFMA_TFLOPS = 2516.8  # same table: dense BF16 MFMA = 16 x the f32 matrix rate ("~2.5 PF dense")
SEED_P, SEED_Q = 0x5EED0001, 0x5EED0002


def usable_cores():
    """CPU threads this process may actually run: affinity mask, capped by the cgroup CPU quota."""
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    cores = min(cores, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    cores = min(cores, max(1, q // per))
            break
        except Exception:
            continue
    return max(1, min(cores, 64))


def cpu_baseline(n, dim, k, gpu_out=None, budget_queries=None):
    """Faithful ball tree (oracle/oracle_impl.h) on this host: single-thread QPS over >= 64 queries (what
    benches/ball_tree.rs:43-62 times: the reference is single-threaded), all-core QPS with one query per thread over
    >= 8 queries per thread (what `Euclidean: Sync`, src/distance.rs:19, lets a caller do), build s -- SURVEY.md 8(d).

    The tree's queries are the FIRST rows of the GPU batch (same seed, same counters), and its answers are kept: with
    gpu_out = (idx, dist) of the GPU step they are compared entry by entry (oracle.compare_knn: distances bit for bit,
    indices outside groups of exactly equal distances) -- the reference's pruned walk (src/ball_tree.rs:203-243, pruning
    on fl(|q - c| - R), :473-481) against the GPU pipeline on the same f32 data at the headline scale, not only each of
    them against the brute force."""
    import oracle
    oracle.build()
    cores = usable_cores()
    n_cpu = min(n, 1_000_000)
    pts = oracle.fill_uniform(n_cpu * dim, SEED_P).reshape(n_cpu, dim)
    par = max(0, min(4, int(np.log2(max(cores, 1)))))
    t0 = time.perf_counter()
    tree = oracle.Tree(pts, build_threads_log2=par)
    t_build = time.perf_counter() - t0
    nq1 = 64
    nqa = budget_queries or max(8 * cores, 64)
    qs = oracle.fill_uniform((nq1 + nqa) * dim, SEED_Q).reshape(nq1 + nqa, dim)
    t0 = time.perf_counter()
    i1, d1 = tree.query_batch(qs[:nq1], k, nthreads=1)
    t1 = time.perf_counter() - t0
    t0 = time.perf_counter()
    ia, da = tree.query_batch(qs[nq1:], k, nthreads=cores)
    ta = time.perf_counter() - t0
    res = {
        "value": round(nqa / ta, 3), "unit": "queries/s", "cores": cores, "kind": "port",
        "single_thread_qps": round(nq1 / t1, 3), "single_thread_queries": nq1, "build_s": round(t_build, 2),
        "sample": (f"C restatement of petal-neighbors' BallTree (oracle/), {n_cpu}x{dim} f32 corpus"
                   f"{'' if n_cpu == n else ' (sub-sampled from %d rows)' % n}, k={k}: {nqa} queries on {cores} threads "
                   f"(one query per thread, {nqa // cores} per thread), {nq1} queries on 1 thread; tree build "
                   f"({2**par} threads) {t_build:.1f} s excluded"),
    }
    if gpu_out is not None and n_cpu == n and gpu_out[0].shape[0] >= nq1 + nqa:
        m = nq1 + nqa
        gi = gpu_out[0][:m].cpu().numpy().astype(np.uint64)
        gd = gpu_out[1][:m].cpu().numpy()
        cmp = oracle.compare_knn(np.concatenate([i1, ia]), np.concatenate([d1, da]), gi, gd)
        res["agrees_with_gpu"] = bool(cmp["agree"])
        res["agreement"] = {**cmp, "what": (f"faithful ball-tree walk vs the GPU step's answers on the first {m} queries of "
                                            f"the batch: distances bit for bit, indices outside exact-tie groups")}
    return res


def verify(out, queries, n, dim, nq, k, rank, world, index, gen):
    """Parity leg, OUTSIDE the timed region, on the results of the last timed step (every rank holds the full answer):
    properties on every query (ascending, in range, no duplicates); the returned distance re-derived by the scalar
    metric for sampled (query, neighbour) pairs, rows regenerated from the counter hash; the CPU oracle's brute force
    over a host-generated corpus on a sample of queries (corpora of at most 2M rows); at one GPU also bit-equality with
    the exact engine on up to 2 048 queries."""
    import threading
    import oracle
    import petal_neighbors_amd as pn
    idx, dst = out
    res = {"ok": True, "checks": []}

    def chk(name, ok):
        res["checks"].append(name if ok else name + ": FAILED")
        res["ok"] = res["ok"] and bool(ok)

    kk = idx.shape[1]
    chk("ascending", bool((dst[:, 1:] >= dst[:, :-1]).all()) if kk > 1 else True)
    chk("in range", bool(((idx >= 0) & (idx < n)).all()))
    srt = torch.sort(idx, dim=1).values
    chk("no duplicates", bool((srt[:, 1:] != srt[:, :-1]).all()) if kk > 1 else True)
    # scalar metric on sampled pairs: rows come from the generator, not from any shard
    m = pn.distance.Euclidean()
    sel = np.linspace(0, nq - 1, 24).astype(np.int64)
    qh, ih, dh = queries[sel].cpu().numpy(), idx[sel].cpu().numpy(), dst[sel].cpu().numpy()
    ok = True
    for a in range(len(sel)):
        for j in (0, kk - 1):
            row = oracle.fill_uniform(dim, SEED_P, int(ih[a, j]) * dim)
            ok = ok and m.distance(qh[a], row).tobytes() == dh[a, j].tobytes()
    chk("distance == Euclidean::distance(query, points[idx]) on 48 pairs", ok)
    if n <= 2_000_000 and rank == 0:
        pts = oracle.fill_uniform(n * dim, SEED_P).reshape(n, dim)
        ns = 32
        osel = np.linspace(0, nq - 1, ns).astype(np.int64)
        oq = queries[osel].cpu().numpy()
        want = [None] * ns
        nth = min(usable_cores(), 16)

        def work(t):
            for a in range(t, ns, nth):
                want[a] = oracle.brute_knn(pts, oq[a:a + 1], k)
        ths = [threading.Thread(target=work, args=(t,)) for t in range(nth)]
        [t.start() for t in ths]
        [t.join() for t in ths]
        gi, gd = idx[osel].cpu().numpy().astype(np.uint64), dst[osel].cpu().numpy()
        ok = all(np.array_equal(gi[a], want[a][0][0]) and gd[a].tobytes() == want[a][1][0].tobytes() for a in range(ns))
        chk(f"oracle brute force on {ns} sampled queries", ok)
    if world == 1 and index.engine.tree is not None:
        tree = index.engine.tree
        ne = min(nq, 2048)
        esel = torch.linspace(0, nq - 1, ne, device=idx.device).long()
        tree.set_engine("exact")
        ei, ed = (tree.query_device(queries[esel].contiguous(), k))
        torch.cuda.synchronize()
        chk(f"exact engine on {ne} queries", torch.equal(ei, idx[esel]) and torch.equal(ed.view(torch.int32), dst[esel].view(torch.int32)))
    return res


def verify_radius(out, q_host, n, dim, nq, radius, tree):
    """Parity leg of --mode radius, outside the timed region: every list ascending, in range and free of duplicates;
    the lists of sampled queries against the exact engine (the two-pass CSR scan under Euclidean::distance) and, for
    corpora of at most 2M rows, against the oracle's brute force over a host-generated corpus; every returned
    (query, row) pair of the sample re-derived with the scalar metric: distance < radius."""
    import oracle
    import petal_neighbors_amd as pn
    off, ids = out
    res = {"ok": True, "checks": []}

    def chk(name, ok):
        res["checks"].append(name if ok else name + ": FAILED")
        res["ok"] = res["ok"] and bool(ok)

    chk("offsets monotone, total = len(ids)", bool((np.diff(off.astype(np.int64)) >= 0).all()) and int(off[-1]) == len(ids))
    chk("in range", bool((ids < n).all()) if len(ids) else True)
    asc = True
    for a in np.linspace(0, nq - 1, min(nq, 4096)).astype(np.int64):
        l = ids[int(off[a]):int(off[a + 1])].astype(np.int64)
        asc = asc and bool((np.diff(l) > 0).all())
    chk("lists strictly ascending (sampled)", asc)
    sel = np.linspace(0, nq - 1, 48).astype(np.int64)
    tree.set_engine("exact")
    eo, ei = tree.query_radius_batch(q_host[sel], radius)
    tree.set_engine("auto")
    chk("exact engine on 48 queries", all(np.array_equal(ids[int(off[a]):int(off[a + 1])], ei[int(eo[j]):int(eo[j + 1])])
                                        for j, a in enumerate(sel)))
    m = pn.distance.Euclidean()
    ok, pairs = True, 0
    for a in sel:
        for row in ids[int(off[a]):int(off[a + 1])][:4]:
            p = oracle.fill_uniform(dim, SEED_P, int(row) * dim)
            ok = ok and bool(m.distance(q_host[a], p) < np.float32(radius))
            pairs += 1
    chk(f"Euclidean::distance(query, points[idx]) < r on {pairs} returned pairs", ok)
    if n <= 2_000_000:
        pts = oracle.fill_uniform(n * dim, SEED_P).reshape(n, dim)
        osel = sel[:16]
        chk("oracle brute force on 16 sampled queries",
            all(np.array_equal(ids[int(off[a]):int(off[a + 1])], oracle.brute_radius(pts, q_host[a], np.float32(radius)))
                for a in osel))
    return res


def committed_traffic(kernel_name, config_key):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc summary of this same command
    (profiles/*pmc*.json: counters cannot be read in-process), or (None, None)."""
    traffic, src = None, None
    try:
        import glob
        for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc*.json"))):
            pmc = json.load(open(f))
            if pmc.get("kernel") == kernel_name and pmc.get("config") == config_key:
                traffic, src = pmc["hbm_bytes_per_launch"], os.path.relpath(f, ROOT)
    except Exception:
        pass
    return traffic, src


def committed_kernel_split(config_key):
    """Mean durations of the dominant kernel's two launches of a step (scout-only launch, main launch) from the committed
    rocprofv3 --kernel-trace --stats record of this same command (profiles/*kernel_split*.json, written by
    tools/kernel_split.py from the stats CSV): the library brackets both launches with ONE pair of hipEvents (every event
    record costs the stream ~6 us), so the live number is their sum, kernel_ms_per_step."""
    try:
        import glob
        best = None
        for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*kernel_split*.json"))):
            d = json.load(open(f))
            if d.get("config") == config_key:
                best = (d, os.path.relpath(f, ROOT))
        return best
    except Exception:
        return None


def bench_f64(args):
    """--dtype f64: the same k-NN step on an f64 index (the reference is generic over A and its own harness is f64,
    benches/ball_tree.rs:9-13): the bf16 MFMA filter over the f64 corpus' tile images, then the f64 re-rank, proof and
    second tier.  One GPU; corpus built through the host API (f64 rows have no device-side generator); queries and
    results resident in HBM, one pn_query_device_f64 call per step."""
    import oracle
    import petal_neighbors_amd as pn
    from petal_neighbors_amd import _lib
    torch.cuda.set_device(0)
    n, dim, nq, k = CONFIGS[args.config]
    cosine = args.metric == "cosine"
    f64 = args.dtype == "f64"
    if f64:  # f64 coordinates with more than 24 significant bits: two f32 draws, the second scaled by 2^-24
        pts = (oracle.fill_uniform(n * dim, SEED_P).astype(np.float64) +
               oracle.fill_uniform(n * dim, SEED_P + 7).astype(np.float64) * 2.0 ** -24).reshape(n, dim)
        qs = (oracle.fill_uniform(nq * dim, SEED_Q).astype(np.float64) +
              oracle.fill_uniform(nq * dim, SEED_Q + 7).astype(np.float64) * 2.0 ** -24).reshape(nq, dim)
    else:
        pts = oracle.fill_uniform(n * dim, SEED_P).reshape(n, dim)
        qs = oracle.fill_uniform(nq * dim, SEED_Q).reshape(nq, dim)
    if cosine:  # uniform [-0.5, 0.5): directions spread over the whole sphere (all-positive data sits in a narrow cone)
        pts = (pts - pts.dtype.type(0.5)).astype(pts.dtype)
        qs = (qs - qs.dtype.type(0.5)).astype(qs.dtype)
    tree = pn.BallTree.new(pts, pn.distance.Cosine()) if cosine else pn.BallTree.euclidean(pts)
    tree.set_engine(args.engine)
    tree.set_option(_lib.PN_OPT_PROFILE, 1)
    if args.seed_model >= 0:
        tree.set_option(_lib.PN_OPT_SEED_MODEL, args.seed_model)
    if args.waves:
        tree.set_option(_lib.PN_OPT_BF16_WAVES, args.waves)
    qd = torch.from_numpy(qs).to("cuda:0")
    out_idx = torch.empty((nq, min(k, n)), dtype=torch.int64, device="cuda:0")
    out_dst = torch.empty((nq, min(k, n)), dtype=torch.float64 if f64 else torch.float32, device="cuda:0")
    for _ in range(args.warmup):
        tree.query_device(qd, k, out_idx, out_dst)
    torch.cuda.synchronize()
    tree.stats(reset=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        tree.query_device(qd, k, out_idx, out_dst)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    st = tree.stats()
    verified = None
    if not args.no_verify:
        verified = {"ok": True, "checks": []}

        def chk(name, ok):
            verified["checks"].append(name if ok else name + ": FAILED")
            verified["ok"] = verified["ok"] and bool(ok)
        gi, gd = out_idx.cpu().numpy(), out_dst.cpu().numpy()
        chk("ascending", bool((gd[:, 1:] >= gd[:, :-1]).all()) if gd.shape[1] > 1 else True)
        if cosine:
            # the oracle's scalar Cosine::distance over a shortlist of the 400 rows of largest f64 cosine (the float
            # result is within 1e-4 of the real cosine; the shortlist's tail is far beyond the k-th), (distance, index) order
            sel = np.linspace(0, nq - 1, 16).astype(np.int64)
            pn64 = pts.astype(np.float64)
            pn64 /= np.linalg.norm(pn64, axis=1, keepdims=True)
            ok = True
            for a in sel:
                c = pn64 @ (qs[a].astype(np.float64) / np.linalg.norm(qs[a].astype(np.float64)))
                short = np.argpartition(-c, 400)[:400]
                d = np.array([oracle.cosine(qs[a], pts[i]) for i in short], dtype=pts.dtype)
                order = np.lexsort((short, d))[:k]
                ok = ok and np.array_equal(gi[a].astype(np.int64), short[order]) and gd[a].tobytes() == d[order].tobytes()
            del pn64
            chk("oracle Cosine::distance on 16 sampled queries (shortlist of 400 by f64 cosine)", ok)
        else:
            sel = np.linspace(0, nq - 1, 32).astype(np.int64)
            wi, wd = oracle.brute_knn(pts, qs[sel], k)
            chk("oracle brute force on 32 sampled queries",
                np.array_equal(gi[sel].astype(np.uint64), wi) and gd[sel].tobytes() == wd.tobytes())
        ne = min(nq, 1024)
        esel = torch.linspace(0, nq - 1, ne, device="cuda:0").long()
        tree.set_engine("exact")
        t_e0 = time.perf_counter()
        ei, ed = tree.query_device(qd[esel].contiguous(), k)
        torch.cuda.synchronize()
        exact_ms_per_query = (time.perf_counter() - t_e0) * 1e3 / ne
        iv = torch.int64 if f64 else torch.int32
        chk(f"exact engine on {ne} queries", torch.equal(ei, out_idx[esel]) and
            torch.equal(ed.view(iv), out_dst[esel].view(iv)))
    launches = max(int(st["hot_launches"]), 1)
    hot_ms = st["hot_ms"] / launches
    flops_per_launch = 2.0 * n * dim * nq * args.steps / launches
    achieved = flops_per_launch / (hot_ms * 1e-3) / 1e12 if hot_ms > 0 else 0.0
    bf = args.engine in ("auto", "bf16") and tree.bf16_eligible
    peak = PEAK_BF16_MFMA_TFLOPS if bf else PEAK_F32_MFMA_TFLOPS
    ms_per_step = elapsed / args.steps * 1e3
    kernel_name = ("bf16_wide_kernel" if dim > 128 else "bf16_filter_kernel") if bf else "exact_knn_kernel"
    tag = ("_f64" if f64 else "") + ("_cosine" if cosine else "")
    traffic, traffic_src = committed_traffic(kernel_name, args.config + tag)
    ty = "fp64" if f64 else "fp32"
    line = {
        "metric": f"exact {'Cosine ' if cosine else ''}k-NN queries/sec ({n} x {dim} {ty}, k={k})",
        "value": round(nq * args.steps / elapsed, 1),
        "unit": "queries/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64" if f64 else "f32", "data": "synthetic",
        "config": {"workload": f"{args.config}{tag.replace('_', ' ')}: {n} points x {dim} dims {ty}, {nq} queries, k={k}, "
                               + ("uniform[-0.5,0.5), BallTree::new(points, Cosine)" if cosine else "uniform[0,1) with 48 random bits"),
                   "n_points": n, "dim": dim, "n_queries": nq, "k": k, "metric": "Cosine" if cosine else "Euclidean",
                   "engine": (f"bf16 filter + {'Cosine::distance' if cosine else ty} re-rank") if bf else "exact scan"},
        "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                     "frac": round(achieved / peak, 4), "traffic": traffic,
                     **({"traffic_unit": f"HBM bytes per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, {traffic_src})"}
                        if traffic_src else {}),
                     "kernel": kernel_name,
                     "mfma_dtype": "bf16" if bf else "none (f64 vector unit)",
                     "kernel_ms_per_step": round(st["hot_ms"] / max(args.steps, 1), 4),
                     "flops_per_step": 2.0 * n * dim * nq, "launches_per_step": round(launches / max(args.steps, 1), 2)},
        "verified": bool(verified["ok"]) if verified else None, "verify": verified,
        "fallback_queries": int(st["fallback_queries"]),
        "candidates_per_query": round(st["candidates"] / max(st["queries"], 1), 2),
        "exact_evaluations_per_query": round(st["evaluations"] / max(st["queries"], 1), 2),
    }
    if not args.no_verify:  # the exact scan on the verification sample, for scale (one un-warmed call: an upper bound)
        line["exact_scan_ms_per_query_1024"] = round(exact_ms_per_query, 5)
    emit(line)


def plumbing(args):
    """--config c1: the reference's own harness shapes and call pattern (benches/ball_tree.rs:8-62) -- f64, one point
    per call through the host API, queries = corpus rows -- plus BASELINE.json's configs[0] shape (1000 x 3, k = 2) and
    the one-point-per-call latency at the headline shape.  GPU path and the CPU restatement side by side."""
    import oracle
    import petal_neighbors_amd as pn
    oracle.build()
    torch.cuda.set_device(0)
    rng = np.random.default_rng(0xBA11)
    res = {}

    def timeit(fn, reps):
        fn()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        return (time.perf_counter() - t0) / reps

    # build: 128 x 10 f64, BallTree::euclidean(array) per iteration (benches/ball_tree.rs:8-20)
    a = rng.random((128, 10))
    res["build_128x10_f64"] = {"gpu_us": round(timeit(lambda: pn.BallTree.euclidean(a).close(), 20) * 1e6, 1),
                               "cpu_port_us": round(timeit(lambda: oracle.Tree(a), 200) * 1e6, 1)}
    # query: 64 x 10 f64, 64 calls per iteration, k = 5 (benches/ball_tree.rs:43-62)
    a = rng.random((64, 10))
    g, c = pn.BallTree.euclidean(a), oracle.Tree(a)
    res["query_64x10_f64_k5"] = {"gpu_us_per_call": round(timeit(lambda: [g.query(a[i], 5) for i in range(64)], 20) / 64 * 1e6, 1),
                                 "cpu_port_us_per_call": round(timeit(lambda: [c.query(a[i], 5) for i in range(64)], 50) / 64 * 1e6, 2)}
    for i in range(64):  # parity on the way
        gi, gd = g.query(a[i], 5)
        ci, cd = c.query(a[i], 5)
        assert gd.tobytes() == cd.tobytes() and list(gi) == list(ci)
    # query_radius: 64 x 10 f64, 64 calls per iteration, r = 0.2 (benches/ball_tree.rs:22-41)
    res["radius_64x10_f64_r0.2"] = {"gpu_us_per_call": round(timeit(lambda: [g.query_radius(a[i], 0.2) for i in range(64)], 20) / 64 * 1e6, 1),
                                    "cpu_port_us_per_call": round(timeit(lambda: [c.query_radius(a[i], 0.2) for i in range(64)], 50) / 64 * 1e6, 2)}
    for i in range(64):
        assert sorted(int(x) for x in g.query_radius(a[i], 0.2)) == sorted(int(x) for x in c.query_radius(a[i], 0.2))
    # BASELINE.json configs[0] as written: 1000 x 3 f64, k = 2
    a = rng.random((1000, 3))
    g, c = pn.BallTree.euclidean(a), oracle.Tree(a)
    res["query_1000x3_f64_k2"] = {"gpu_us_per_call": round(timeit(lambda: [g.query(a[i], 2) for i in range(100)], 10) / 100 * 1e6, 1),
                                  "cpu_port_us_per_call": round(timeit(lambda: [c.query(a[i], 2) for i in range(100)], 20) / 100 * 1e6, 2),
                                  "gpu_us_per_query_batched": round(timeit(lambda: g.query_batch(a, 2), 20) / 1000 * 1e6, 3)}
    # one point per call at the headline shape: 1M x 128 f32, k = 10
    L = pn._lib.lib()
    pts = torch.empty((1_000_000, 128), dtype=torch.float32, device="cuda:0")
    assert L.pn_fill_uniform_device_f32(pts.data_ptr(), pts.numel(), SEED_P, 0, 0, None) == 0
    torch.cuda.synchronize()
    big = pn.BallTree.from_device(pts)
    q = oracle.fill_uniform(64 * 128, SEED_Q).reshape(64, 128)
    res["query_1Mx128_f32_k10_one_point_per_call"] = {
        "gpu_us_per_call": round(timeit(lambda: [big.query(q[i], 10) for i in range(64)], 5) / 64 * 1e6, 1),
        "cpu_port_single_thread_qps_see_default_config": None}
    line = {"metric": "one point per call, host API (benches/ball_tree.rs shapes; microseconds per call)",
            "value": round(1e6 / res["query_64x10_f64_k5"]["gpu_us_per_call"], 1), "unit": "calls/s", "n_gpus": 1,
            "steps": args.steps, "warmup": args.warmup, "higher_is_better": True, "scaling": "n/a", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "c1: plumbing shapes of benches/ball_tree.rs (build 128x10, query 64x10 k=5, "
                                   "radius 64x10 r=0.2) + BASELINE.json configs[0] (1000x3 k=2) + nq=1 at the headline shape"},
            "results": res, "cpu_baseline": {"kind": "port", "cores": 1,
                                             "sample": "C restatement of petal-neighbors' BallTree (oracle/), same arrays, one thread"}}
    emit(line)


# The contract is ONE JSON line on stdout.  Native libraries write there too (RCCL prints its version banner to fd 1 when
# the first communicator is created), so for the life of the process fd 1 is pointed at stderr and the line goes to the
# saved descriptor of the real stdout.
_REAL_STDOUT = None


def _keep_stdout_for_the_line():
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)


def emit(line):
    data = (json.dumps(line) + "\n").encode()
    if _REAL_STDOUT is None:
        sys.stdout.write(data.decode())
        sys.stdout.flush()
    else:
        sys.stdout.flush()
        os.write(_REAL_STDOUT, data)


def main():
    _keep_stdout_for_the_line()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS) + ["c1"])
    ap.add_argument("--engine", default="auto", choices=["auto", "exact", "mfma", "bf16"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--segments", type=int, default=0)
    ap.add_argument("--slots", type=int, default=0, help="PN_OPT_FILTER_SLOTS (k' of the MFMA filter); 0 = auto")
    ap.add_argument("--structure", type=int, default=0, help="PN_OPT_MFMA_STRUCTURE; 0 = auto")
    ap.add_argument("--no-verify", action="store_true", help="skip the parity leg (outside the timed region)")
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"],
                    help="f64: the k-NN step on an f64 index (bf16 filter + f64 re-rank), one GPU")
    ap.add_argument("--metric", default="euclidean", choices=["euclidean", "cosine"],
                    help="cosine: BallTree::new(points, Cosine) -- bf16 filter over the normalised rows + Cosine::distance "
                         "re-rank (one GPU, index built through the host API)")
    ap.add_argument("--mode", default="knn", choices=["knn", "radius"],
                    help="radius: BallTree::query_radius over the batch (host queries in, host CSR out), one GPU")
    ap.add_argument("--radius", default="0.5",
                    help="radius of --mode radius (BASELINE configs[2]: 0.5); 'nn' = the median nearest-neighbour distance "
                         "of the queries x 1.0005, so that about half of the lists are non-empty")
    ap.add_argument("--shared-thresholds", type=int, default=-1,
                    help="PN_OPT_SHARED_THRESHOLDS: 0 off, 1 auto (library default), >= 2 the rank itself")
    ap.add_argument("--waves", type=int, default=0, choices=[0, 4, 8],
                    help="PN_OPT_BF16_WAVES: 0 = library default (by run length), 4 / 8 = force that main-pass kernel")
    ap.add_argument("--seed-model", type=int, default=-1, choices=[-1, 0, 1],
                    help="PN_OPT_SEED_MODEL: 0 = always scout, 1 = thresholds from the index's seed model where accepted "
                         "(library default)")
    ap.add_argument("--comm", default="abi", choices=["abi", "torch"],
                    help="abi: the all-gather is RCCL behind the C ABI (pn_sharded_*); torch: torch.distributed")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # Bare `python bench.py --gpus N`: this parent has made NO GPU call (importing torch and parsing flags does
        # not initialise HIP); it starts N rank processes as children -- one per GPU, the launch line of the
        # contract -- and leaves with their exit code.  (Never exec: a process that touched the GPU must not.)
        import socket
        import subprocess
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.run(cmd, stdout=_REAL_STDOUT).returncode)  # (the ranks' line goes to the real stdout)

    if args.config == "c1":
        return plumbing(args)
    if args.dtype == "f64" or args.metric == "cosine":
        if args.gpus != 1 or args.mode != "knn":
            sys.exit("bench.py: --dtype f64 / --metric cosine are the one-GPU k-NN step")
        return bench_f64(args)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                 f"(python bench.py --gpus N starts them itself)")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = world > 1 or os.environ.get("PN_BENCH_DIST") == "1"  # PN_BENCH_DIST=1: an nccl group even at world size 1
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{local_rank}"))
        rccl_world = dist.get_world_size()
    else:
        dist = None
        torch.cuda.set_device(0)
        rccl_world = 1
    dev = torch.device(f"cuda:{local_rank if use_dist else 0}")

    import petal_neighbors_amd as pn
    from petal_neighbors_amd import _lib
    from petal_neighbors_amd.sharded import AbiShardEngine, HipShardEngine, ShardedBallTree
    L = _lib.lib()

    n, dim, nq, k = CONFIGS[args.config]

    def gen(rows_lo, rows_hi, seed):
        t = torch.empty((rows_hi - rows_lo, dim), dtype=torch.float32, device=dev)
        rc = L.pn_fill_uniform_device_f32(t.data_ptr(), t.numel(), seed, rows_lo * dim, dev.index, None)
        assert rc == 0, _lib.last_error()
        return t

    queries = gen(0, nq, SEED_Q)
    torch.cuda.synchronize()
    # "abi": the whole step (local shard, ONE ncclAllGather, merge) is one asynchronous call into the C ABI;
    # "torch": the exchange is torch.distributed's all_gather_into_tensor (kept for comparison)
    comm_note = None
    index = None
    if args.comm == "abi":
        try:
            index = ShardedBallTree(n, lambda lo, hi: gen(lo, hi, SEED_P), engine=AbiShardEngine(dev.index))
        except Exception as e:  # PN_ERR_COMM and friends: the bench still has to produce its line
            comm_note = f"{type(e).__name__}: {e}"
        if dist and world > 1:  # every rank takes the same path
            ok = torch.tensor([0 if index is None else 1], device=dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0:
                index = None
                comm_note = comm_note or "another rank could not build its pn_sharded handle"
        if index is None:
            if world == 1:
                sys.exit(f"bench.py: {comm_note}")
            print(f"bench.py: rank {rank}: C-ABI exchange unavailable ({comm_note}); "
                  f"falling back to torch.distributed for the exchange", file=sys.stderr)
            args.comm = "torch"
    if index is None:
        index = ShardedBallTree(n, lambda lo, hi: gen(lo, hi, SEED_P), engine=HipShardEngine(dev.index))
    tree = index.engine.tree  # None for a rank without rows on the torch path
    if tree is not None:
        tree.set_engine(args.engine)
        if args.segments:
            tree.set_option(_lib.PN_OPT_SEGMENTS, args.segments)
        if args.slots:
            tree.set_option(_lib.PN_OPT_FILTER_SLOTS, args.slots)
        if args.structure:
            tree.set_option(_lib.PN_OPT_MFMA_STRUCTURE, args.structure)
        if os.environ.get("PN_BENCH_EXCHANGE") == "1" and args.comm == "abi":
            tree.set_option(_lib.PN_OPT_EXCHANGE_ALWAYS, 1)  # rehearsal of the N > 1 path on one GPU (tests)
        if args.shared_thresholds >= 0:
            tree.set_option(_lib.PN_OPT_SHARED_THRESHOLDS, args.shared_thresholds)
        if args.waves:
            tree.set_option(_lib.PN_OPT_BF16_WAVES, args.waves)
        if args.seed_model >= 0:
            tree.set_option(_lib.PN_OPT_SEED_MODEL, args.seed_model)
        tree.set_option(_lib.PN_OPT_PROFILE, 1)
    n_local = index.n_local
    out_idx = torch.empty((nq, min(k, n)), dtype=torch.int64, device=dev)
    out_dst = torch.empty((nq, min(k, n)), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()

    radius = None
    if args.mode == "radius":
        if world != 1 or tree is None:
            sys.exit("bench.py: --mode radius runs on one GPU")
        q_host = queries.cpu().numpy()
        if args.radius == "nn":
            _, d1 = tree.query_device(queries[:2048].contiguous(), 1)
            torch.cuda.synchronize()
            radius = float(np.float32(float(torch.median(d1[:, 0]).item()) * 1.0005))
        else:
            radius = float(np.float32(float(args.radius)))

    def step():
        if args.mode == "radius":  # the reference's call, batched: host queries in, host CSR out (sizes are data-dependent)
            return tree.query_radius_batch(q_host, radius)
        if args.comm == "abi":  # results land in preallocated buffers: nothing but the one C call per step
            return index.engine.index.query_device(queries, k, out_idx, out_dst)
        return index.query_batch(queries, k)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if tree is not None:
        tree.stats(reset=True)
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    st = tree.stats() if tree is not None else {"hot_launches": 0, "hot_ms": 0.0, "fallback_queries": 0, "candidates": 0,
                                                 "queries": 0, "evaluations": 0}

    radius_device = None
    if args.mode == "radius":
        # the same queries through pn_query_radius_device_* (queries and CSR in HBM, counts scanned on the device, nothing
        # read back inside the timed region), side by side with the host API's number above
        total = int(out[0][-1])
        cap = total + total // 8 + 1024
        d_off = torch.empty(nq + 1, dtype=torch.int64, device=dev)
        d_ids = torch.empty(cap, dtype=torch.int64, device=dev)
        d_tot = torch.empty(1, dtype=torch.int64, device=dev)
        for _ in range(max(args.warmup, 1)):
            tree.query_radius_device(queries, radius, cap, d_off, d_ids, d_tot)
        torch.cuda.synchronize()
        t0d = time.perf_counter()
        for _ in range(args.steps):
            tree.query_radius_device(queries, radius, cap, d_off, d_ids, d_tot)
        torch.cuda.synchronize()
        el_d = time.perf_counter() - t0d
        same = (int(d_tot.item()) == total and np.array_equal(d_off.cpu().numpy().astype(np.uint64), out[0]) and
                np.array_equal(d_ids[:total].cpu().numpy().astype(np.uint64), out[1]))
        radius_device = {"entry": "pn_query_radius_device_f32 (queries and CSR resident in HBM, no host round trip)",
                         "ms_per_step": round(el_d / args.steps * 1e3, 4), "value": round(nq * args.steps / el_d, 1),
                         "unit": "queries/s", "capacity": cap, "equals_host_api": bool(same)}
    verified = None
    if not args.no_verify and args.mode == "radius":
        verified = verify_radius(out, q_host, n, dim, nq, radius, tree)
        if radius_device is not None and not radius_device["equals_host_api"]:
            verified["ok"] = False
            verified["checks"].append("device API == host API: FAILED")
    elif not args.no_verify:
        verified = verify(out, queries, n, dim, nq, k, rank, world, index, gen)
        if dist:
            v = torch.tensor([1 if verified["ok"] else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(v, op=dist.ReduceOp.MIN)
            verified["ok"] = bool(v.item())

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        qps = nq * args.steps / elapsed
        # dominant kernel: its launches of one step together cover all nq queries against this rank's shard (the bf16
        # tier launches it twice per step: a short scout launch and the main launch); achieved = algorithmic flops of
        # a step / the kernel's time in a step = flops per launch / mean launch duration
        launches = max(int(st["hot_launches"]), 1)
        hot_ms = st["hot_ms"] / launches
        # 2*N*D flop per query; a step of more than 262 144 queries is served in several launches
        flops_per_launch = 2.0 * n_local * dim * nq * args.steps / launches
        achieved = flops_per_launch / (hot_ms * 1e-3) / 1e12 if hot_ms > 0 else 0.0
        if args.engine in ("auto", "bf16") and tree is not None and tree.bf16_eligible and n_local >= 4096 and dim >= 8:
            engine_used, peak = "bf16", PEAK_BF16_MFMA_TFLOPS
            kernel_name = "bf16_wide_kernel" if dim > 128 else "bf16_filter_kernel"
        elif args.engine != "exact" and tree is not None and tree.mfma_eligible:
            engine_used, peak = "mfma", PEAK_F32_MFMA_TFLOPS
            kernel_name = ("mfma_filter_wide_kernel" if dim > 128 else
                           "mfma_filter_v2_kernel")
        else:
            engine_used, kernel_name, peak = "exact", "exact_knn_kernel", PEAK_F32_MFMA_TFLOPS
        # HBM bytes per launch of the dominant kernel come from a SEPARATE rocprofv3 --pmc run of this same
        # command (counters cannot be read in-process); a committed summary is used when it describes
        # this kernel and config on one GPU, else null.
        traffic, traffic_src = (None, None) if world != 1 else \
            committed_traffic(kernel_name, args.config + ("_radius" if args.mode == "radius" else ""))
        traffic_ms = None
        if traffic_src:
            try:
                traffic_ms = json.load(open(os.path.join(ROOT, traffic_src))).get("kernel_ms_mean_under_pmc")
            except Exception:
                traffic_ms = None
        if args.mode == "radius":
            kernel_name = "bf16_wide_kernel" if dim > 128 else "bf16_filter_kernel"
        line = {
            "metric": (f"exact radius queries/sec ({n} x {dim} fp32, r={radius:.6g})" if args.mode == "radius" else
                       "exact k-NN queries/sec (1M x 128 fp32, k=10)" if args.config == "c2"
                       else f"exact k-NN queries/sec ({n} x {dim} fp32, k={k})"),
            "value": round(qps, 1), "unit": "queries/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": (f"{args.config} radius: {n} points x {dim} dims f32, {nq} queries, query_radius r={radius:.6g}, "
                                    f"uniform[0,1); host queries in, host CSR out (the API's boundary: list sizes are "
                                    f"data-dependent), so value includes the transfers" if args.mode == "radius" else
                                    f"{args.config}: {n} points x {dim} dims f32, {nq} queries, k={k}, uniform[0,1)"),
                       "n_points": n, "dim": dim, "n_queries": nq, "k": k, "engine": engine_used,
                       "sharding": f"corpus rows / {world}" if world > 1 else "none",
                       "exchange": ("none (one shard)" if world == 1 else
                                    "one ncclAllGather per step behind the C ABI (pn_sharded_query_device_f32)"
                                    if args.comm == "abi" else "torch.distributed.all_gather_into_tensor"),
                       "rccl_world_size": rccl_world,
                       **({"exchange_note": f"fell back from the C-ABI exchange: {comm_note}"} if comm_note else {})},
            "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": peak,
                         "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": traffic,
                         "traffic_unit": f"HBM bytes per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, {traffic_src})",
                         "kernel": kernel_name, "mfma_dtype": "bf16" if engine_used == "bf16" else "f32",
                         # live: ONE hipEvent bracket per step around the kernel's launches (scout-only + main for the
                         # bf16 tier); achieved = the step's algorithmic flops / that time
                         "kernel_ms_per_step": round(st["hot_ms"] / max(args.steps, 1), 4),
                         "flops_per_step": 2.0 * n_local * dim * nq,
                         "launches_per_step": round(launches / max(args.steps, 1), 2),
                         **(lambda ks: ({"kernel_ms_main": ks[0]["main_ms"], "kernel_ms_scout": ks[0]["scout_ms"],
                                         "kernel_split_source": f"{ks[1]} (rocprofv3 --kernel-trace --stats of this command; "
                                                                f"main + scout = {ks[0]['main_ms'] + ks[0]['scout_ms']:.4f} ms)"}
                                        | ({"kernel_split_note": ks[0]["note"]} if "note" in ks[0] else {})
                                        if ks else {}))(committed_kernel_split(args.config) if world == 1 and args.mode == "knn" else None),
                         **({"traffic_kernel_ms_under_pmc": traffic_ms} if traffic_ms else {}),
                         "whole_step_frac": round(2.0 * n * dim * nq / (ms_per_step * 1e-3) / 1e12
                                                  / (peak * world), 4)},
            "verified": bool(verified["ok"]) if verified else None,
            "verify": verified,
            **({"radius": radius, "results_per_query": round(float(out[0][-1]) / nq, 4),
                "host_api": {"ms_per_step": round(ms_per_step, 4), "value": round(qps, 1), "unit": "queries/s"},
                "device_api": radius_device} if args.mode == "radius" else {}),
            # N > 1 (or a forced exchange): the step's local half (this rank's filter + re-rank over its shard) and its
            # exchange half (ncclAllGather + merge), hipEvents on the call's stream inside the library, rank 0's
            **({"shard_ms": round(st["shard_ms"] / max(args.steps, 1), 4),
                "exchange_ms": round(st["exchange_ms"] / max(args.steps, 1), 4)} if st.get("exchange_ms", 0.0) > 0.0 else {}),
            "fallback_queries": int(st["fallback_queries"]),
            "candidates_per_query": round(st["candidates"] / max(st["queries"], 1), 2),
            "exact_evaluations_per_query": round(st["evaluations"] / max(st["queries"], 1), 2),
        }
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(n, dim, k, gpu_out=out if args.mode == "knn" else None)
            line["cpu_baseline"] = cb
            if cb.get("agrees_with_gpu") is False:  # the tree walk and the GPU disagree outside exact ties: not verified
                line["verified"] = False
        emit(line)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
