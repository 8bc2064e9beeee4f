"""The product's host-only code under AddressSanitizer + UndefinedBehaviorSanitizer (sanitizers covered
oracle/ only).  csrc/tree.cpp (the reference's ball tree behind the introspection API, src/ball_tree.rs:296-353, 445-613) and
csrc/metric.cpp (scalar Metric<A>, src/distance.rs:21-122) need no GPU: they are compiled stand-alone with g++
-fsanitize=address,undefined together with tests/cpp/host_sanitize.cpp, run on seeded corpora, and every node they produce
is compared with the oracle's faithful tree -- the node-by-node check of tests/test_gpu_tree_accessors.py, on the CPU."""
import os
import struct
import subprocess

import numpy as np
import pytest

from conftest import ROOT, uniform

CSRC = os.path.join(ROOT, "petal-neighbors_amd", "csrc")


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("san") / "host_sanitize")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", "-ffp-contract=off", "-fsanitize=address,undefined",
           "-fno-sanitize-recover=undefined", "-Wall", os.path.join(ROOT, "tests", "cpp", "host_sanitize.cpp"),
           os.path.join(CSRC, "tree.cpp"), os.path.join(CSRC, "metric.cpp"), "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0 and "asan" in (r.stderr or "").lower() and "cannot find" in r.stderr:
        pytest.skip("libasan not installed")
    assert r.returncode == 0, r.stderr
    return out


@pytest.mark.parametrize("metric", ["euclidean", "cosine"])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("n,dim", [(1, 3), (2, 2), (3, 1), (7, 4), (8, 4), (100, 3), (1000, 10), (4097, 17)])
def test_tree_and_metric_under_asan_ubsan_match_the_oracle(exe, oracle_mod, tmp_path, metric, dtype, n, dim):
    pts = uniform((n, dim), 500 + n + dim, dtype) - (dtype(0.25) if metric == "cosine" else dtype(0))
    if n >= 8:
        pts[5] = pts[1]  # equal coordinates: the quick-select's tie handling
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    with open(fin, "wb") as f:
        f.write(struct.pack("<IIQQ", pts.itemsize, 1 if metric == "cosine" else 0, n, dim))
        f.write(np.ascontiguousarray(pts).tobytes())
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([exe, str(fin), str(fout)], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, (r.returncode, r.stderr[-3000:])
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-3000:]
    raw = open(fout, "rb").read()
    ref = oracle_mod.Tree(pts, metric=metric)
    nn = struct.unpack_from("<Q", raw, 0)[0]
    assert nn == ref.num_nodes == (1 << n.bit_length()) - 1
    idx = np.frombuffer(raw, dtype=np.uint64, count=n, offset=8)
    assert np.array_equal(idx, ref.idx.astype(np.uint64))
    off = 8 + 8 * n
    rec = 32 + dim * pts.itemsize
    for i in range(nn):
        s, e, leaf = struct.unpack_from("<QQQ", raw, off)
        radius = struct.unpack_from("<d", raw, off + 24)[0]
        cen = np.frombuffer(raw, dtype=dtype, count=dim, offset=off + 32)
        node = ref.node(i)
        assert (s, e) == tuple(node["range"]) and bool(leaf) == node["is_leaf"], i
        assert dtype(radius).tobytes() == dtype(node["radius"]).tobytes(), i
        assert cen.tobytes() == node["centroid"].tobytes(), i
        off += rec
    eu, reu, co = struct.unpack_from("<ddd", raw, off)
    a, b = pts[0], pts[n - 1]
    assert dtype(eu).tobytes() == dtype(oracle_mod.euclidean(a, b)).tobytes()
    assert dtype(reu).tobytes() == dtype(oracle_mod.reuclidean(a, b)).tobytes()
    want_c = oracle_mod.cosine(a, b)
    assert (np.isnan(co) and np.isnan(want_c)) or dtype(co).tobytes() == dtype(want_c).tobytes()
