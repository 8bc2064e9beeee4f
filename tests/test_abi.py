"""CPU-side checks of the C-ABI boundary (no GPU compute calls).

* libpetal_mi355x.so builds for gfx950, loads, and exports every symbol that
  include/petal_mi355x.h declares;
* the reference's construction errors (src/ball_tree.rs:44-49, src/lib.rs:9-16)
  are raised by the library itself, before any device is touched;
* without a GPU the product fails loudly (PN_ERR_DEVICE) -- no CPU fallback;
* the scalar Metric functions (host-side by design) are bit-identical to the oracle.
"""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT, uniform


def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "petal_mi355x.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(pn_[a-z0-9_]+)\s*\(", txt)))


def test_library_builds_and_exports_every_declared_symbol(pn):
    from petal_neighbors_amd import _lib
    declared = _header_symbols()
    assert len(declared) >= 30
    assert sorted(_lib.SIGNATURES) == declared, "ctypes table and header disagree"
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True)
    exported = set(re.findall(r" T (pn_[a-z0-9_]+)", out.stdout))
    missing = [s for s in declared if s not in exported]
    assert not missing, f"not exported: {missing}"
    assert _lib.lib().pn_abi_version() == 3


def test_gfx950_code_object_is_embedded(pn):
    from petal_neighbors_amd import _lib
    data = open(_lib.LIB_PATH, "rb").read()
    assert b"gfx950" in data
    for kern in (b"exact_knn_kernel", b"select_exact_kernel", b"row_norms_kernel", b"mfma_filter"):
        assert kern in data, kern


def test_array_errors_come_from_the_library(pn):
    """ball_tree_empty / ball_tree_column_base (src/ball_tree.rs:623-638)."""
    with pytest.raises(pn.ArrayError.Empty) as e:
        pn.BallTree.euclidean(np.zeros((0, 0)))
    assert str(e.value) == "array is empty"
    with pytest.raises(pn.ArrayError.Empty):
        pn.BallTree.euclidean(np.zeros((0, 3), dtype=np.float32))
    arr = np.array([[1.0, 1.0], [1.0, 1.1], [9.0, 9.0]])
    with pytest.raises(pn.ArrayError.NotContiguous) as e:
        pn.BallTree.euclidean(np.asfortranarray(arr))
    assert str(e.value) == "array is not contiguous in memory"
    with pytest.raises(pn.ArrayError.NotContiguous):
        pn.BallTree.euclidean(arr[:, ::-1])
    assert issubclass(pn.ArrayError.Empty, pn.ArrayError)
    # raw ABI: same codes, same strings (src/lib.rs:12-15)
    from petal_neighbors_amd import _lib
    L = _lib.lib()
    h = C.c_void_p(0)
    assert L.pn_index_create_f32(None, 0, 3, 3, 1, 0, C.byref(h)) == _lib.PN_ERR_EMPTY
    assert L.pn_strerror(_lib.PN_ERR_EMPTY) == b"array is empty"
    assert L.pn_strerror(_lib.PN_ERR_NOT_CONTIGUOUS) == b"array is not contiguous in memory"
    assert L.pn_last_error() == b"array is empty"


def test_metric_equality_like_ball_tree_metric(pn):
    """ball_tree_metric (src/ball_tree.rs:640-647): new(.., Euclidean) == euclidean(..) metric."""
    assert pn.distance.Euclidean() == pn.distance.Euclidean()


def test_no_cpu_fallback_without_gpu(pn):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(pn.PetalError) as e:
        pn.BallTree.euclidean(np.ones((4, 2), dtype=np.float32))
    assert e.value.code == 4 and "no CPU path" in str(e.value)
    with pytest.raises(pn.PetalError):
        pn.distance.pairwise(np.ones((4, 2)))


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_scalar_metric_bit_exact_vs_oracle(pn, oracle_mod, dtype):
    m = pn.distance.Euclidean()
    for d in (0, 1, 2, 3, 7, 10, 96, 128, 131, 768):
        a = uniform((d,), 11 + d, dtype) * dtype(3) - dtype(1)
        b = uniform((d,), 97 + d, dtype)
        assert m.distance(a, b).tobytes() == oracle_mod.euclidean(a, b).tobytes()
        assert m.rdistance(a, b).tobytes() == oracle_mod.reuclidean(a, b).tobytes()
    # zip truncation (src/distance.rs:27-28)
    assert m.distance(np.array([3.0, 4.0, 100.0]), np.array([0.0, 0.0])) == 5.0
    x = dtype(2.25)
    assert m.rdistance_to_distance(x) == oracle_mod.rdistance_to_distance(x, dtype) == dtype(1.5)
    assert m.distance_to_rdistance(dtype(1.5)) == oracle_mod.distance_to_rdistance(1.5, dtype) == x
    nan = m.distance(np.array([np.nan], dtype=dtype), np.array([0.0], dtype=dtype))
    assert np.isnan(nan)
