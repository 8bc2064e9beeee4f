"""The C++ mirror (include/petal_neighbors.hpp) compiled against the C ABI and run as a plain
C++ program: the reference's own unit tests restated in C++ (tests/cpp/host_mirror.cpp)."""
import os
import subprocess

import pytest

from conftest import ROOT


def _build(pn):
    from petal_neighbors_amd import _lib
    exe = os.path.join(ROOT, "tests", "cpp", "host_mirror")
    src = exe + ".cpp"
    hdr = os.path.join(ROOT, "include", "petal_neighbors.hpp")
    if (not os.path.exists(exe)) or max(os.path.getmtime(src), os.path.getmtime(hdr),
                                        os.path.getmtime(_lib.LIB_PATH)) > os.path.getmtime(exe):
        libdir = os.path.dirname(_lib.LIB_PATH)
        subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), src, "-o", exe,
                        "-L", libdir, "-lpetal_mi355x", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"],
                       check=True)
    return exe


def test_cpp_mirror_compiles_and_validates_on_cpu(pn):
    exe = _build(pn)
    r = subprocess.run([exe, "cpu"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.gpu
def test_cpp_mirror_reference_tests_on_gpu(pn):
    exe = _build(pn)
    r = subprocess.run([exe, "gpu"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
