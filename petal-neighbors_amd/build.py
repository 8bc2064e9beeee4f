"""Build recipe for libpetal_mi355x.so (gfx950 only; hipcc cross-compiles without a GPU).

    python petal-neighbors_amd/build.py [--force] [--asm]

Every translation unit is compiled in-tree to an object, then linked into
``petal-neighbors_amd/libpetal_mi355x.so`` (git-ignored; it travels to the GPU
box with the gpurun snapshot).  The exact-arithmetic units are compiled with
``-ffp-contract=off`` so the reference's unfused fold (src/distance.rs:26-35)
can never be contracted into an FMA; ``--asm`` also keeps the gfx950 ISA of
those units under ``build/`` for the no-FMA audit in tests/test_build.py.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ARCH = "gfx950"
# diagnostic (timing-only / instrumented) builds: PN_DIAG_FLAGS="-DPN_DIAG_BF_COUNT" python build.py
# They are a DIFFERENT library in a different object directory (libpetal_mi355x_diag.so, build_diag/): some of those
# flags give wrong results by design, and nothing a diagnostic build does may touch the product library.  Load one with
# PN_LIBRARY_PATH=<path> (petal-neighbors_amd/_lib.py).
DIAG = os.environ.get("PN_DIAG_FLAGS", "").split()
_TAG = os.environ.get("PN_DIAG_TAG", "")  # several diagnostic variants side by side: ..._diag_<tag>.so
_SFX = ("_diag" + ("_" + _TAG if _TAG else "")) if DIAG else ""
BUILD_PRODUCT = os.path.join(HERE, "build")
BUILD = os.path.join(HERE, "build" + _SFX)
LIB = os.path.join(HERE, "libpetal_mi355x" + _SFX + ".so")

# (source, extra flags)
UNITS = [
    ("index.hip", []),
    ("exact_scan.hip", ["-ffp-contract=off"]),
    ("select.hip", ["-ffp-contract=off"]),
    ("pack.hip", []),
    ("mfma_filter_v2.hip", []),
    ("bf16_filter.hip", []),
    ("sharded.hip", []),
    ("radius_device.hip", []),
    ("metric.cpp", ["-ffp-contract=off"]),
    ("tree.cpp", ["-ffp-contract=off"]),
]
COMMON = ["-O3", "-fPIC", "-std=c++17", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function",
          "-fno-fast-math", "-fhip-fp32-correctly-rounded-divide-sqrt"]
HEADERS = [os.path.join(CSRC, "pn_internal.h"), os.path.join(CSRC, "topk_buffer.h"), os.path.join(CSRC, "host_tree.h"),
           os.path.join(os.path.dirname(HERE), "include", "petal_mi355x.h")]


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: the MI355X library cannot be built (no CPU fallback exists)")


def _stale(target: str, deps: list[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, keep_asm: bool = False, verbose: bool = False) -> str:
    os.makedirs(BUILD, exist_ok=True)
    cc = hipcc()
    # the flag set the objects were built with is part of their identity: a diagnostic (wrong-result) build must
    # never survive into a later plain build() -- a changed flag set rebuilds everything
    stamp = os.path.join(BUILD, ".flags")
    flags_now = " ".join(COMMON + DIAG)
    try:
        flags_then = open(stamp).read()
    except OSError:
        flags_then = None
    if flags_then != flags_now:
        force = True
    objs = []
    jobs = []  # compile commands, run side by side below (the units are independent; bf16_filter.hip alone takes minutes)
    me = os.path.abspath(__file__)
    for src, extra in UNITS:
        sp = os.path.join(CSRC, src)
        obj = os.path.join(BUILD, os.path.splitext(src)[0] + ".o")
        if DIAG and "PN_DIAG" not in open(sp).read():
            # a unit that never looks at a diagnostic flag: the product object is the same object
            prod = os.path.join(BUILD_PRODUCT, os.path.splitext(src)[0] + ".o")
            if os.path.exists(prod) and not _stale(prod, [sp, me] + HEADERS):
                objs.append(prod)
                continue
        objs.append(obj)
        if force or _stale(obj, [sp, me] + HEADERS):
            lang = ["-x", "hip"] if src.endswith(".hip") else []
            cmd = [cc] + COMMON + extra + DIAG + lang + ["-c", sp, "-o", obj]
            jobs.append(cmd)
        asm = os.path.join(BUILD, os.path.splitext(src)[0] + ".s")
        want_asm = keep_asm is True or (keep_asm and os.path.splitext(src)[0] in keep_asm)  # True: every unit
        if want_asm and src.endswith(".hip") and (force or _stale(asm, [sp, me] + HEADERS)):
            cmd = [cc] + COMMON + extra + DIAG + ["-x", "hip", "--cuda-device-only", "-S", sp, "-o", asm]
            jobs.append(cmd)
    if jobs:
        from concurrent.futures import ThreadPoolExecutor

        def run(cmd):
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
        with ThreadPoolExecutor(max_workers=min(len(jobs), int(os.environ.get("PN_BUILD_JOBS", "6")))) as ex:
            list(ex.map(run, jobs))  # (re-raises the first failure)
    if force or _stale(LIB, objs):
        cmd = [cc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs + ["-ldl"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    with open(stamp, "w") as f:
        f.write(flags_now)
    return LIB


def build_flags() -> str:
    """Diagnostic flags of the library as built (empty for the product build)."""
    try:
        txt = open(os.path.join(BUILD, ".flags")).read()
    except OSError:
        return "?"
    return " ".join(x for x in txt.split() if x.startswith("-DPN_DIAG"))


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, keep_asm="--asm" in sys.argv, verbose=True))
