"""Static audit of the exact kernels' gfx950 ISA (CPU only): the reference's
fold is sub, mul, add with separate roundings (src/distance.rs:30-31), so the
coordinate loop of every exact kernel must contain no fused multiply-add."""
import os
import re

import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def asm():
    import importlib.util as u
    spec = u.spec_from_file_location("pn_build", os.path.join(ROOT, "petal-neighbors_amd", "build.py"))
    b = u.module_from_spec(spec)
    spec.loader.exec_module(b)
    units = ("exact_scan", "select")
    b.build(keep_asm=units)  # ISA text of these units only
    out = {}
    for name in units:
        out[name] = open(os.path.join(ROOT, "petal-neighbors_amd", "build", name + ".s")).read()
    return out


def _blocks(text):
    cur, name = [], "entry"
    for line in text.splitlines():
        m = re.match(r"^(\.LBB\d+_\d+|_Z\w+):", line)
        if m:
            yield name, cur
            cur, name = [], m.group(1)
        else:
            cur.append(line.strip())
    yield name, cur


EXACT_UNITS = ("exact_scan", "select")  # the units that evaluate the reference's fold (the filters only bound it)


def test_no_fma_in_exact_coordinate_loops(asm):
    fold_blocks = 0
    for unit in EXACT_UNITS:
        text = asm[unit]
        for name, lines in _blocks(text):
            ins = [l.split()[0] for l in lines if l and not l.startswith((";", "."))]
            is_fold = any(i.startswith(("v_pk_mul_f32", "v_mul_f32", "v_mul_f64")) for i in ins) and \
                any(i.startswith(("ds_read_b128", "ds_read2_b64", "global_load_dwordx4", "global_load_dword")) for i in ins) and \
                not any(i.startswith(("v_sqrt", "v_rsq")) for i in ins)
            if not is_fold:
                continue
            fold_blocks += 1
            # the correctly rounded division (Cosine: dot / (|a| |b|)) expands to v_div_scale .. v_div_fixup with fmas
            # of its own: that sequence IS the IEEE quotient, not a contraction of the fold
            kept, in_div = [], False
            for i in ins:
                if i.startswith("v_div_scale"):
                    in_div = True
                if not in_div:
                    kept.append(i)
                if i.startswith("v_div_fixup"):
                    in_div = False
            bad = [i for i in kept if re.match(r"v_(pk_)?fma|v_fmac|v_mad_f|v_mac_f", i)]
            assert not bad, f"{unit}:{name} contracts the fold: {bad[:4]}"
    assert fold_blocks >= 6  # f32+f64 x (knn, radius, pairwise) at least


def test_exact_kernels_do_not_spill(asm):
    for unit in EXACT_UNITS:
        text = asm[unit]
        for m in re.finditer(r"\.vgpr_spill_count:\s+(\d+)", text):
            assert int(m.group(1)) == 0, unit


def test_bf16_filter_kernels_neither_spill_nor_lose_their_occupancy(asm, tmp_path):
    """Every instantiation of the first-tier kernels stays within 256 VGPRs without scratch: two waves per SIMD (two
    workgroups per CU for the narrow kernel, one 8-wave workgroup for the wide one) is what their LDS and register
    budgets are planned for.  (Round 2: tags written in the matrix pipe's shadow cost ~50 registers, spilled for
    KS >= 5 and made the full-size k = 100 configuration 10 % slower before anyone looked.)  Read from the code
    object inside the product's own object file -- no second compilation."""
    import shutil
    import subprocess
    llvm = "/opt/rocm/lib/llvm/bin"
    obj = os.path.join(ROOT, "petal-neighbors_amd", "build", "bf16_filter.o")
    work = tmp_path / "bf16_filter.o"
    shutil.copy(obj, work)
    subprocess.run([os.path.join(llvm, "llvm-objdump"), "--offloading", str(work)], check=True, capture_output=True,
                   cwd=tmp_path)
    co = [f for f in os.listdir(tmp_path) if "gfx950" in f]
    assert len(co) == 1, co
    text = subprocess.run([os.path.join(llvm, "llvm-readelf"), "--notes", str(tmp_path / co[0])], check=True,
                          capture_output=True, text=True).stdout
    kernels = 0
    for blk in text.split("  - .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        if "bf16_filter_kernel" not in name and "bf16_wide_kernel" not in name:
            continue
        kernels += 1
        g = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1))
        assert g("vgpr_spill_count") == 0 and g("private_segment_fixed_size") == 0, name
        assert g("vgpr_count") <= 256, name
    assert kernels >= 100


def test_product_library_carries_no_diagnostic_flags():
    """timing-only PN_DIAG_* builds give wrong results; the flag set is stamped next to the objects and a
    plain build() rebuilds when it differs -- the library the tests and bench.py load is the product build."""
    import importlib.util as u
    spec = u.spec_from_file_location("pn_build", os.path.join(ROOT, "petal-neighbors_amd", "build.py"))
    b = u.module_from_spec(spec)
    spec.loader.exec_module(b)
    if os.environ.get("PN_DIAG_FLAGS"):
        pytest.skip("a diagnostic build was asked for explicitly")
    b.build()
    assert b.build_flags() == ""


def test_oracle_under_address_and_ub_sanitizers(tmp_path):
    """The oracle's C restatement under -fsanitize=address,undefined (oracle/Makefile `asan`), in a child process
    (the sanitizer runtime must be the first library loaded): golden vector G1, the faithful tree against the
    brute force, radius, pairwise and the degenerate 8-identical-points build."""
    import subprocess
    import sys
    mk = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "asan"], capture_output=True, text=True)
    assert mk.returncode == 0, mk.stderr
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("libasan not found")
    script = tmp_path / "asan_child.py"
    script.write_text(f"""
import sys
sys.path.insert(0, {ROOT!r})
import numpy as np
import oracle
oracle._LIB_PATH = {os.path.join(ROOT, 'oracle', 'liboracle_asan.so')!r}
pts = np.array([[1., 1.], [1., 2.], [9., 9.]])
t = oracle.Tree(pts)
i, d = t.query(np.array([3., 3.]), 2)
assert list(i) == [1, 0] and abs(d[0] - 5 ** 0.5) < 1e-15
rng = np.random.default_rng(3)
for dt in (np.float32, np.float64):
    p = rng.random((300, 5)).astype(dt); q = rng.random((20, 5)).astype(dt)
    t = oracle.Tree(p)
    ti, td = t.query_batch(q, 7, nthreads=2)
    bi, bd = oracle.brute_knn(p, q, 7)
    assert td.tobytes() == bd.tobytes()
    for a in range(20):
        assert sorted(t.query_radius(q[a], dt(0.4)).tolist()) == oracle.brute_radius(p, q[a], dt(0.4)).tolist()
        t.query_nearest(q[a])
    oracle.pairwise(p[:40]); oracle.pairwise_cosine(p[:40])
same = oracle.Tree(np.ones((8, 2)))
assert same.query(np.array([1., 2.]), 3)[1].tolist() == [1.0, 1.0, 1.0]
print("ASAN_OK")
""")
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0")
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ASAN_OK" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
