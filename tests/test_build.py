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
    b.build(keep_asm=True)
    out = {}
    for name in ("exact_scan", "select"):
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


def test_no_fma_in_exact_coordinate_loops(asm):
    fold_blocks = 0
    for unit, text in asm.items():
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
    for unit, text in asm.items():
        for m in re.finditer(r"\.vgpr_spill_count:\s+(\d+)", text):
            assert int(m.group(1)) == 0, unit
