#!/usr/bin/env python3
"""Register counts, spills and scratch of the kernels in one compilation unit (compiled with the product flags +
$PN_DIAG_FLAGS; tests/test_build.py reads the same numbers from the product object without compiling).
usage: tools/kernel_regs.py [unit.hip] [name substring ...]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
unit = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "petal-neighbors_amd", "csrc", "bf16_filter.hip")
subs = sys.argv[2:] or ["bf16_filter_kernel", "bf16_wide_kernel"]
with tempfile.TemporaryDirectory() as d:
    out = os.path.join(d, "u.s")
    extra = os.environ.get("PN_DIAG_FLAGS", "").split()
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-fast-math",
                    "-fhip-fp32-correctly-rounded-divide-sqrt", "-x", "hip", "--cuda-device-only", "-S", unit, "-o", out]
                   + extra, check=True, stderr=subprocess.DEVNULL, cwd=d)
    t = open(out).read()
for b in t.split("  - .agpr_count:")[1:]:
    name = re.search(r"\.name:\s+(\S+)", b).group(1)
    if not any(s in name for s in subs):
        continue
    # template arguments from the mangled name: ILi<number>E / ILb<0|1>E
    targs = re.findall(r"IL?[ib](\d+)E|L[ib](\d+)E", name)
    targs = [a or b_ for a, b_ in targs]
    g = lambda k: re.search(r"\.%s:\s+(\d+)" % k, b).group(1)
    print("%-28s <%s>  vgpr %s  sgpr %s  spill %s  scratch %s B  lds %s" % (
        [s for s in subs if s in name][0], ",".join(targs), g("vgpr_count"), g("sgpr_count"), g("vgpr_spill_count"),
        g("private_segment_fixed_size"), g("group_segment_fixed_size")))
