#!/bin/bash
# usage (GPU box): tools/ab_libs.sh "<bench args>" tag1 tag2 ...   -- timing-only comparison of diagnostic libraries
# (built here with PN_DIAG_TAG=<tag> PN_DIAG_FLAGS=... python petal-neighbors_amd/build.py); "product" = the product library.
# Two interleaved rounds on one device (timings of different devices must not be compared).
cd $GRAFT_REPO_ROOT
BA="$1"; shift
for rep in 1 2; do
  for t in "$@"; do
    if [ "$t" = product ]; then L=""; else L=$GRAFT_REPO_ROOT/petal-neighbors_amd/libpetal_mi355x_diag_$t.so; fi
    PN_LIBRARY_PATH=$L timeout -k 10 300 python bench.py --no-verify --no-cpu-baseline --steps 20 --warmup 5 $BA > gpurun_out/ab_$t.json 2> gpurun_out/ab_$t.err || { echo $t failed; tail -3 gpurun_out/ab_$t.err; continue; }
    python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/ab_$t.json') if l.startswith('{')][-1]); r=d['roofline']
print('%-16s rep $rep kernel ms/step %.4f  step %.4f  frac %.4f' % ('$t', r['kernel_ms_per_step'], d['ms_per_step'], r['frac']))
"
  done
done
