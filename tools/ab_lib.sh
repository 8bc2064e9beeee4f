#!/bin/bash
# usage (GPU box): tools/ab_lib.sh <diag tag> "<configs>" [reps]  -- interleaved A/B of the product library (A) and
# petal-neighbors_amd/libpetal_mi355x_diag_<tag>.so (B) on ONE device, per bench configuration
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
TAG=$1; CFGS=${2:-c2}; R=${3:-3}
for c in $CFGS; do
  for rep in $(seq 1 $R); do
    for v in A B; do
      if [ $v = A ]; then unset PN_LIBRARY_PATH; else export PN_LIBRARY_PATH=$GRAFT_REPO_ROOT/petal-neighbors_amd/libpetal_mi355x_diag_$TAG.so; fi
      timeout -k 10 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --config $c > gpurun_out/abl_$v.json 2> gpurun_out/abl_$v.err || { echo "$v failed"; tail -3 gpurun_out/abl_$v.err; continue; }
      python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/abl_$v.json') if l.startswith('{')][-1]); r=d['roofline']
print('%-6s %s rep $rep kernel ms/step %.4f  step %.4f  frac %.4f  cand/q %.1f eval/q %.1f fb %d verified %s' % ('$c', '$v', r['kernel_ms_per_step'], d['ms_per_step'], r['frac'], d['candidates_per_query'], d['exact_evaluations_per_query'], d['fallback_queries'], d['verified']))
"
    done
  done
done
