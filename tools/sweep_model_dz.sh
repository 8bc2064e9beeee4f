#!/bin/bash
# usage (GPU box): tools/sweep_model_dz.sh "<configs>" "<dz values>" -- the seed model's z moved by dz (PN_EXP_MODEL_DZ):
# how sensitive the main launch is to where the starting thresholds sit, and where unproven queries begin
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for c in ${1:-c2}; do
  for dz in ${2:-0}; do
    PN_EXP_MODEL_DZ=$dz timeout -k 10 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --config $c $PN_AB_EXTRA > gpurun_out/dz.json 2> gpurun_out/dz.err || { echo "$c $dz failed"; tail -3 gpurun_out/dz.err; continue; }
    python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/dz.json') if l.startswith('{')][-1]); r=d['roofline']
print('%-6s dz %5s kernel ms/step %.4f  step %.4f  frac %.4f  cand/q %.1f eval/q %.1f fb %d verified %s' % ('$c', '$dz', r['kernel_ms_per_step'], d['ms_per_step'], r['frac'], d['candidates_per_query'], d['exact_evaluations_per_query'], d['fallback_queries'], d['verified']))
"
  done
done
