#!/bin/bash
# usage (GPU box): tools/ab_model_all_plans.sh "<configs>" [steps] -- plans without a shared scout (one workgroup per query
# tile, grids in rounds): the run's own scout pass (PN_EXP_MODEL_ALL_PLANS=0) against thresholds from the seed model
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for c in ${1:-c5m}; do
  for v in 0 1 0 1; do
    PN_DEBUG_PLAN=1 PN_EXP_MODEL_ALL_PLANS=$v timeout -k 10 600 python bench.py --no-cpu-baseline --steps ${2:-5} --warmup 2 --config $c > gpurun_out/abmp.json 2> gpurun_out/abmp.err || { echo "$c $v failed"; tail -3 gpurun_out/abmp.err; continue; }
    python3 -c "
import json,re
d=json.loads([l for l in open('gpurun_out/abmp.json') if l.startswith('{')][-1]); r=d['roofline']
m=re.findall(r'shared_scout (\d) .*model_seed (\d)', open('gpurun_out/abmp.err').read())
print('%-8s all-plans %s (shared_scout %s model %s) kernel ms/step %.4f  step %.4f  frac %.4f  cand/q %.1f fb %d verified %s' % ('$c', '$v', m[-1][0] if m else '?', m[-1][1] if m else '?', r['kernel_ms_per_step'], d['ms_per_step'], r['frac'], d['candidates_per_query'], d['fallback_queries'], d['verified']))
"
  done
done
