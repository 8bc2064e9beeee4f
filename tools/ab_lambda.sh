#!/bin/bash
# usage (GPU box): tools/ab_lambda.sh <config> "<lambdas>"  -- the scout sample's size (PN_EXP_SCOUT_LAMBDA: expected number of
# the R relevant rows inside the shared scout's sample) against step and kernel time, one device
cd $GRAFT_REPO_ROOT
C=${1:-c3s}
for lam in ${2:-1.2 3 6 12}; do
  PN_EXP_SCOUT_LAMBDA=$lam python bench.py --config $C --no-cpu-baseline --steps ${3:-20} --warmup 3 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print('$C lambda $lam: step %.3f kernel %.3f cand/q %.0f eval/q %.0f fb %d verified %s' % (d['ms_per_step'], r['kernel_ms_per_step'], d['candidates_per_query'], d['exact_evaluations_per_query'], d['fallback_queries'], d['verified']))"
done
