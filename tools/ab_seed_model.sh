#!/bin/bash
# usage (GPU box): tools/ab_seed_model.sh "<configs>" [reps] -- interleaved A/B on ONE device: starting thresholds from
# the scout launch (--seed-model 0) against the index's seed model (--seed-model 1), per configuration
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
CFGS=${1:-c2}; R=${2:-2}
for c in $CFGS; do
  for rep in $(seq 1 $R); do
    for v in 0 1; do
      EXTRA=${PN_AB_EXTRA:-}
      PN_DEBUG_PLAN=$([ $rep = 1 ] && echo 1 || echo 0) timeout -k 10 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --config $c --seed-model $v $EXTRA > gpurun_out/absm_$v.json 2> gpurun_out/absm_$v.err || { echo "$c $v failed"; tail -3 gpurun_out/absm_$v.err; continue; }
      [ $rep = 1 ] && grep -h "seed model\|model_seed" gpurun_out/absm_$v.err | head -4
      python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/absm_$v.json') if l.startswith('{')][-1]); r=d['roofline']
print('%-6s model %s rep $rep kernel ms/step %.4f  step %.4f  frac %.4f  cand/q %.1f eval/q %.1f fb %d verified %s' % ('$c', '$v', r['kernel_ms_per_step'], d['ms_per_step'], r['frac'], d['candidates_per_query'], d['exact_evaluations_per_query'], d['fallback_queries'], d['verified']))
"
    done
  done
done
