#!/bin/bash
# usage (GPU box): tools/sweep_model_rank.sh "<configs>" "<ranks>" -- the corpus rank the seed model aims its thresholds
# at (PN_EXP_MODEL_RANK; 0 = the library's rule): main launch time and unproven queries per setting
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for c in ${1:-c2}; do
  for r in ${2:-0}; do
    PN_DEBUG_PLAN=1 PN_EXP_MODEL_RANK=$r timeout -k 10 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --config $c $PN_AB_EXTRA > gpurun_out/mr.json 2> gpurun_out/mr.err || { echo "$c $r failed"; tail -3 gpurun_out/mr.err; continue; }
    python3 -c "
import json,re
d=json.loads([l for l in open('gpurun_out/mr.json') if l.startswith('{')][-1]); r=d['roofline']
z=[m.group(1) for m in re.finditer(r'model_seed 1 z ([0-9.]+)', open('gpurun_out/mr.err').read())]
print('%-6s rank %5s z %s kernel ms/step %.4f  step %.4f  frac %.4f  cand/q %.1f fb %d verified %s' % ('$c', '$r', z[0] if z else '-', r['kernel_ms_per_step'], d['ms_per_step'], r['frac'], d['candidates_per_query'], d['fallback_queries'], d['verified']))
"
  done
  grep -h "seed model" gpurun_out/mr.err | head -1
done
