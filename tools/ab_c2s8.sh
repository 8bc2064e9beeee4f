#!/bin/bash
# one GPU's share of the 8-GPU headline: variants interleaved on one device
cd $GRAFT_REPO_ROOT
run() { # tag, env assignments..., then -- bench args
  local tag=$1; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout -k 10 300 python bench.py --no-verify --no-cpu-baseline --steps 30 --warmup 5 "$@" > gpurun_out/c2s8_$tag.json 2> gpurun_out/c2s8_$tag.err || { echo "$tag failed"; tail -3 gpurun_out/c2s8_$tag.err; return; }
  python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/c2s8_$tag.json') if l.startswith('{')][-1]); r=d['roofline']
print('%-14s kernel ms/step %.4f  step %.4f  frac %.4f  cand/q %.1f fb %d' % ('$tag', r['kernel_ms_per_step'], d['ms_per_step'], r['frac'], d['candidates_per_query'], d['fallback_queries']))
"
}
for cfg in c2s8 c2s4; do
for rep in 1 2; do
  echo "== $cfg rep $rep"
  run base X=1 -- --config $cfg
  run noss PN_LIBRARY_PATH=$GRAFT_REPO_ROOT/petal-neighbors_amd/libpetal_mi355x_diag_noss.so -- --config $cfg
  run sh100 PN_EXP_SH_MIN_RUN=100 -- --config $cfg
  run lam06 PN_EXP_SCOUT_LAMBDA=0.6 -- --config $cfg
  run lam24 PN_EXP_SCOUT_LAMBDA=2.4 -- --config $cfg
done
done
