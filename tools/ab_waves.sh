#!/bin/bash
# usage (GPU box): tools/ab_waves.sh "<configs>" [reps] [diag tags...]  -- interleaved A/B on ONE device of the 4-wave
# kernel (--waves 4), the 8-wave kernel (library default) and diagnostic libraries of the 8-wave kernel, per configuration
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
CFGS=${1:-c2}; R=${2:-2}; shift; shift
for c in $CFGS; do
  for rep in $(seq 1 $R); do
    for v in w4 w8 "$@"; do
      unset PN_LIBRARY_PATH; W=0
      case $v in
        w4) W=4;;
        w8) W=8;;
        *) export PN_LIBRARY_PATH=$GRAFT_REPO_ROOT/petal-neighbors_amd/libpetal_mi355x_diag_$v.so;;
      esac
      timeout -k 10 300 python bench.py --no-cpu-baseline --steps ${PN_AB_STEPS:-20} --warmup ${PN_AB_WARMUP:-5} --config $c --waves $W > gpurun_out/abw_$v.json 2> gpurun_out/abw_$v.err || { echo "$c $v failed"; tail -3 gpurun_out/abw_$v.err; continue; }
      python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/abw_$v.json') if l.startswith('{')][-1]); r=d['roofline']
print('%-6s %-4s rep $rep kernel ms/step %.4f  step %.4f  frac %.4f  cand/q %.1f eval/q %.1f fb %d verified %s' % ('$c', '$v', r['kernel_ms_per_step'], d['ms_per_step'], r['frac'], d['candidates_per_query'], d['exact_evaluations_per_query'], d['fallback_queries'], d['verified']))
"
    done
  done
done
