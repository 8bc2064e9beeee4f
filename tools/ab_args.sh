#!/bin/bash
# usage (GPU box): tools/ab_args.sh "<common bench args>" "<args A>" "<args B>" [reps]  -- interleaved A/B of two bench
# argument sets on ONE device with the product library (timings of different devices must not be compared)
cd $GRAFT_REPO_ROOT
C="$1"; A="$2"; B="$3"; R=${4:-3}
for rep in $(seq 1 $R); do
  for v in A B; do
    if [ $v = A ]; then X="$A"; else X="$B"; fi
    timeout -k 10 300 python bench.py --no-verify --no-cpu-baseline --steps 20 --warmup 5 $C $X > gpurun_out/abx_$v.json 2> gpurun_out/abx_$v.err || { echo "$v failed"; tail -3 gpurun_out/abx_$v.err; continue; }
    python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/abx_$v.json') if l.startswith('{')][-1]); r=d['roofline']
print('%s [%-24s] rep $rep kernel ms/step %.4f  step %.4f  frac %.4f  cand/q %.1f fb %d' % ('$v', '$X', r['kernel_ms_per_step'], d['ms_per_step'], r['frac'], d['candidates_per_query'], d['fallback_queries']))
"
  done
done
