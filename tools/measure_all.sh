#!/bin/bash
# usage (GPU box): tools/measure_all.sh <tag>  -- bench lines of every BASELINE config that fits one GPU + shard shapes of C2
cd $GRAFT_REPO_ROOT
T=${1:-rXX}
O=gpurun_out/measure_$T
mkdir -p $O
for c in c2 c2k1 c3s c4s c5s c2s2 c2s4 c2s8 c3 c4 c5shard; do
  steps=20; [ $c = c3 ] && steps=5; [ $c = c4 ] && steps=3; [ $c = c5shard ] && steps=2
  timeout -k 10 600 python bench.py --config $c --steps $steps --warmup 2 --no-cpu-baseline > $O/$c.json 2> $O/$c.err || { echo "$c failed"; tail -2 $O/$c.err; continue; }
  python3 -c "
import json
d=json.loads([l for l in open('$O/$c.json') if l.startswith('{')][-1]); r=d['roofline']
print('%-8s q/s %12.0f  ms/step %9.3f  kernel ms/step %9.3f  frac %.4f  verified %s  fallback %d  cand/q %.1f eval/q %.1f' % ('$c', d['value'], d['ms_per_step'], r['kernel_ms_per_step'], r['frac'], d['verified'], d['fallback_queries'], d['candidates_per_query'], d['exact_evaluations_per_query']))
"
done
