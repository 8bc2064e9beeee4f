#!/bin/bash
# bench several configs in one GPU call: bench_many.sh "<cfg> <cfg> ..." [steps]
cd $GRAFT_REPO_ROOT
for c in $1; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --config $c --steps ${2:-5} --warmup 2 > gpurun_out/many_$c.json 2> gpurun_out/many_$c.err || { echo $c failed; tail -2 gpurun_out/many_$c.err; continue; }
  python3 -c "
import json
d=json.loads(open('gpurun_out/many_$c.json').read().strip().splitlines()[-1]); r=d['roofline']
print('$c', 'qps', d['value'], 'ms', d['ms_per_step'], 'kern', r['kernel_ms_per_step'], 'frac', r['frac'], 'fb', d['fallback_queries'], 'cand', d['candidates_per_query'])
"
done
