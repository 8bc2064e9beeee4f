#!/bin/bash
# usage (GPU box): tools/sweep_min_per_tile.sh <config> "<values>" [steps] -- at least that many workgroups (row ranges) per
# query tile where the plan would take fewer (PN_EXP_MIN_PER_TILE): shorter runs drift less, more segments cost more
cd $GRAFT_REPO_ROOT
for v in ${2:-1 2 4}; do
  PN_DEBUG_PLAN=1 PN_EXP_MIN_PER_TILE=$v timeout -k 10 600 python bench.py --config $1 --steps ${3:-1} --warmup 1 --no-cpu-baseline --no-verify 2> gpurun_out/mpt.err | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print('$1 min_per_tile $v: step %.1f kernel %.1f frac %.4f cand/q %.1f fb %d' % (d['ms_per_step'], r['kernel_ms_per_step'], r['frac'], d['candidates_per_query'], d['fallback_queries']))"
  grep -h "bf16_plan:" gpurun_out/mpt.err | sort | uniq -c | head -2 | cut -c1-200
done
