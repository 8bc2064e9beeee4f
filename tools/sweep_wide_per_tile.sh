cd $GRAFT_REPO_ROOT
for rep in 1 2; do for pt in 3 5 7; do
  PN_EXP_WIDE_PER_TILE=$pt timeout -k 10 400 python bench.py --config c4 --steps 2 --warmup 1 --no-cpu-baseline --no-verify 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print('c4 per_tile $pt rep $rep: step %.1f kernel %.1f frac %.4f cand/q %.1f fb %d' % (d['ms_per_step'], r['kernel_ms_per_step'], r['frac'], d['candidates_per_query'], d['fallback_queries']))"
done; done
