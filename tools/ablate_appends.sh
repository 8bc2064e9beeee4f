cd $GRAFT_REPO_ROOT
for c in c2 c2s8 c2s2; do
 for dz in 0 2; do
  for rep in 1 2; do
  PN_EXP_MODEL_DZ=$dz timeout -k 10 200 python bench.py --config $c --steps 1 --warmup 0 --no-verify --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print('$c dz $dz rep $rep: kernel ms %.4f step %.3f fb %d cand/q %.1f' % (r['kernel_ms_per_step'], d['ms_per_step'], d['fallback_queries'], d['candidates_per_query']))"
  done
 done
done
