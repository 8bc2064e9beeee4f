#!/bin/bash
# usage (GPU box): tools/sweep_args.sh "<common bench args>" reps "<args 1>" "<args 2>" ...  -- interleaved sweep of bench
# argument sets on ONE device with the product library (timings of different devices must not be compared)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
C="$1"; R=$2; shift; shift
for rep in $(seq 1 $R); do
  for X in "$@"; do
    timeout -k 10 300 python bench.py --no-verify --no-cpu-baseline --steps 20 --warmup 5 $C $X > gpurun_out/sw.json 2> gpurun_out/sw.err || { echo "[$X] failed"; tail -3 gpurun_out/sw.err; continue; }
    python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/sw.json') if l.startswith('{')][-1]); r=d['roofline']
print('[%-26s] rep $rep kernel ms/step %.4f  step %.4f  frac %.4f  cand/q %.1f eval/q %.1f fb %d' % ('$X', r['kernel_ms_per_step'], d['ms_per_step'], r['frac'], d['candidates_per_query'], d['exact_evaluations_per_query'], d['fallback_queries']))
"
  done
done
