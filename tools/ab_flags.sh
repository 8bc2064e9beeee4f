#!/bin/bash
# A/B of a compile-time variant on the GPU box: ab_flags.sh "<flags>" "<bench args>" tag  (warm steps, results valid)
cd $GRAFT_REPO_ROOT
PN_DIAG_FLAGS="$1" python petal-neighbors_amd/build.py --force > /dev/null 2>&1 || { echo build failed; exit 1; }
timeout -k 10 300 PN_LIBRARY_PATH=$GRAFT_REPO_ROOT/petal-neighbors_amd/libpetal_mi355x_diag.so python bench.py --no-verify --no-cpu-baseline $2 > gpurun_out/ab_$3.json 2> gpurun_out/ab_$3.err || { echo run failed; tail -3 gpurun_out/ab_$3.err; exit 1; }
python3 -c "
import json
d=json.loads(open('gpurun_out/ab_$3.json').read().strip().splitlines()[-1]); r=d['roofline']
print('$3', 'qps', d['value'], 'ms', d['ms_per_step'], 'kern', r['kernel_ms_per_step'], 'frac', r['frac'], 'fb', d['fallback_queries'], 'cand', d['candidates_per_query'])
"
