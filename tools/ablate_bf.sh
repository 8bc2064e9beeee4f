#!/bin/bash
# usage: ablate_bf.sh "<diag flags>" "<bench args>" tag
cd $GRAFT_REPO_ROOT
PN_DIAG_FLAGS="$1" python petal-neighbors_amd/build.py --force > /dev/null 2>&1 || { echo build failed; exit 1; }
timeout -k 10 300 PN_LIBRARY_PATH=$GRAFT_REPO_ROOT/petal-neighbors_amd/libpetal_mi355x_diag.so python bench.py --no-verify --no-cpu-baseline --steps 10 $2 > gpurun_out/abl_$3.json 2> gpurun_out/abl_$3.err || { echo run failed; tail -3 gpurun_out/abl_$3.err; }
python3 -c "
import json,sys
d=json.loads(open('gpurun_out/abl_$3.json').read().strip().splitlines()[-1]); r=d['roofline']
print('$3', 'kern', r['kernel_ms_per_step'], 'ms', d['ms_per_step'], 'fb', d['fallback_queries'], 'cand', d['candidates_per_query'])
"
