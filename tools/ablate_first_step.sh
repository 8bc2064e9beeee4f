#!/bin/bash
cd $GRAFT_REPO_ROOT
PN_DIAG_FLAGS="$1" python petal-neighbors_amd/build.py --force > /dev/null 2>&1 || { echo build failed; exit 1; }
for i in 1 2 3; do
timeout -k 10 300 PN_LIBRARY_PATH=$GRAFT_REPO_ROOT/petal-neighbors_amd/libpetal_mi355x_diag.so python bench.py --no-verify --no-cpu-baseline --steps 1 --warmup 0 $3 > gpurun_out/abl1_$2.json 2> gpurun_out/abl1_$2.err || { echo run failed; tail -3 gpurun_out/abl1_$2.err; }
python3 -c "
import json
d=json.loads(open('gpurun_out/abl1_$2.json').read().strip().splitlines()[-1]); r=d['roofline']
print('$2', 'kern/step', r['kernel_ms_per_step'], 'fb', d['fallback_queries'])
"
done
