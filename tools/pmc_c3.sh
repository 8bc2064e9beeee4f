#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_c3
mkdir -p $O
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc/p1 -o p -- python3 $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 --config c3 > $O/p1.log 2>&1 || { echo fail; tail -3 $O/p1.log; exit 1; }
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc/p2 -o p -- python3 $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 --config c3 > $O/p2.log 2>&1 || { echo fail; tail -3 $O/p2.log; exit 1; }
python3 $R/tools/pmc_summary.py bf16_filter_kernel $O/pmc
tail -1 $O/p1.log | cut -c1-300
