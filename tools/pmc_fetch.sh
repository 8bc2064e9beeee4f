#!/bin/bash
# L2-miss traffic + hit rate of one kernel for one bench configuration (two --pmc passes, kernel-trace only), optionally
# with a diagnostic library.  usage (GPU box): bash tools/pmc_fetch.sh <tag> "<bench args>" "<kernel substring>" [diag tag]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=$1; BA=$2; KN=$3; DT=$4
if [ -n "$DT" ]; then export PN_LIBRARY_PATH=$R/petal-neighbors_amd/libpetal_mi355x_diag_$DT.so; fi
O=$R/gpurun_out/prof_$T
mkdir -p $O
i=0
for set in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmcf/p$i -o p -- python3 $R/bench.py --no-verify --no-cpu-baseline --steps 2 --warmup 1 $BA > $O/pmcf_p$i.log 2>&1 || { echo pmc pass $i failed; tail -3 $O/pmcf_p$i.log; exit 1; }
done
python3 $R/tools/pmc_summary.py "$KN" $O/pmcf > $O/pmc_fetch.json
cat $O/pmc_fetch.json
rm -rf $O/pmcf
