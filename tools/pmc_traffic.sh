#!/bin/bash
# HBM traffic + L2 hit rate + MFMA-pipe occupancy of one bench configuration (4 separate --pmc passes, kernel-trace only).
# usage (on the GPU box): bash tools/pmc_traffic.sh <tag> "<bench args>" <kernel substring>   -> gpurun_out/prof_<tag>/pmc_summary.json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=$1; BA=$2; KN=${3:-bf16_filter_kernel}
O=$R/gpurun_out/prof_$T
mkdir -p $O
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES SQ_BUSY_CYCLES"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc/p$i -o p -- python3 $R/bench.py --no-verify --no-cpu-baseline --steps 2 --warmup 1 $BA > $O/pmc_p$i.log 2>&1 || { echo pmc pass $i failed; tail -3 $O/pmc_p$i.log; exit 1; }
  echo "pass $i done"
done
python3 $R/tools/pmc_summary.py "$KN" $O/pmc > $O/pmc_summary.json
cat $O/pmc_summary.json
