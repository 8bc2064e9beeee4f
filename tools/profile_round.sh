#!/bin/bash
# rocprofv3 evidence of a round for one configuration: kernel-trace stats, the scout / main split, the fill / copy census,
# PMC traffic + occupancy of the main kernel.  usage (GPU box): bash tools/profile_round.sh <tag> "<bench args>" "<main kernel substring>"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=$1; BA=$2; KN=${3:-bf16_filter}
O=$R/gpurun_out/prof_$T
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 $R/bench.py --no-verify --no-cpu-baseline --steps 20 --warmup 3 $BA > $O/stats.log 2>&1 || { echo stats pass failed; tail -3 $O/stats.log; exit 1; }
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
find $O/stats -name "*kernel_trace.csv" | head -1 | xargs -I{} python3 $R/tools/fill_copy_census.py {} > $O/fill_copy_census.json
cat $O/fill_copy_census.json
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES SQ_BUSY_CYCLES"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc/p$i -o p -- python3 $R/bench.py --no-verify --no-cpu-baseline --steps 3 --warmup 1 $BA > $O/pmc_p$i.log 2>&1 || { echo pmc pass $i failed; tail -3 $O/pmc_p$i.log; exit 1; }
done
python3 $R/tools/pmc_summary.py "$KN" $O/pmc > $O/pmc_summary.json
cat $O/pmc_summary.json
head -6 $O/kernel_stats.csv | cut -c1-200
tail -1 $O/stats.log | cut -c1-600
rm -rf $O/stats $O/pmc
