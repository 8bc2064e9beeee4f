#!/bin/bash
# rocprofv3 evidence for the default bench (C2): kernel-trace stats + separate PMC passes (kernel-trace only).
# usage (on the GPU box): bash tools/profile_c2.sh <tag> ["<bench args>" [<kernel>]]   -> gpurun_out/prof_<tag>/...
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=${1:-c2}
O=$R/gpurun_out/prof_$T
BA=${2:-}
KN=${3:-bf16_filter_kernel}
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 $R/bench.py --no-verify --no-cpu-baseline --steps 20 --warmup 3 $BA > $O/stats.log 2>&1 || { echo stats pass failed; tail -3 $O/stats.log; exit 1; }
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES SQ_BUSY_CYCLES" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc/p$i -o p -- python3 $R/bench.py --no-verify --no-cpu-baseline --steps 3 --warmup 1 $BA > $O/pmc_p$i.log 2>&1 || { echo pmc pass $i failed; tail -3 $O/pmc_p$i.log; exit 1; }
done
python3 $R/tools/pmc_summary.py "$KN" $O/pmc > $O/pmc_summary.json
cat $O/pmc_summary.json
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
head -8 $O/kernel_stats.csv
tail -1 $O/stats.log
