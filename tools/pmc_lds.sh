#!/bin/bash
# LDS / wait-state counters of the dominant kernel for one bench configuration (separate --pmc passes, kernel-trace only).
# usage (GPU box): bash tools/pmc_lds.sh <tag> "<bench args>" <kernel substring>   -> gpurun_out/prof_<tag>/pmc_lds.json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=$1; BA=$2; KN=${3:-bf16_filter}
O=$R/gpurun_out/prof_$T
mkdir -p $O
i=0
for set in "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_LDS"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmcl/p$i -o p -- python3 $R/bench.py --no-verify --no-cpu-baseline --steps 2 --warmup 1 $BA > $O/pmcl_p$i.log 2>&1 || { echo pmc pass $i failed; tail -3 $O/pmcl_p$i.log; exit 1; }
done
python3 $R/tools/pmc_summary.py "$KN" $O/pmcl > $O/pmc_lds.json
cat $O/pmc_lds.json
