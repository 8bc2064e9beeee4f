#!/bin/bash
# PMC passes over the bf16 bench (separate passes, kernel-trace only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH" \
           "SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL" \
           "SQ_IFETCH SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_VMEM SQ_BUSY_CYCLES" \
           "SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmcbf/p$i -o p -- python3 $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 $1 > $R/gpurun_out/pmcbf_p$i.log 2>&1 || { echo pass $i failed; tail -3 $R/gpurun_out/pmcbf_p$i.log; exit 1; }
done
python3 $R/tools/pmc_summary.py bf16_filter_kernel $R/gpurun_out/pmcbf
