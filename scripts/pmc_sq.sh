#!/bin/bash
# usage (GPU box): scripts/pmc_sq.sh   -- SQ activity counters of the step's kernels, one pass each group
set -e
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_INST_LEVEL_VMEM"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/sq_$i -- python3 $GRAFT_REPO_ROOT/bench.py --mode bf16 --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-graph > $GRAFT_REPO_ROOT/gpurun_out/sq_$i.log 2>&1 || echo "group $i failed"
done
echo done
