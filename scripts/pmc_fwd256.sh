#!/bin/bash
# usage (GPU box): scripts/pmc_fwd256.sh <tag>  -- kernel trace + SQ counters of the d=256 forward
set -e
TAG=${1:-x}
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/fwd256_$TAG
mkdir -p $O
NS=4096 REPS=5 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $GRAFT_REPO_ROOT/scripts/fwd256_bench.py > $O/bench.log 2>&1
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  NS=4096 REPS=2 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/sq_$i -- python3 $GRAFT_REPO_ROOT/scripts/fwd256_bench.py > $O/sq_$i.log 2>&1 || echo "group $i failed"
done
grep "whole call" $O/bench.log
