#!/bin/bash
# usage (GPU box): scripts/pmc_fwd256.sh <tag> [N]  -- kernel trace + SQ counters of the d=256 forward
set -e
TAG=${1:-x}
N=${2:-2048}
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/fwd256_$TAG
rm -rf $O && mkdir -p $O
NS=$N REPS=10 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $GRAFT_REPO_ROOT/scripts/fwd256_bench.py > $O/bench.log 2>&1
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  NS=$N REPS=2 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/sq_$i -- python3 $GRAFT_REPO_ROOT/scripts/fwd256_bench.py > $O/sq_$i.log 2>&1 || echo "group $i failed"
done
grep "whole call" $O/bench.log
python3 $GRAFT_REPO_ROOT/scripts/sq_summary.py $O k_isab1 > $O/summary.txt 2>&1 || true
cat $O/summary.txt
