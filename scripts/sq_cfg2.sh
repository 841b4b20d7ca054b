set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/sqcfg2/sq_$i -- python3 $R/bench.py --steps 6 --warmup 2 --windows 1 --no-cpu-baseline --no-roofline --no-graph > $O/sqcfg2_$i.log 2>&1 || echo "cfg2 group $i failed"
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/sqcfg2/trace -- python3 $R/bench.py --steps 20 --warmup 3 --windows 1 --no-cpu-baseline --no-roofline --no-graph > $O/sqcfg2_trace.log 2>&1
echo "cfg2 SQ passes done"
