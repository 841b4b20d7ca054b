#!/bin/bash
# usage (GPU box, through gpurun): scripts/profiles_r02.sh
# Everything under profiles/r02_* : bench lines, rocprofv3 kernel statistics of the same commands,
# the HBM traffic PMC passes of the cfg2 step, SQ counter passes (MFMA utilisation) of the cfg4 step
# and of the d=256 forward at B=128, N=2048 / 4096.  Outputs land in gpurun_out/r02/.
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# --- bench lines (the default invocation is the driver's) ---
python3 $R/bench.py > $O/bf16_cfg2_bench.json 2> $O/bf16_cfg2_bench.err
python3 $R/bench.py --mode f32 --no-cpu-baseline > $O/f32_cfg2_bench.json 2>> $O/bench.err
python3 $R/bench.py --config cfg3 --no-cpu-baseline > $O/bf16_cfg3_bench.json 2>> $O/bench.err
python3 $R/bench.py --config cfg4 --steps 50 --warmup 5 --no-cpu-baseline > $O/bf16_cfg4_bench.json 2>> $O/bench.err
python3 $R/bench.py --config cfg4 --mode fp8 --steps 50 --warmup 5 --no-cpu-baseline > $O/fp8_cfg4_bench.json 2>> $O/bench.err
python3 $R/bench.py --config cfg5 --steps 50 --warmup 5 --no-cpu-baseline > $O/bf16_cfg5_bench.json 2>> $O/bench.err
python3 $R/bench.py --config cfg5 --mode fp8 --steps 50 --warmup 5 --no-cpu-baseline > $O/fp8_cfg5_bench.json 2>> $O/bench.err
python3 $R/bench.py --batch 1024 --steps 50 --warmup 5 --no-cpu-baseline > $O/bf16_cfg2_B1024_bench.json 2>> $O/bench.err
echo "bench lines done"
# --- kernel statistics ---
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cfg2 -- python3 $R/bench.py --no-cpu-baseline > $O/stats_cfg2.json 2> $O/stats_cfg2.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cfg4 -- python3 $R/bench.py --config cfg4 --steps 20 --warmup 3 --windows 1 --no-cpu-baseline --no-roofline > $O/stats_cfg4.json 2> $O/stats_cfg4.err
echo "kernel stats done"
# --- HBM traffic of the cfg2 step (separate passes per counter, as the guide prescribes) ---
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_$c -- python3 $R/bench.py --mode bf16 --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-graph > $R/gpurun_out/pmc_$c.log 2>&1
done
echo "traffic passes done"
# --- SQ counters: the cfg4 step ---
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/sqcfg4/sq_$i -- python3 $R/bench.py --config cfg4 --steps 3 --warmup 1 --windows 1 --no-cpu-baseline --no-roofline --no-graph > $O/sqcfg4_$i.log 2>&1 || echo "cfg4 group $i failed"
done
mkdir -p $O/sqcfg4/trace && cp -r $O/stats_cfg4/* $O/sqcfg4/trace/
echo "cfg4 SQ passes done"
# --- SQ counters: the cfg2 step (the driver's bench workload) ---
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/sqcfg2/sq_$i -- python3 $R/bench.py --steps 6 --warmup 2 --windows 1 --no-cpu-baseline --no-roofline --no-graph > $O/sqcfg2_$i.log 2>&1 || echo "cfg2 group $i failed"
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/sqcfg2/trace -- python3 $R/bench.py --steps 20 --warmup 3 --windows 1 --no-cpu-baseline --no-roofline --no-graph > $O/sqcfg2_trace.log 2>&1
echo "cfg2 SQ passes done"
# --- SQ counters: the d=256 forward (single launch, and the two-launch pair it replaced) ---
for v in fused pair; do
  D=$O/fwd256_$v
  mkdir -p $D
  if [ $v = pair ]; then export PCA_D256_FUSED=0; else unset PCA_D256_FUSED; fi
  NS=2048,4096 REPS=10 rocprofv3 --kernel-trace --stats --output-format csv -d $D/trace -- python3 $R/scripts/fwd256_bench.py > $D/bench.log 2>&1
  i=0
  for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"; do
    i=$((i+1))
    NS=2048 REPS=2 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $D/sq_$i -- python3 $R/scripts/fwd256_bench.py > $D/sq_$i.log 2>&1 || echo "fwd256 $v group $i failed"
  done
done
unset PCA_D256_FUSED
echo "all done"
