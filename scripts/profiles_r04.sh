#!/bin/bash
# usage (GPU box, through gpurun): scripts/profiles_r04.sh [part]   part = bench | stats | pmc | sq | fwd | isab | all
# Everything under profiles/r04_* (see scripts/copy_profiles_r04.sh).  Outputs land in gpurun_out/r04/.
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04
PART=${1:-all}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
GRP=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT")
if [ $PART = bench ] || [ $PART = all ]; then
  python3 $R/bench.py > $O/bf16_cfg2_bench.json 2> $O/bf16_cfg2_bench.err
  PCA_SET128=0 python3 $R/bench.py --no-cpu-baseline > $O/bf16_cfg2_perblock_bench.json 2>> $O/bench.err
  PCA_SET128_HEAD=0 python3 $R/bench.py --no-cpu-baseline --no-roofline > $O/bf16_cfg2_headlaunch_bench.json 2>> $O/bench.err
  python3 $R/bench.py --mode f32 --no-cpu-baseline > $O/f32_cfg2_bench.json 2>> $O/bench.err
  python3 $R/bench.py --config cfg3 --no-cpu-baseline > $O/bf16_cfg3_bench.json 2>> $O/bench.err
  python3 $R/bench.py --config cfg4 --steps 50 --warmup 60 --no-cpu-baseline > $O/bf16_cfg4_bench.json 2>> $O/bench.err
  python3 $R/bench.py --config cfg4 --mode fp8 --steps 50 --warmup 60 --no-cpu-baseline > $O/fp8_cfg4_bench.json 2>> $O/bench.err
  python3 $R/bench.py --config cfg5 --steps 50 --warmup 60 --no-cpu-baseline > $O/bf16_cfg5_bench.json 2>> $O/bench.err
  python3 $R/bench.py --config cfg5 --mode fp8 --steps 50 --warmup 60 --no-cpu-baseline > $O/fp8_cfg5_bench.json 2>> $O/bench.err
  python3 $R/bench.py --batch 1024 --steps 50 --warmup 5 --no-cpu-baseline > $O/bf16_cfg2_B1024_bench.json 2>> $O/bench.err
  python3 $R/bench.py --config fst --steps 50 --warmup 5 --no-cpu-baseline --no-roofline > $O/bf16_fst_bench.json 2>> $O/bench.err
  python3 $R/bench.py --config 3st --steps 50 --warmup 5 --no-cpu-baseline --no-roofline > $O/bf16_3st_bench.json 2>> $O/bench.err
  (python3 $R/scripts/infer_bench.py fst 128; python3 $R/scripts/infer_bench.py 3st 16; python3 $R/scripts/infer_bench.py fst 8; python3 $R/scripts/infer_bench.py 3st 8) > $O/infer.txt 2>> $O/bench.err
  echo "bench lines done"
fi
if [ $PART = stats ] || [ $PART = all ]; then
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cfg2 -- python3 $R/bench.py --no-cpu-baseline > $O/stats_cfg2.json 2> $O/stats_cfg2.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cfg4 -- python3 $R/bench.py --config cfg4 --steps 20 --warmup 3 --windows 1 --no-cpu-baseline --no-roofline > $O/stats_cfg4.json 2> $O/stats_cfg4.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_fst -- python3 $R/bench.py --config fst --steps 20 --warmup 3 --windows 1 --no-cpu-baseline --no-roofline > $O/stats_fst.json 2> $O/stats_fst.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_3st -- python3 $R/bench.py --config 3st --steps 20 --warmup 3 --windows 1 --no-cpu-baseline --no-roofline > $O/stats_3st.json 2> $O/stats_3st.err
  echo "kernel stats done"
fi
if [ $PART = pmc ] || [ $PART = all ]; then
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_$c -- python3 $R/bench.py --mode bf16 --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-graph > $R/gpurun_out/pmc_$c.log 2>&1
  done
  echo "traffic passes done"
fi
if [ $PART = sq ] || [ $PART = all ]; then
  for cfg in cfg2 cfg4 fst; do
    i=0
    if [ $cfg = cfg2 ]; then A="--steps 6 --warmup 2"; else A="--config $cfg --steps 3 --warmup 1"; fi
    for grp in "${GRP[@]}"; do
      i=$((i+1))
      rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/sq$cfg/sq_$i -- python3 $R/bench.py $A --windows 1 --no-cpu-baseline --no-roofline --no-graph > $O/sq${cfg}_$i.log 2>&1 || echo "$cfg group $i failed"
    done
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/sq$cfg/trace -- python3 $R/bench.py $A --windows 1 --no-cpu-baseline --no-roofline --no-graph > $O/sq${cfg}_trace.log 2>&1
  done
  echo "SQ passes done"
fi
if [ $PART = fwd ] || [ $PART = all ]; then
  for v in ab; do
    D=$O/fwd256_$v
    mkdir -p $D
    if [ $v = onerole ]; then export PCA_D256_AB=0; else unset PCA_D256_AB; fi
    NS=2048,4096 REPS=10 rocprofv3 --kernel-trace --stats --output-format csv -d $D/trace -- python3 $R/scripts/fwd256_bench.py > $D/bench.log 2>&1
    i=0
    for grp in "${GRP[@]}"; do
      i=$((i+1))
      NS=2048 REPS=2 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $D/sq_$i -- python3 $R/scripts/fwd256_bench.py > $D/sq_$i.log 2>&1 || echo "fwd256 $v group $i failed"
    done
  done
  unset PCA_D256_AB
  echo "forward passes done"
fi
if [ $PART = isab ] || [ $PART = fwd ] || [ $PART = all ]; then
  mkdir -p $O/isab256
  N=2048 REPS=10 rocprofv3 --kernel-trace --stats --output-format csv -d $O/isab256/trace -- python3 $R/scripts/isab256_fwd_bench.py > $O/isab256/bench.log 2>&1 || true
  mkdir -p $O/isab256fb
  N=2048 REPS=10 rocprofv3 --kernel-trace --stats --output-format csv -d $O/isab256fb/trace -- python3 $R/scripts/isab256_fwdbwd_bench.py > $O/isab256fb/bench.log 2>&1 || true
  echo "ISAB forward / forward + backward passes done"
fi
echo "all done"
