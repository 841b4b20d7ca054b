#!/bin/bash
# usage (GPU box): scripts/prof_batch.sh <tag> <config> <batch> [steps] -- kernel trace of bench.py at another batch
set -e
TAG=${1:-x}; CFG=${2:-cfg2}; B=${3:-128}; STEPS=${4:-20}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 $R/bench.py --config $CFG --batch $B --steps $STEPS --warmup 3 --windows 1 --no-cpu-baseline --no-roofline > $R/gpurun_out/bench_$TAG.log 2>&1
grep -o "\"ms_per_step[^,]*" $R/gpurun_out/bench_$TAG.log
