#!/bin/bash
# usage (GPU box): scripts/prof_env.sh <tag> <config> <batch>   (environment is inherited) -- kernel trace of bench.py
set -e
TAG=$1; CFG=$2; B=$3
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 $R/bench.py --config $CFG --batch $B --steps 20 --warmup 3 --windows 1 --no-cpu-baseline --no-roofline > $R/gpurun_out/bench_$TAG.log 2>&1
