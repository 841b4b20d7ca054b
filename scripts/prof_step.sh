#!/bin/bash
# usage: scripts/prof_step.sh <tag>   (run on the GPU box through gpurun)
set -e
TAG=${1:-x}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG -- python3 $GRAFT_REPO_ROOT/bench.py --mode bf16 --steps 100 --warmup 5 --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/gpurun_out/bench_$TAG.log 2>&1
grep -o "\"ms_per_step[^,]*" $GRAFT_REPO_ROOT/gpurun_out/bench_$TAG.log
