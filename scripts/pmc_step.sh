#!/bin/bash
# usage: scripts/pmc_step.sh <tag> "<counters>"
set -e
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $@ --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG -- python3 $GRAFT_REPO_ROOT/bench.py --mode bf16 --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-graph > $GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG.log 2>&1
ls $GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG/*/ | head
