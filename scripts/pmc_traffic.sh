#!/bin/bash
# HBM traffic of the step's kernels: FETCH_SIZE and WRITE_SIZE in SEPARATE passes
set -e
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$c -- python3 $GRAFT_REPO_ROOT/bench.py --mode bf16 --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-graph > $GRAFT_REPO_ROOT/gpurun_out/pmc_$c.log 2>&1
done
ls $GRAFT_REPO_ROOT/gpurun_out/pmc_FETCH_SIZE/*/ | head -3
