#!/bin/bash
# GPU box: phase stamps of k_mab1_bwd (diagnostic build of that one file), cfg2
set -e
cd $GRAFT_REPO_ROOT
touch point-cloud-audio_amd/csrc/mab1_bwd_bf16.hip
HIPCC="/opt/rocm/bin/hipcc -DPCA_DEBUG_CLOCKS" bash point-cloud-audio_amd/csrc/build.sh > /dev/null
for wg in 0; do
  PCA_DBG_WG=$wg python bench.py --steps 3 --warmup 2 --windows 1 --no-graph --no-cpu-baseline --no-roofline 2>&1 | grep "stamps\|per-workgroup"
done
