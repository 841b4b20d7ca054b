#!/bin/bash
# GPU box: ablation timings of k_isab1_fwd256_ab (diagnostic build of that one file: -DPCA_FWD_ABLATE)
set -e
cd $GRAFT_REPO_ROOT
touch point-cloud-audio_amd/csrc/d256_fused.hip
HIPCC="/opt/rocm/bin/hipcc -DPCA_FWD_ABLATE" bash point-cloud-audio_amd/csrc/build.sh > /dev/null
for m in ${MASKS:-0 1 2 3 4 8 16 32 64 96 99 127}; do
  echo -n "ablate=$m: "
  PCA_AB_ABLATE=$m NS=2048 REPS=20 python scripts/fwd256_bench.py 2>&1 | grep "whole call" | cut -c40-80
done
