#!/bin/bash
# GPU box: phase stamps of k_isab1_fwd256_ab (diagnostic build of that one file); one stamp position
# per run (PCA_AB_STAMPSEL), then all of them together
set -e
cd $GRAFT_REPO_ROOT
touch point-cloud-audio_amd/csrc/d256_fused.hip
HIPCC="/opt/rocm/bin/hipcc -DPCA_FWD_STAMPS" bash point-cloud-audio_amd/csrc/build.sh > /dev/null
for s in 1 2 3 4 5 6; do PCA_AB_STAMPSEL=$s python scripts/experiments/ab_stamps.py; done
PCA_AB_STAMPSEL=15 python scripts/experiments/ab_stamps.py
