#!/bin/bash
# GPU box: phase stamps of k_isab1_fwd256_ab (diagnostic build of that one file)
set -e
cd $GRAFT_REPO_ROOT
touch point-cloud-audio_amd/csrc/d256_fused.hip
HIPCC="/opt/rocm/bin/hipcc -DPCA_FWD_STAMPS" bash point-cloud-audio_amd/csrc/build.sh > /dev/null
python scripts/experiments/ab_stamps.py
