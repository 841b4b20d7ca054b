#!/bin/bash
# GPU box: phase stamps of k_isab1_fwd256_ab (diagnostic build of that one file); one stamp position
# per run (PCA_AB_STAMPSEL), then all of them together
set -e
cd $GRAFT_REPO_ROOT
# (diagnostic objects and library live in their own directory; the product library is swapped in for the
#  measurement and restored on every exit path)
D=point-cloud-audio_amd/pca_hip
cp $D/libpca_hip.so /tmp/lib_keep.so
trap 'cp /tmp/lib_keep.so $D/libpca_hip.so' EXIT
PCA_EXTRA_FLAGS="-DPCA_FWD_STAMPS" PCA_BUILD_DIR=/tmp/pca_build_stamps PCA_OUT=/tmp/libpca_stamps.so \
  bash point-cloud-audio_amd/csrc/build.sh > /dev/null
cp /tmp/libpca_stamps.so $D/libpca_hip.so
for s in 1 2 3 4 5 6; do PCA_AB_STAMPSEL=$s python scripts/experiments/ab_stamps.py; done
PCA_AB_STAMPSEL=15 python scripts/experiments/ab_stamps.py
