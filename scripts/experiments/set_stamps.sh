#!/bin/bash
# GPU box: phase stamps of k_set128_fwd (diagnostic build in its own object directory)
set -e
cd $GRAFT_REPO_ROOT
D=point-cloud-audio_amd/pca_hip
cp $D/libpca_hip.so /tmp/lib_keep.so
trap 'cp /tmp/lib_keep.so $D/libpca_hip.so' EXIT
PCA_EXTRA_FLAGS="-DPCA_SET_STAMPS" PCA_BUILD_DIR=/tmp/pca_build_setst PCA_OUT=/tmp/libpca_setst.so \
  bash point-cloud-audio_amd/csrc/build.sh > /dev/null
cp /tmp/libpca_setst.so $D/libpca_hip.so
python scripts/experiments/set_stamps.py
