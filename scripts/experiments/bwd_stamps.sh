#!/bin/bash
# GPU box: phase stamps of k_mab1_bwd (diagnostic build of that one file), cfg2
set -e
cd $GRAFT_REPO_ROOT
# (diagnostic objects and library live in their own directory; the product library is swapped in for the
#  measurement and restored on every exit path)
D=point-cloud-audio_amd/pca_hip
cp $D/libpca_hip.so /tmp/lib_keep.so
trap 'cp /tmp/lib_keep.so $D/libpca_hip.so' EXIT
PCA_EXTRA_FLAGS="-DPCA_DEBUG_CLOCKS" PCA_BUILD_DIR=/tmp/pca_build_clocks PCA_OUT=/tmp/libpca_clocks.so \
  bash point-cloud-audio_amd/csrc/build.sh > /dev/null
cp /tmp/libpca_clocks.so $D/libpca_hip.so
for wg in 0; do
  PCA_DBG_WG=$wg python bench.py --steps 3 --warmup 2 --windows 1 --no-graph --no-cpu-baseline --no-roofline 2>&1 | grep "stamps\|per-workgroup"
done
