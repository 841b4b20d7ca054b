#!/bin/bash
# GPU box: ablation timings of k_isab1_fwd256_ab (diagnostic build of that one file: -DPCA_FWD_ABLATE)
set -e
cd $GRAFT_REPO_ROOT
# (diagnostic objects and library live in their own directory; the product library is swapped in for the
#  measurement and restored on every exit path)
D=point-cloud-audio_amd/pca_hip
cp $D/libpca_hip.so /tmp/lib_keep.so
trap 'cp /tmp/lib_keep.so $D/libpca_hip.so' EXIT
PCA_EXTRA_FLAGS="-DPCA_FWD_ABLATE" PCA_BUILD_DIR=/tmp/pca_build_ablate PCA_OUT=/tmp/libpca_ablate.so \
  bash point-cloud-audio_amd/csrc/build.sh > /dev/null
cp /tmp/libpca_ablate.so $D/libpca_hip.so
for m in ${MASKS:-0 1 2 3 4 8 16 32 64 96 99 127}; do
  echo -n "ablate=$m: "
  PCA_AB_ABLATE=$m NS=2048 REPS=20 python scripts/fwd256_bench.py 2>&1 | grep "whole call" | cut -c40-80
done
