#!/bin/bash
# GPU box: same-box A/B of two prebuilt libraries (point-cloud-audio_amd/pca_hip/ab/libA.so, libB.so:
# built in the container, e.g. HEAD against the working tree), measured alternately - box-to-box
# differences are +-5 %, larger than most single changes.   usage: ab_libs.sh [reps] -- command...
set -e
cd $GRAFT_REPO_ROOT
REPS=${1:-3}; shift; shift
D=point-cloud-audio_amd/pca_hip
cp $D/libpca_hip.so /tmp/lib_keep.so
trap 'cp /tmp/lib_keep.so $D/libpca_hip.so' EXIT      # a failing command must not leave lib B installed
for rep in $(seq $REPS); do
  for v in A B; do
    cp $D/ab/lib$v.so $D/libpca_hip.so
    echo "== rep $rep lib $v"
    "$@"
  done
done
