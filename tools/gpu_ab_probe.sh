#!/bin/bash
# A/B of builds of the library under ONE probe command on the same box, alternating:
#   tools/gpu_ab_probe.sh <tag> <reps> "<probe command>" <libA.so> <libB.so> [...]      (library paths relative to the repo root)
TAG=$1; REPS=$2; CMD=$3; shift 3
ROOTDIR=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOTDIR/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOTDIR
for r in $(seq 1 $REPS); do for L in "$@"; do
  n=$(basename $L .so)
  echo "== $n rep $r: $CMD"
  CASMVS_LIB_PATH=$ROOTDIR/$L timeout 300 bash -c "$CMD" 2>&1 | grep -v Warning | tee -a $OUT/${n}.txt
done; done
