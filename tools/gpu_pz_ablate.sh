#!/bin/bash
# Ablations of the z-walk `prob` head (profiling builds, WRONG results): which part of a plane step costs the time.
TAG=${1:-pzabl}
ROOTDIR=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOTDIR/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOTDIR
export LAYER_PROBE_ITEMS=prob LAYER_PROBE_ZCHUNKS=8 LAYER_PROBE_REPS=6
for a in 0 1 2 4 8 16 32 3 7 63; do
  if [ $a = 0 ]; then L=libcasmvs_hip.so; else L=libcasmvs_pzabl$a.so; fi
  echo "== ablation $a" >> $OUT/ablate.txt
  for B in 2 1; do
    CASMVS_LIB_PATH=$ROOTDIR/casmvsnet_pl_amd/$L timeout 120 python tools/gpu_layer_probe.py 512 640 $B 2>/dev/null | grep "prob head" | sed -e 's/| softmax.*//' -e "s/^/b$B /" >> $OUT/ablate.txt
  done
done
cat $OUT/ablate.txt
