#!/bin/bash
# Ablations of the split-f16 conv0 kernel (profiling builds, WRONG results): which phase costs the time.
TAG=${1:-sfabl}
ROOTDIR=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOTDIR/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOTDIR
for a in 0 1 2 4 8 6 7 15; do
  if [ $a = 0 ]; then L=libcasmvs_hip.so; else L=libcasmvs_sfabl$a.so; fi
  echo "== ablation $a (1 no MFMA, 2 no staging split/writes, 4 no global loads, 8 no tap reads)" >> $OUT/ablate.txt
  CASMVS_LIB_PATH=$ROOTDIR/casmvsnet_pl_amd/$L timeout 120 python tools/gpu_conv0_probe.py 2>/dev/null | grep "^level" | sed -e 's/f32 MFMA.*split-f16/split-f16/' -e 's/x4.*//' >> $OUT/ablate.txt
done
cat $OUT/ablate.txt
