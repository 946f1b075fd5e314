#!/bin/bash
# Ablations of the LDS cost-volume kernel (trace build; results wrong, timings informative)
TAG=${1:-cvabl}
ROOTDIR=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOTDIR/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOTDIR
export CV_PROBE_IMPLS=${CV_PROBE_IMPLS:-gather,lds} CASMVS_LIB_PATH=$ROOTDIR/casmvsnet_pl_amd/libcasmvs_trace.so
for abl in ${ABLS:-0 1 2 3 4}; do
  echo "== ablate=$abl"
  CASMVS_CV_ABLATE=$abl timeout 120 python tools/gpu_costvol_probe.py 512 640 3 1 2>/dev/null | grep "depth=smooth\|homo_warp" | grep -v bitwise
done | tee $OUT/ablate.txt
