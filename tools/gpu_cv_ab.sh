#!/bin/bash
# A/B of the LDS cost-volume plan (trace build): plane groups x channel split x tile width, batch 1 and 2
TAG=${1:-cvab}
ROOTDIR=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOTDIR/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOTDIR
export CASMVS_LIB_PATH=$ROOTDIR/casmvsnet_pl_amd/libcasmvs_trace.so
for B in 1 2; do
  echo "== gather B=$B"; CV_PROBE_IMPLS=gather timeout 120 python tools/gpu_costvol_probe.py 512 640 3 $B 2>/dev/null | grep "depth=smooth" | grep -v bitwise
  for pg in 1 2; do for cs in 8 16; do for tw in 64 32; do
    echo "== lds B=$B PG=$pg CS=$cs TW=$tw"
    CV_PROBE_IMPLS=lds CASMVS_CV_PG=$pg CASMVS_CV_CS=$cs CASMVS_CV_TW=$tw timeout 120 python tools/gpu_costvol_probe.py 512 640 3 $B 2>/dev/null | grep "depth=\|homo_warp" | grep -v bitwise
  done; done; done
done | tee $OUT/ab.txt
