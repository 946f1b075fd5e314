#!/bin/bash
# A/B of the libcv_<name>.so builds (tools/build_cv_variants.sh) on one box: cost-volume probe at batch 1 and 2,
# bitwise check LDS vs gather per build, then the cost-volume parity tests on the LAST build named.
#   tools/gpu_cv_variants.sh <tag> name1 name2 ...
TAG=${1:-cvvar}; shift
ROOTDIR=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOTDIR/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOTDIR
export TMPDIR=/tmp
for B in 2 1; do for n in "$@"; do
  echo "== $n B=$B"
  CV_PROBE_IMPLS=gather,lds CASMVS_LIB_PATH=$ROOTDIR/casmvsnet_pl_amd/libcv_$n.so timeout 200 python tools/gpu_costvol_probe.py 512 640 3 $B 2>&1 \
    | grep -v "depth=noisy  gather" | grep -E "lds|bitwise|rror" | sed 's/ V=3//; s/ G=1//'
done; done | tee $OUT/ab.txt
last=${@: -1}
CASMVS_LIB_PATH=$ROOTDIR/casmvsnet_pl_amd/libcv_$last.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 300 -p no:cacheprovider -x \
   -k "costvol or homo_warp or partial or public" > $OUT/pytest_$last.log 2>&1
echo "pytest ($last) exit: $?" | tee -a $OUT/pytest_$last.log
tail -5 $OUT/pytest_$last.log
