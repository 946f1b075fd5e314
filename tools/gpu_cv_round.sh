#!/bin/bash
# Cost-volume kernels: parity subset, gather vs LDS probe (production build), tile / channel-split A/B (trace build).
TAG=${1:-cv}
ROOTDIR=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOTDIR/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOTDIR
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 300 -p no:cacheprovider -x \
   -k "costvol or homo_warp or partial or public or convbnrelu3d or hypotheses or softmax" > $OUT/pytest_cv.log 2>&1
echo "pytest exit: $?" >> $OUT/pytest_cv.log
tail -15 $OUT/pytest_cv.log
timeout 300 python tools/gpu_costvol_probe.py 512 640 3 1 > $OUT/probe_b1.txt 2>&1
timeout 300 python tools/gpu_costvol_probe.py 512 640 3 2 > $OUT/probe_b2.txt 2>&1
for cs in 16 32; do
  CV_PROBE_IMPLS=lds CASMVS_LIB_PATH=$ROOTDIR/casmvsnet_pl_amd/libcasmvs_trace.so CASMVS_CV_CS=$cs timeout 200 python tools/gpu_costvol_probe.py 512 640 3 1 > $OUT/probe_b1_cs$cs.txt 2>&1
done
CV_PROBE_IMPLS=lds CASMVS_LIB_PATH=$ROOTDIR/casmvsnet_pl_amd/libcasmvs_trace.so CASMVS_CV_TW=32 timeout 200 python tools/gpu_costvol_probe.py 512 640 3 1 > $OUT/probe_b1_tw32.txt 2>&1
CV_PROBE_IMPLS=lds CASMVS_LIB_PATH=$ROOTDIR/casmvsnet_pl_amd/libcasmvs_cvdirect.so timeout 200 python tools/gpu_costvol_probe.py 512 640 3 1 > $OUT/probe_b1_direct.txt 2>&1
CV_PROBE_G=8 timeout 200 python tools/gpu_costvol_probe.py 512 640 3 1 > $OUT/probe_b1_gwc8.txt 2>&1
cat $OUT/probe_b1.txt
for f in probe_b2 probe_b1_cs16 probe_b1_cs32 probe_b1_tw32 probe_b1_direct; do echo "== $f"; grep -h "lds" $OUT/$f.txt | grep -v bitwise; done
cat $OUT/probe_b1_gwc8.txt | grep -v homo
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
python tools/show_bench.py $OUT/bench.json 2>/dev/null | head -40 || cut -c1-400 $OUT/bench.json
