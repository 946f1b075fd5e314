#!/bin/bash
# Round 3, first GPU call: the new `prob` head (depth walk, fused regression) and the deconv skip prefetch -
# parity tests, layer probes (production build vs the no-prefetch build vs the old head in the trace build), bench lines.
TAG=${1:-r3a}
ROOTDIR=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOTDIR/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOTDIR
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 300 -p no:cacheprovider -k "prob_head or conv3d_layer or costreg" > $OUT/pytest_new.log 2>&1
echo "pytest(new) exit: $?" >> $OUT/pytest_new.log
tail -5 $OUT/pytest_new.log
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest exit: $?" >> $OUT/pytest_gpu.log
tail -5 $OUT/pytest_gpu.log
cp gpurun_out/parity_report.json $OUT/ 2>/dev/null
timeout 200 python tools/gpu_layer_probe.py 512 640 2 > $OUT/layer_probe_b2.txt 2>&1
CASMVS_LIB_PATH=$ROOTDIR/casmvsnet_pl_amd/libcasmvs_noprefetch.so LAYER_PROBE_ITEMS=deconv timeout 200 python tools/gpu_layer_probe.py 512 640 2 > $OUT/layer_probe_b2_noprefetch.txt 2>&1
CASMVS_NO_PROB_ZWALK=1 CASMVS_LIB_PATH=$ROOTDIR/casmvsnet_pl_amd/libcasmvs_trace.so LAYER_PROBE_ITEMS=prob timeout 200 python tools/gpu_layer_probe.py 512 640 2 > $OUT/layer_probe_b2_oldprob.txt 2>&1
timeout 200 python tools/gpu_layer_probe.py 512 640 1 > $OUT/layer_probe_b1.txt 2>&1
cat $OUT/layer_probe_b2.txt $OUT/layer_probe_b2_noprefetch.txt $OUT/layer_probe_b2_oldprob.txt $OUT/layer_probe_b1.txt
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
echo "bench exit: $?" >> $OUT/bench.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fuse-regress --no-batch1 > $OUT/bench_nofuse.json 2>> $OUT/bench.err
tail -3 $OUT/bench.err
python tools/show_bench.py $OUT/bench.json | head -30
python tools/show_bench.py $OUT/bench_nofuse.json | head -12
