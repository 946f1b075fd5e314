#!/bin/bash
# Round 3, call f: double-buffered fused FPN tail: tests, bench (fused vs the three-step tail), files -> depth maps throughput.
TAG=${1:-r3f}
ROOTDIR=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOTDIR/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOTDIR
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_pipeline.py -m gpu -q --timeout 600 -p no:cacheprovider -k "fpn or featurenet or end_to_end or conv2d or files or prefetcher" > $OUT/pytest_sel.log 2>&1
echo "pytest exit: $?" >> $OUT/pytest_sel.log
tail -8 $OUT/pytest_sel.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --streams 1 --no-batch1 > $OUT/bench.json 2> $OUT/bench.err
tail -3 $OUT/bench.err
python tools/show_bench.py $OUT/bench.json | grep -E "ms/step|roofline_feature|^feature"
timeout 600 python tools/gpu_files_throughput.py 49 1 4 16 32 64 > $OUT/files_throughput.txt 2>&1
cat $OUT/files_throughput.txt | grep -v amdgpu.ids
