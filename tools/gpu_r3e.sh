#!/bin/bash
# Round 3, call e: fused FPN tail (tests + bench A/B), ablations of the z-walk head.
TAG=${1:-r3e}
ROOTDIR=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOTDIR/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOTDIR
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -p no:cacheprovider -k "fpn or featurenet or end_to_end or conv2d" > $OUT/pytest_sel.log 2>&1
echo "pytest exit: $?" >> $OUT/pytest_sel.log
tail -8 $OUT/pytest_sel.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --streams 1 --no-batch1 > $OUT/bench.json 2> $OUT/bench.err
tail -3 $OUT/bench.err
python tools/show_bench.py $OUT/bench.json | grep -E "ms/step|roofline_feature|^feature"
bash tools/gpu_pz_ablate.sh $TAG
