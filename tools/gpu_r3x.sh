#!/bin/bash
# Round 3, call x: batch sweep of the final build (single stream), SQ / LDS / L2 counters of the new f16 kernels.
TAG=${1:-r3x}
ROOTDIR=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOTDIR/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOTDIR
export TMPDIR=/tmp
for b in 1 2 4 8 12 16; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-events --no-batch1 --batch $b > $OUT/bench_b${b}.json 2>> $OUT/bench.err
  python -c "
import json; j=json.load(open('$OUT/bench_b${b}.json')); print('batch $b x 1 stream (one hipGraph replay per step):', round(j['value'],1), 'depth-maps/s,', round(j['ms_per_step'],3), 'ms/step, median', round(j['median_ms_per_step'],3))" | tee -a $OUT/batch_sweep.txt
done
PMC_CMD="python $ROOTDIR/tools/gpu_fpn_probe.py 512 640 24" PMC_FILTER="fpn_tail0" bash tools/gpu_layer_pmc.sh $TAG/pmc_fpn > /dev/null 2>&1
PMC_CMD="python $ROOTDIR/tools/gpu_layer_probe.py" LAYER_PROBE_ITEMS=bottom PMC_FILTER="conv_ci_sf" bash tools/gpu_layer_pmc.sh $TAG/pmc_ci > /dev/null 2>&1
cat $OUT/pmc_fpn/summary.txt | cut -c1-400; cat $OUT/pmc_ci/summary.txt | cut -c1-400
