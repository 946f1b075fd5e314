#!/bin/bash
# A/B of two builds of the library on the same box: bench lines (single stream, batch 2, no CPU baseline), alternating.
#   tools/gpu_ab_lib.sh <tag> <libA.so> <libB.so> [reps]
TAG=${1:-ablib}; A=$2; B=$3; REPS=${4:-2}
ROOTDIR=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOTDIR/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOTDIR
for r in $(seq 1 $REPS); do for L in $A $B; do
  n=$(basename $L .so)
  CASMVS_LIB_PATH=$ROOTDIR/$L timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --streams 1 --no-batch1 > $OUT/${n}_$r.json 2>> $OUT/err.txt
  python - <<PY
import json
j = json.load(open("$OUT/${n}_$r.json"))
s = j["stage_ms_per_step"]
print("$n rep $r: %.1f maps/s  conv0 %.3f %.3f %.3f  costreg frac %.3f  feature %.3f ms  costvol %.3f" % (
    j["value"], s["costreg_2/conv0"], s["costreg_1/conv0"], s["costreg_0/conv0"], j["roofline_costreg"]["frac"], s["feature"],
    s["costvol_2"] + s["costvol_1"] + s["costvol_0"]))
PY
done; done
