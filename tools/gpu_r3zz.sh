#!/bin/bash
# Round 3, call z: 2D CI kernel capped at 2 workgroups per CU: whole suite, bench
TAG=${1:-r3zz}
ROOTDIR=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOTDIR/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOTDIR
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest exit: $?" >> $OUT/pytest_gpu.log
tail -6 $OUT/pytest_gpu.log
cp gpurun_out/parity_report.json $OUT/ 2>/dev/null
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
tail -3 $OUT/bench.err
python tools/show_bench.py $OUT/bench.json | head -14
python - <<PY
import json
j = json.load(open("$OUT/bench.json"))
print("two_streams_float32", j.get("two_streams_float32"))
print("conv0_other_modes", j.get("conv0_other_modes"))
PY
