#!/bin/bash
# Round 3, call h: whole GPU suite, train-mode bench leg, batch / stream sweep of the headline.
TAG=${1:-r3h}
ROOTDIR=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOTDIR/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOTDIR
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest exit: $?" >> $OUT/pytest_gpu.log
tail -6 $OUT/pytest_gpu.log
cp gpurun_out/parity_report.json $OUT/ 2>/dev/null
timeout 300 python bench.py --mode train --steps 10 --warmup 3 > $OUT/bench_train.json 2> $OUT/bench_train.err
tail -2 $OUT/bench_train.err; cut -c1-600 $OUT/bench_train.json
timeout 300 python bench.py --mode train --steps 10 --warmup 3 --no-graph > $OUT/bench_train_eager.json 2>> $OUT/bench_train.err
cut -c1-300 $OUT/bench_train_eager.json
for cfg in "2 2" "4 2" "3 2" "4 1" "2 3"; do set -- $cfg
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-events --no-batch1 --batch $1 --streams $2 > $OUT/bench_b$1_s$2.json 2>> $OUT/bench.err
python - <<PY
import json
j = json.load(open("$OUT/bench_b$1_s$2.json"))
print("batch $1 x streams $2: %.1f maps/s  ms/step %.3f  single-stream %s" % (j["value"], j["ms_per_step"], j.get("single_stream", {}).get("value")))
PY
done
tail -2 $OUT/bench.err
