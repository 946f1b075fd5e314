#!/bin/bash
# Round 3, call b: z-walk head with two staging register sets (A/B vs one set), the fp64-truth gradient test.
TAG=${1:-r3b}
ROOTDIR=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOTDIR/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOTDIR
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_training.py -m gpu -q --timeout 600 -p no:cacheprovider -k "prob_head or train_mode_matches or costreg or end_to_end" > $OUT/pytest_sel.log 2>&1
echo "pytest exit: $?" >> $OUT/pytest_sel.log
tail -6 $OUT/pytest_sel.log
for B in 2 1; do
LAYER_PROBE_ITEMS=prob timeout 200 python tools/gpu_layer_probe.py 512 640 $B > $OUT/probe_b${B}_pf2.txt 2>&1
CASMVS_LIB_PATH=$ROOTDIR/casmvsnet_pl_amd/libcasmvs_pf1.so LAYER_PROBE_ITEMS=prob timeout 200 python tools/gpu_layer_probe.py 512 640 $B > $OUT/probe_b${B}_pf1.txt 2>&1
cat $OUT/probe_b${B}_pf2.txt $OUT/probe_b${B}_pf1.txt
done
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --streams 1 --no-batch1 > $OUT/bench_pf2.json 2> $OUT/bench.err
CASMVS_LIB_PATH=$ROOTDIR/casmvsnet_pl_amd/libcasmvs_pf1.so timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --streams 1 --no-batch1 > $OUT/bench_pf1.json 2>> $OUT/bench.err
tail -3 $OUT/bench.err
for f in pf2 pf1; do python - <<PY
import json
j = json.load(open("$OUT/bench_$f.json")); s = j["stage_ms_per_step"]
print("$f: %.1f maps/s  prob %.4f %.4f %.4f  costreg frac %.3f" % (j["value"], s["costreg_2/prob"], s["costreg_1/prob"], s["costreg_0/prob"], j["roofline_costreg"]["frac"]))
PY
done
cp gpurun_out/parity_report.json $OUT/ 2>/dev/null
