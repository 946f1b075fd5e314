#!/bin/bash
# A/B of the plane sweep's division (reciprocal + Newton step vs correctly rounded, -DCASMVS_IEEE_DIV): depth-index
# flips of the full-size parity tests with each build.
TAG=${1:-abdiv}
ROOTDIR=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOTDIR/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOTDIR
for v in hip ieeediv; do
  CASMVS_LIB_PATH=$ROOTDIR/casmvsnet_pl_amd/libcasmvs_$v.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "full_size" > $OUT/pytest_$v.log 2>&1
  cp gpurun_out/parity_report.json $OUT/parity_$v.json
  tail -2 $OUT/pytest_$v.log
done
python - <<PY
import json
for v in ("hip", "ieeediv"):
    r = [e for e in json.load(open("$OUT/parity_%s.json" % v)) if e["name"] == "e2e_full_size"]
    for e in r:
        print(v, e["config"], "flips", [e["index_flips_%d" % l] for l in (2, 1, 0)], "max boundary dist",
              ["%.1e" % e["flip_max_boundary_dist_%d" % l] for l in (2, 1, 0)], "depth_rel_0 %.1e" % e["depth_rel_0"])
PY
