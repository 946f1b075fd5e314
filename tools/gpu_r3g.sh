#!/bin/bash
# Round 3, call g: headline A/B of the fused FPN tail (2 streams), files -> depth maps with worker processes.
TAG=${1:-r3g}
ROOTDIR=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOTDIR/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOTDIR
export TMPDIR=/tmp
for r in 1 2; do for f in 1 0; do
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-events --fuse-tail $f > $OUT/bench_tail${f}_$r.json 2>> $OUT/bench.err
python - <<PY
import json
j = json.load(open("$OUT/bench_tail${f}_$r.json"))
print("fuse_tail $f rep $r: %.1f maps/s (2 streams)  single %.1f  batch1 %.1f / concurrent %.1f" % (j["value"], j["single_stream"]["value"], j["batch1"]["value"], j["batch1"]["concurrent"]["value"]))
PY
done; done
tail -2 $OUT/bench.err
timeout 900 python tools/gpu_files_throughput.py 49 16 32 64 128 > $OUT/files_throughput.txt 2>&1
grep -v amdgpu.ids $OUT/files_throughput.txt
