#!/bin/bash
# One GPU round on the gpurun box: parity tests, bench line, rocprofv3 kernel stats.
# usage: tools/gpu_round.sh [tag]     (outputs under gpurun_out/<tag>/)
TAG=${1:-r1}
ROOTDIR=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOTDIR/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOTDIR
export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -E "Marketing Name|Compute Unit|gfx" | head -8 > $OUT/rocminfo.txt
nproc > $OUT/nproc.txt; lscpu | grep -E "Model name|^CPU\(s\)" >> $OUT/nproc.txt
python -c "
from casmvsnet_pl_amd import ops
names = {0: '4x4x1_16b', 1: '16x16x4', 2: '32x32x2', 3: '16x16x1_4b'}
for shape in (1,):
    for blocks in (2048,):
        print('mfma', names[shape], 'blocks', blocks, 'TFLOP/s %.1f' % ops.selftest_mfma_rate(shape, blocks, 4096))
" > $OUT/mfma_rate.txt 2>&1
timeout 1200 python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest exit: $?" >> $OUT/pytest_gpu.log
cp gpurun_out/parity_report.json $OUT/ 2>/dev/null
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
echo "bench exit: $?" >> $OUT/bench.err
if [ "$2" != "noprof" ]; then
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o stats -- python $ROOTDIR/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-events > $OUT/prof_bench.json 2> $OUT/prof.err)
  echo "rocprof exit: $?" >> $OUT/prof.err
  # keep only the small csv summaries
  find $OUT/prof -name "*.db" -delete 2>/dev/null
  find $OUT/prof -type f -size +4M -delete 2>/dev/null
fi
cat $OUT/mfma_rate.txt; tail -5 $OUT/pytest_gpu.log; cat $OUT/bench.json | cut -c1-600; tail -3 $OUT/bench.err
