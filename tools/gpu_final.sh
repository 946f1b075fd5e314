#!/bin/bash
# Evidence run for profiles/: parity tests, bench line (batch 2 and batch 1), rocprofv3 kernel stats,
# and two SEPARATE PMC passes (FETCH_SIZE, WRITE_SIZE) for the HBM traffic of the hot kernels.
TAG=${1:-final}
ROOTDIR=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOTDIR/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOTDIR
export TMPDIR=/tmp
bash tools/gpu_round.sh $TAG
timeout 300 python bench.py --batch 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_batch1.json 2>> $OUT/bench.err
BENCH="python $ROOTDIR/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-events"
run_pmc () { name=$1; shift
  (cd /tmp && timeout 150 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o pmc -- $BENCH > $OUT/$name.log 2>&1)
  find $OUT/$name -type f -size +8M -delete 2>/dev/null
}
run_pmc pmc_fetch FETCH_SIZE
run_pmc pmc_write WRITE_SIZE
ls -R $OUT | head -30
