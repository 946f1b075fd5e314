#!/bin/bash
# PMC traffic passes (FETCH_SIZE, WRITE_SIZE: one pass each, --kernel-trace only beside them) and a kernel-stats pass over the
# torch-free step runner: three rocprofv3 runs of a process that starts in a second - ~1.5 min of GPU time instead of the ~10 the same
# passes over bench.py cost in round 3.  Output in the layout tools/summarize_profile.py reads:
#   /usr/local/graft/bin/gpurun --timeout 240 -- 'bash tools/gpu_pmc_notorch.sh r4pmc'
#   PMC_BATCH=8 PMC_DATE=... python tools/summarize_profile.py gpurun_out/r4pmc r04      (HERE, on the merged gpurun_out)
# so that the bench line of the same build reports roofline.traffic with traffic_source.same_library = true.
TAG=${1:-pmc_notorch}
ROOTDIR=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOTDIR/gpurun_out/$TAG
BATCH=${PMC_BATCH:-8}
mkdir -p $OUT
export TMPDIR=/tmp
RUN="python $ROOTDIR/tools/notorch/step_runner.py --batch $BATCH --steps 3 --warmup 1"
run_pmc () { name=$1; shift
  (cd /tmp && timeout 120 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o pmc -- $RUN > $OUT/$name.log 2>&1)
  find $OUT/$name -type f -size +8M -delete 2>/dev/null
}
run_pmc pmc_fetch FETCH_SIZE
run_pmc pmc_write WRITE_SIZE
(cd /tmp && timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o stats -- $RUN > $OUT/prof.log 2>&1)
find $OUT/prof -name "*.db" -delete 2>/dev/null; find $OUT/prof -type f -size +4M -delete 2>/dev/null
date -u +%Y-%m-%dT%H:%MZ > $OUT/collected.txt
echo "PMC_CMD_NOTE=\"$RUN\" PMC_BATCH=$BATCH PMC_DATE=$(cat $OUT/collected.txt)" > $OUT/summarize_env.txt
ls -la $OUT/pmc_fetch $OUT/pmc_write $OUT/prof | head -30; tail -2 $OUT/pmc_fetch.log
