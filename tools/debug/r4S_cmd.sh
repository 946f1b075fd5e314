#!/bin/bash
# kernel breakdown of the training step (bench.py --mode train under rocprofv3 --kernel-trace --stats)
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_train -o stats -- python $GRAFT_REPO_ROOT/bench.py --mode train --steps 10 --warmup 3 > $OUT/bench_train_prof.json 2> $OUT/bench_train_prof.err
find $OUT/prof_train -name "*.db" -delete 2>/dev/null; find $OUT/prof_train -type f -size +4M -delete 2>/dev/null
cut -c1-300 $OUT/bench_train_prof.json
head -50 $OUT/prof_train/stats_kernel_stats.csv | cut -c1-260
