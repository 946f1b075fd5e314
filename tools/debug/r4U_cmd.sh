#!/bin/bash
# fixed-point LDS image in the variance-volume backward: parity tests, then the training step with its kernel breakdown
timeout 600 python -m pytest tests/test_gpu_training.py tests/test_gpu_autograd.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -5
timeout 300 python bench.py --mode train --steps 20 --warmup 5 > $OUT/bench_train.json 2> $OUT/bench_train.err; grep -o '"train_step_ms": [0-9.]*' $OUT/bench_train.json
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_train -o stats -- python $GRAFT_REPO_ROOT/bench.py --mode train --steps 10 --warmup 3 > $OUT/bench_train_prof.json 2> $OUT/bench_train_prof.err
find $OUT/prof_train -name "*.db" -delete 2>/dev/null; find $OUT/prof_train -type f -size +4M -delete 2>/dev/null
head -12 $OUT/prof_train/stats_kernel_stats.csv | cut -c1-200
grep "costvol_var_bwd" $OUT/prof_train/stats_kernel_stats.csv | cut -c1-260
