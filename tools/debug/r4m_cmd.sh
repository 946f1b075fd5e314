#!/bin/bash
# A/B of the gwc8 training step: the previous _GroupwiseVolume.backward (torch ops + one warp backward and one recomputed warped volume per view) vs one launch
cp casmvsnet_pl_amd/training.py /tmp/training_new.py
for which in before after before after; do
  if [ $which = before ]; then cp tools/debug/tmp/training_before_gwc_fused.py casmvsnet_pl_amd/training.py   # (git show <commit before>:casmvsnet_pl_amd/training.py > tools/debug/tmp/...); else cp /tmp/training_new.py casmvsnet_pl_amd/training.py; fi
  timeout 300 python bench.py --mode train --config dtu_640x512_v3_gwc8 --steps 20 --warmup 5 > $OUT/bench_train_gwc8_$which.json 2> $OUT/err.txt
  echo "$which: $(grep -o '"train_step_ms": [0-9.]*' $OUT/bench_train_gwc8_$which.json) $(grep -o '"peak_memory_gib": [0-9.]*' $OUT/bench_train_gwc8_$which.json)"
done
cp /tmp/training_new.py casmvsnet_pl_amd/training.py
