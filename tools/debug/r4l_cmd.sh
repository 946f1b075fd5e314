#!/bin/bash
# group-wise correlation backward as one launch: parity tests, then the training step of the gwc8 config (and of the default config again)
timeout 600 python -m pytest tests/test_gpu_training.py tests/test_gpu_autograd.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -4
timeout 300 python bench.py --mode train --config dtu_640x512_v3_gwc8 --steps 20 --warmup 5 > $OUT/bench_train_gwc8.json 2> $OUT/bench_train_gwc8.err; grep -o '"train_step_ms": [0-9.]*' $OUT/bench_train_gwc8.json; tail -2 $OUT/bench_train_gwc8.err
timeout 300 python bench.py --mode train --steps 20 --warmup 5 > $OUT/bench_train.json 2> $OUT/bench_train.err; grep -o '"train_step_ms": [0-9.]*' $OUT/bench_train.json
