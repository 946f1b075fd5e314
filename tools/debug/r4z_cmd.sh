#!/bin/bash
# round 4, call z: FeatureNet over groups of images (cache blocking) in the torch-free runner
for k in 1 2 3 4 6 8 12 24 1; do
  echo "== feature-split $k"
  timeout 120 python tools/notorch/step_runner.py --batch 8 --feature-split $k 2>&1 | grep "^step\|stages\|rror\|checksum"
done
