#!/bin/bash
# round 4, call w: FeatureNet without the (N,C,h,w) stores of levels 0 / 1 (A/B in the torch-free runner), then the new GPU tests
for args in "" "--nchw-feats" "" "--nchw-feats"; do
  echo "== step_runner $args"
  timeout 90 python tools/notorch/step_runner.py --batch 8 $args 2>&1 | grep "^step\|checksum\|stages"
done
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "conv_s2 or pixel_major or bit_stable or float32_layers_equal or featurenet_matches or benched_launch" 2>&1 | tail -5
