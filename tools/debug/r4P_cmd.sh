#!/bin/bash
# round 4, call P: first GPU run of conv2d_k5s2_sf_kernel (FeatureNet conv1.0 / conv2.0 on the f16 cores), then the step with and without it
timeout 200 tools/probes/bin/conv2d_k5s2_check 8; echo "-- exit $?"
for args in "" "--f32-k5s2" "" "--f32-k5s2"; do
  echo "== step_runner $args"
  timeout 90 python tools/notorch/step_runner.py --batch 8 $args 2>&1 | grep "^step\|checksum\|stages"
done
