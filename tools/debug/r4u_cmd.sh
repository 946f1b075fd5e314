#!/bin/bash
# round 4, call u: first GPU run of conv_s2_sf_kernel (conv1 / conv3 on the f16 cores): native check, then the step with and without it
timeout 120 tools/probes/bin/conv_s2_check 8; echo "-- conv_s2_check: exit $?"
for args in "" "--f32-layers conv1,conv3" "--f32-layers conv3" "--f32-layers conv1" ""; do
  echo "== step_runner $args"
  timeout 90 python tools/notorch/step_runner.py --batch 8 $args 2>&1 | tail -4
done
