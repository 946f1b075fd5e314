#!/bin/bash
# round 4, call y: batch size of the benched launch (torch-free runner), and the conv_s2 kernels capped at two workgroups per CU
for b in 8 12 16 24 32 8; do
  echo "== batch $b"
  timeout 120 python tools/notorch/step_runner.py --batch $b 2>&1 | grep "^step\|stages\|rror"
done
timeout 120 tools/probes/bin/conv_s2_check 8 | grep "B=8\|ALL\|FAIL"
