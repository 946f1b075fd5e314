#!/bin/bash
# round 4, call D: conv11 + prob fused, producer / consumer wave groups (default) against all waves in step (libcasmvs_zfsync.so)
for v in hip zfsync; do
  echo "== $v"
  LD_PRELOAD=$PWD/casmvsnet_pl_amd/libcasmvs_$v.so timeout 200 tools/probes/bin/conv11_prob_check 8 | grep "B=8\|ALL\|FAIL\|B=1 in 2x\|B=2"
done
echo "== batch 1, 2"
timeout 100 tools/probes/bin/conv11_prob_check 1 | grep "B=1 in [0-9]*x[0-9][0-9]*x[0-9][0-9][0-9]\|B=1 in 24\|FAIL"
timeout 100 tools/probes/bin/conv11_prob_check 2 | grep "B=2 in [0-9]*x[0-9][0-9]*x[0-9][0-9][0-9]\|B=2 in 24\|FAIL"
for args in "" "--lib casmvsnet_pl_amd/libcasmvs_nozf.so" ""; do
  echo "== step_runner $args"
  timeout 90 python tools/notorch/step_runner.py --batch 8 $args 2>&1 | grep "^step\|checksum\|stages"
done
