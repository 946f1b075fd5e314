#!/bin/bash
# round 4, call F: fused tail after the rebalancing, new GPU tests
timeout 200 tools/probes/bin/conv11_prob_check 8 | grep "B=8\|ALL\|FAIL"
for args in "" "--lib casmvsnet_pl_amd/libcasmvs_nozf.so" ""; do
  echo "== step_runner $args"
  timeout 90 python tools/notorch/step_runner.py --batch 8 $args 2>&1 | grep "^step\|checksum"
done
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "zfused or bit_stable or absolute_error" 2>&1 | tail -5
