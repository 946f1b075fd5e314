#!/bin/bash
# round 4, call C: the step with conv11 + prob fused (default) against the two kernels (libcasmvs_nozf.so)
for args in "" "--lib casmvsnet_pl_amd/libcasmvs_nozf.so" "" "--lib casmvsnet_pl_amd/libcasmvs_nozf.so"; do
  echo "== step_runner $args"
  timeout 90 python tools/notorch/step_runner.py --batch 8 $args 2>&1 | grep "^step\|checksum\|stages"
done
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bit_stable or float32_layers_equal or benched_launch or full_size" 2>&1 | tail -5
