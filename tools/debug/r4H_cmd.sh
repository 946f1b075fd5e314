#!/bin/bash
# round 4, call H: fused tail with the alternating fifth unit, packed epilogue arithmetic and unconditional slot stores (default) against the previous build
for v in hip zfold hip zfold; do
  echo "== $v"
  LD_PRELOAD=$PWD/casmvsnet_pl_amd/libcasmvs_$v.so timeout 200 tools/probes/bin/conv11_prob_check 8 | grep "B=8\|FAIL"
done
for args in "" "--lib casmvsnet_pl_amd/libcasmvs_zfold.so" ""; do
  echo "== step_runner $args"
  timeout 90 python tools/notorch/step_runner.py --batch 8 $args 2>&1 | grep "^step\|checksum"
done
