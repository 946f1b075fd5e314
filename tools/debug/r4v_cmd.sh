#!/bin/bash
# round 4, call v: conv_s2_sf_kernel with units in flight per workgroup = (conv1, conv3): default (1, 2) against (2, 3)
for v in hip s2d23; do
  echo "== $v"
  LD_PRELOAD=$PWD/casmvsnet_pl_amd/libcasmvs_$v.so timeout 120 tools/probes/bin/conv_s2_check 8 | grep "B=8\|ALL\|FAIL"
done
echo "== batch 1"
for v in hip s2d23; do
  LD_PRELOAD=$PWD/casmvsnet_pl_amd/libcasmvs_$v.so timeout 120 tools/probes/bin/conv_s2_check 1 | grep "B=1 in [1-9][0-9]*x[0-9][0-9][0-9]\|B=1 in [0-9]*x64x80\|FAIL"
done
for args in "" "--f32-layers conv1,conv3" "--lib casmvsnet_pl_amd/libcasmvs_s2d23.so" ""; do
  echo "== step_runner $args"
  timeout 90 python tools/notorch/step_runner.py --batch 8 $args 2>&1 | grep "^step\|checksum"
done
