#!/bin/bash
# round 4, call M: the fused tail with `prob`'s weights from an LDS table (default) against scalar loads (libcasmvs_zfsw.so)
for v in hip zfsw hip zfsw; do
  echo "== $v"
  LD_PRELOAD=$PWD/casmvsnet_pl_amd/libcasmvs_$v.so timeout 200 tools/probes/bin/conv11_prob_check 8 | grep "B=8\|FAIL"
done
for args in "" "--lib casmvsnet_pl_amd/libcasmvs_zfsw.so" ""; do
  echo "== step_runner $args"
  timeout 90 python tools/notorch/step_runner.py --batch 8 $args 2>&1 | grep "^step\|checksum"
done
