#!/bin/bash
# the 56 KiB LDS floor of the split-f16 kernels (at most two workgroups per CU) against none: per-kernel times of the step, and the depth checksum
cd /tmp
for v in hip ab3 hip ab3; do
  echo "== library $v"
  LIB=$GRAFT_REPO_ROOT/casmvsnet_pl_amd/libcasmvs_$v.so
  CASMVS_LIB_PATH=$LIB timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -o stats -- python $GRAFT_REPO_ROOT/tools/notorch/step_runner.py --batch 8 --steps 10 --warmup 3 --lib $LIB 2>&1 | grep -E "checksum|^step"
  f=$(find /tmp/prof_$v -name "*kernel_stats.csv" | head -1)
  grep -E "conv_s2_sf_kernel<8, 16|conv2d_ci_sf_kernel|deconv11_sf|conv0_zm_kernel" $f | awk -F'","' '{gsub(/"/,"",$1); printf "   %-70s calls %s avg_us %.1f\n", substr($1,1,70), $2, $4/1000}'
  rm -rf /tmp/prof_$v
done
