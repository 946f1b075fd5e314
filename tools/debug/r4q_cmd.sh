for v in hip ws7 ws7n4; do
  echo "== $v"
  LD_PRELOAD=$PWD/casmvsnet_pl_amd/libcasmvs_$v.so timeout 120 tools/probes/bin/conv0_zm_check 8 | grep "^B=8\|^B=1 cin=\(16\|8\) [0-9]*x[0-9][0-9][0-9]"
done
echo "== whole step"
timeout 200 python tools/notorch/ab_step.py --rounds 3 casmvsnet_pl_amd/libcasmvs_hip.so casmvsnet_pl_amd/libcasmvs_ws7.so | grep -v "feature\|costvol\|hypoth"
