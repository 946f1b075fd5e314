for v in hip sk2 sk6 sk16; do
  echo "== $v"
  LD_PRELOAD=$PWD/casmvsnet_pl_amd/libcasmvs_$v.so timeout 120 tools/probes/bin/conv0_zm_check 8 | grep "^B=8"
done
echo "== whole step"
timeout 300 python tools/notorch/ab_step.py --rounds 3 casmvsnet_pl_amd/libcasmvs_hip.so casmvsnet_pl_amd/libcasmvs_sk2.so casmvsnet_pl_amd/libcasmvs_sk6.so casmvsnet_pl_amd/libcasmvs_sk16.so | grep -v "feature\|costvol\|hypoth"
