for v in hip ns3 ns4; do
  echo "== $v"
  LD_PRELOAD=$PWD/casmvsnet_pl_amd/libcasmvs_$v.so timeout 120 tools/probes/bin/conv0_zm_check 8 | grep "^B=8 cin=32\|cin=32 6x20"
done
