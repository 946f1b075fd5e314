for v in hip ns3 ns4; do
  echo "== $v"
  LD_PRELOAD=$PWD/casmvsnet_pl_amd/libcasmvs_$v.so timeout 120 tools/probes/bin/conv0_zm_check 8 | grep "cin=32"
done
echo "== trace (NSET 4)"
LD_PRELOAD=$PWD/casmvsnet_pl_amd/libcasmvs_zwtrace.so tools/probes/bin/zw_trace 8 | grep -v "100 MHz"
