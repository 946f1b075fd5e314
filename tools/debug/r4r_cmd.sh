for v in hip zwpk; do
  echo "== $v"
  LD_PRELOAD=$PWD/casmvsnet_pl_amd/libcasmvs_$v.so timeout 120 tools/probes/bin/conv0_zm_check 8 | grep "^B=8\|ALL\|FAIL"
done
echo "== trace"
LD_PRELOAD=$PWD/casmvsnet_pl_amd/libcasmvs_zwtrace.so tools/probes/bin/zw_trace 8 | grep -v "100 MHz"
timeout 60 python tools/notorch/step_runner.py --batch 8 | tail -3
