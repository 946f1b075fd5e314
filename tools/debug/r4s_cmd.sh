for v in hip mix4; do
  echo "== $v"
  LD_PRELOAD=$PWD/casmvsnet_pl_amd/libcasmvs_$v.so timeout 120 tools/probes/bin/conv0_zm_check 8 | grep "^B=8\|ALL\|FAIL\|cin=32 6x20"
  LD_PRELOAD=$PWD/casmvsnet_pl_amd/libcasmvs_$v.so timeout 60 tools/probes/bin/deconv9_check 8 | tail -2
  LD_PRELOAD=$PWD/casmvsnet_pl_amd/libcasmvs_$v.so timeout 60 tools/probes/bin/deconv11_check 8 | tail -2
done
echo "== whole step (checksums must be equal: the same roundings)"
timeout 60 python tools/notorch/step_runner.py --batch 8 | grep "checksum"
timeout 60 python tools/notorch/step_runner.py --batch 8 --lib casmvsnet_pl_amd/libcasmvs_mix4.so | grep "checksum"
timeout 200 python tools/notorch/ab_step.py --rounds 3 casmvsnet_pl_amd/libcasmvs_hip.so casmvsnet_pl_amd/libcasmvs_mix4.so
