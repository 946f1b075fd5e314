F='^B=1 cin=\(8\|16\) [0-9]*x[0-9]*x[0-9][0-9] \|^B=2 cin=16 9x'
echo "== production: cin 32 = warp-specialised 16 x 32"
timeout 120 tools/probes/bin/conv0_zm_check 8 | grep -v "$F"
echo "== ws4w: cin 32 = warp-specialised 8 x 64"
LD_PRELOAD=$PWD/casmvsnet_pl_amd/libcasmvs_ws4w.so timeout 120 tools/probes/bin/conv0_zm_check 8 | grep "cin=32"
echo "== whole step"
timeout 200 python tools/notorch/ab_step.py --rounds 3 casmvsnet_pl_amd/libcasmvs_hip.so casmvsnet_pl_amd/libcasmvs_ws4w.so
echo "== pytest (bit stability of every f16 instantiation, z-march parity, benched launch vs oracle, full-size configs)"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "bit_stable or zmarch or benched_launch or full_size_config or non_finite or float32_layers_equal" 2>&1 | tail -6
