echo "== production (16 x 32 patches)"
timeout 120 tools/probes/bin/conv0_zm_check 8 | grep -v "^B=1 cin=\(8\|16\|32\) [0-9]*x[0-9]*x[0-9][0-9] \|^B=2 cin=16 9x"
echo "== variant (8 x 64 patches)"
LD_PRELOAD=$PWD/casmvsnet_pl_amd/libcasmvs_zmwide.so timeout 120 tools/probes/bin/conv0_zm_check 8 | grep -v "^B=1 cin=\(8\|16\|32\) [0-9]*x[0-9]*x[0-9][0-9] \|^B=2 cin=16 9x"
echo "== whole step A/B"
timeout 200 python tools/notorch/ab_step.py --rounds 3 casmvsnet_pl_amd/libcasmvs_hip.so casmvsnet_pl_amd/libcasmvs_zmwide.so
