F='^B=1 cin=\(8\|16\|32\) [0-9]*x[0-9]*x[0-9][0-9] \|^B=2 cin=16 9x'
echo "== ws0: z-march, two-phase workgroups (cin 8 / 16: 8 x 64 patches, cin 32: 16 x 32)"
LD_PRELOAD=$PWD/casmvsnet_pl_amd/libcasmvs_ws0.so timeout 120 tools/probes/bin/conv0_zm_check 8 | grep -v "$F"
echo "== ws7: z-march, warp-specialised workgroups for every cin"
LD_PRELOAD=$PWD/casmvsnet_pl_amd/libcasmvs_ws7.so timeout 120 tools/probes/bin/conv0_zm_check 8 | grep -v "$F"
echo "== whole step A/B: production (cin 32 tiled) | ws0 (cin 32 z-march two-phase) | ws4 (cin 32 warp-specialised) | ws7 (all warp-specialised)"
timeout 300 python tools/notorch/ab_step.py --rounds 3 casmvsnet_pl_amd/libcasmvs_hip.so casmvsnet_pl_amd/libcasmvs_ws0.so casmvsnet_pl_amd/libcasmvs_ws4.so casmvsnet_pl_amd/libcasmvs_ws7.so
