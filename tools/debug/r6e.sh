export CV_PROBE_DIRTY=512 CV_PROBE_REPS=6
F="grep -v Warning\|amdgpu.ids"
echo "== V5 (1152x864, batch 4) production lib: gather vs the rolled LDS form (all four source boxes resident)"; CV_PROBE_IMPLS=gather,lds timeout 250 python tools/gpu_costvol_probe.py 864 1152 5 4 2>&1 | grep -v "Warning\|amdgpu.ids\|homo_warp"
echo "== V5 unrolled (compile-time view count NV = 4), CS = 16 at levels 2 / 1"; CASMVS_LIB_PATH=casmvsnet_pl_amd/libcasmvs_cvu.so CV_PROBE_IMPLS=lds timeout 250 python tools/gpu_costvol_probe.py 864 1152 5 4 2>&1 | grep -v "Warning\|amdgpu.ids\|homo_warp"
echo "== V5 unrolled NV = 4, CS = 8 everywhere (two workgroups per CU)"; CASMVS_CV_CS=8 CASMVS_LIB_PATH=casmvsnet_pl_amd/libcasmvs_cvu.so CV_PROBE_IMPLS=lds timeout 250 python tools/gpu_costvol_probe.py 864 1152 5 4 2>&1 | grep -v "Warning\|amdgpu.ids\|homo_warp"
echo "== V7 (768x576, batch 8) production lib"; CV_PROBE_IMPLS=gather,lds timeout 250 python tools/gpu_costvol_probe.py 576 768 7 8 2>&1 | grep -v "Warning\|amdgpu.ids\|homo_warp"
echo "== V7 unrolled NV = 6, CS = 8"; CASMVS_CV_CS=8 CASMVS_LIB_PATH=casmvsnet_pl_amd/libcasmvs_cvu.so CV_PROBE_IMPLS=lds timeout 250 python tools/gpu_costvol_probe.py 576 768 7 8 2>&1 | grep -v "Warning\|amdgpu.ids\|homo_warp"
