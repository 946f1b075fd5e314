export CV_PROBE_DIRTY=512 CV_PROBE_REPS=6
echo "== V5 production lib (rolled LDS vs gather)"; CV_PROBE_IMPLS=gather,lds timeout 250 python tools/gpu_costvol_probe.py 864 1152 5 4 2>&1 | grep -v Warning | tail -8
echo "== V5 unrolled NV=4, CS default"; CASMVS_LIB_PATH=casmvsnet_pl_amd/libcasmvs_cvu.so CV_PROBE_IMPLS=lds timeout 250 python tools/gpu_costvol_probe.py 864 1152 5 4 2>&1 | grep -v Warning | tail -7
echo "== V5 unrolled NV=4, CS=8"; CASMVS_CV_CS=8 CASMVS_LIB_PATH=casmvsnet_pl_amd/libcasmvs_cvu.so CV_PROBE_IMPLS=lds timeout 250 python tools/gpu_costvol_probe.py 864 1152 5 4 2>&1 | grep -v Warning | tail -7
echo "== V7 production lib"; CV_PROBE_IMPLS=gather,lds timeout 250 python tools/gpu_costvol_probe.py 576 768 7 8 2>&1 | grep -v Warning | tail -8
echo "== V7 unrolled NV=6, CS=8"; CASMVS_CV_CS=8 CASMVS_LIB_PATH=casmvsnet_pl_amd/libcasmvs_cvu.so CV_PROBE_IMPLS=lds timeout 250 python tools/gpu_costvol_probe.py 576 768 7 8 2>&1 | grep -v Warning | tail -7
echo "== FPN tail: early prefetch (production)"; timeout 200 python tools/gpu_fpn_probe.py 512 640 24 2>&1 | grep -v Warning | tail -4
echo "== FPN tail: round-5 order"; CASMVS_LIB_PATH=casmvsnet_pl_amd/libcasmvs_fsold.so timeout 200 python tools/gpu_fpn_probe.py 512 640 24 2>&1 | grep -v Warning | tail -4
echo "== FPN tail again: production"; timeout 200 python tools/gpu_fpn_probe.py 512 640 24 2>&1 | grep -v Warning | tail -4
