#!/bin/bash
# persistent workgroups in the plane sweep: parity, then A/B persistent vs one workgroup per item (trace build, CASMVS_CV_PERSIST), then the round-4 build
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "costvol or homo_warp or plane_sweep or full_size or benched" 2>&1 | tail -3
for P in 0 1 0 1; do for B in 8 1; do echo "== trace build, CASMVS_CV_PERSIST=$P batch $B"; CASMVS_CV_PERSIST=$P CASMVS_LIB_PATH=$GRAFT_REPO_ROOT/casmvsnet_pl_amd/libcasmvs_trace.so CV_PROBE_DIRTY=512 CV_PROBE_IMPLS=lds CV_PROBE_REPS=6 timeout 200 python tools/gpu_costvol_probe.py 512 640 3 $B 2>&1 | grep -E "depth=smooth|homo_warp \(un-fused op\) lds |bitwise: False"; done; done
bash tools/gpu_ab_lib.sh r5o casmvsnet_pl_amd/libcasmvs_cvold.so casmvsnet_pl_amd/libcasmvs_hip.so 2
