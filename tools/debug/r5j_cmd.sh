#!/bin/bash
# final plane sweep (zero-padded boxes, 16 planes per workgroup, guarded stores): same-box A/B against the round-4 build
for L in cvold hip; do for B in 8 1; do echo "== lib $L batch $B"; CASMVS_LIB_PATH=$GRAFT_REPO_ROOT/casmvsnet_pl_amd/libcasmvs_$L.so CV_PROBE_DIRTY=512 CV_PROBE_IMPLS=lds CV_PROBE_REPS=6 timeout 200 python tools/gpu_costvol_probe.py 512 640 3 $B 2>&1 | grep -E "depth=|homo_warp \(un-fused op\) lds |bitwise"; done; done
bash tools/gpu_ab_lib.sh r5j casmvsnet_pl_amd/libcasmvs_cvold.so casmvsnet_pl_amd/libcasmvs_hip.so 2
