#!/bin/bash
# 16 planes per workgroup + dense CS = 16 box layout: parity, same-box A/B against the round-4 build (probe with dirtied caches, bench)
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "costvol or homo_warp or full_size or benched" 2>&1 | tail -3
for L in cvold hip; do echo "== lib $L batch 8"; CASMVS_LIB_PATH=$GRAFT_REPO_ROOT/casmvsnet_pl_amd/libcasmvs_$L.so CV_PROBE_DIRTY=512 CV_PROBE_IMPLS=lds CV_PROBE_REPS=6 timeout 200 python tools/gpu_costvol_probe.py 512 640 3 8 2>&1 | grep -E "depth=|homo_warp \(un-fused op\) lds |bitwise"; done
for L in cvold hip; do echo "== lib $L batch 1"; CASMVS_LIB_PATH=$GRAFT_REPO_ROOT/casmvsnet_pl_amd/libcasmvs_$L.so CV_PROBE_DIRTY=512 CV_PROBE_IMPLS=lds CV_PROBE_REPS=6 timeout 200 python tools/gpu_costvol_probe.py 512 640 3 1 2>&1 | grep -E "depth=|homo_warp \(un-fused op\) lds "; done
bash tools/gpu_ab_lib.sh r5h casmvsnet_pl_amd/libcasmvs_cvold.so casmvsnet_pl_amd/libcasmvs_hip.so 2
