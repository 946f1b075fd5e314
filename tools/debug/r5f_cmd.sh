#!/bin/bash
# zero-padded boxes in the LDS plane sweep: parity (three families bit-equal, oracle), then the same-box A/B against the previous build
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "costvol or homo_warp or full_size or benched" 2>&1 | tail -4
for L in cvold hip cvold hip; do
  for B in 8 1; do
    echo "== lib $L batch $B"
    CASMVS_LIB_PATH=$GRAFT_REPO_ROOT/casmvsnet_pl_amd/libcasmvs_$L.so CV_PROBE_DIRTY=512 CV_PROBE_IMPLS=lds CV_PROBE_REPS=6 timeout 200 python tools/gpu_costvol_probe.py 512 640 3 $B 2>&1 | grep -v "^$" | tail -8
  done
done
