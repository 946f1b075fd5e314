#!/bin/bash
# A/B: fpn_tail0_sf_kernel with the next chunk's loads issued before the split (default build) vs behind the second barrier (libcasmvs_fslate.so)
for rep in 1 2 3; do
  for lib in libcasmvs_hip.so libcasmvs_fslate.so; do
    echo "== $lib"
    python tools/notorch/step_runner.py --batch 8 --steps 20 --warmup 5 --lib casmvsnet_pl_amd/$lib | grep -E "^step|stages" | cut -c1-140
  done
done
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "fpn or feature or bit_stable" 2>&1 | tail -3
