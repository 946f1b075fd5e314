#!/bin/bash
# Round 4, work-order A/B of the persistent kernels on the whole step (no torch: ~30 s of GPU time for three builds x 3 runs).
# HERE (no GPU needed), before the call:
#   python tools/build_variant.py ord1 -DCASMVS_DB_ORDER=1 -DCASMVS_CI_ORDER=1
#   python tools/build_variant.py ord2 -DCASMVS_DB_ORDER=2 -DCASMVS_CI_ORDER=2 -DCASMVS_C2_PAIR=1 -DCASMVS_FS_PAIR=1
#   (add casmvsnet_pl_amd/build_ord1/ and build_ord2/ to .gpurunignore: only the .so files travel)
# THEN:  /usr/local/graft/bin/gpurun --timeout 120 -- 'bash tools/gpu_r4_order_ab.sh'
# Reading the result: ab_step.py prints every stage's delta against the production build; a kernel family whose stage gets faster
# takes that order as its default (per shape if the levels disagree, as conv0_splitf16.hip does), the macro stays for the next A/B.
ROOTDIR=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOTDIR
mkdir -p gpurun_out
P=casmvsnet_pl_amd
python tools/notorch/ab_step.py --rounds 3 $P/libcasmvs_hip.so $P/libcasmvs_ord1.so $P/libcasmvs_ord2.so > gpurun_out/order_ab.txt 2>&1
cat gpurun_out/order_ab.txt
