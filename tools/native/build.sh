#!/bin/bash
# Builds the torch-free check binaries against the in-tree library (run after `python -m casmvsnet_pl_amd.build`).
# tools/probes/bin/ is git-ignored but travels with the gpurun snapshot:
#   /usr/local/graft/bin/gpurun --timeout 60 -- 'tools/probes/bin/prob_wgrad_check; tools/probes/bin/fusion_check'
set -e
cd "$(dirname "$0")/../.."
mkdir -p tools/probes/bin
/opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 tools/native/conv0_ab.cpp -Iinclude -ldl -o tools/probes/bin/conv0_ab 2>&1 | grep -E "error" || true
ls -la tools/probes/bin/conv0_ab
for name in fusion_check prob_wgrad_check conv0_zm_check deconv11_check deconv9_check conv_s2_check conv11_prob_check conv2d_k5s2_check; do
  /opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 tools/native/$name.cpp -Iinclude -Lcasmvsnet_pl_amd -lcasmvs_hip \
    -Wl,-rpath,'$ORIGIN/../../../casmvsnet_pl_amd' -o tools/probes/bin/$name 2>&1 | grep -E "error" || true
  ls -la tools/probes/bin/$name
done
# the co-residency programs (gpu_run.sh stage `coresidency`): the library's float32 kernels beside synthetic neighbours, and the stand-alone reproducer
/opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 tools/native/coresidency_lib_victim.cpp -Iinclude -Lcasmvsnet_pl_amd -lcasmvs_hip \
  -Wl,-rpath,'$ORIGIN/../../../casmvsnet_pl_amd' -o tools/probes/bin/coresidency_lib_victim 2>&1 | grep -E "error" || true
/opt/rocm/bin/hipcc -O2 -std=c++17 --offload-arch=gfx950 tools/probes/pk_fma_opsel_repro.hip -o tools/probes/bin/pk_fma_opsel_repro 2>&1 | grep -E "error" || true
ls -la tools/probes/bin/coresidency_lib_victim tools/probes/bin/pk_fma_opsel_repro
