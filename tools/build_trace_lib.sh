#!/bin/bash
# Builds the -DCASMVS_TRACE profiling copy of the library (cross-compiles without a GPU).
cd "$(dirname "$0")/.." && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -DCASMVS_TRACE \
  -Iinclude -Icasmvsnet_pl_amd/csrc casmvsnet_pl_amd/csrc/*.hip -o casmvsnet_pl_amd/libcasmvs_trace.so
