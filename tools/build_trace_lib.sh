#!/bin/bash
# Builds variant copies of the library next to the production one (cross-compiles without a GPU):
#   libcasmvs_trace.so   -DCASMVS_TRACE     profiling build: shader-clock traces + the CASMVS_* A/B environment switches
#   libcasmvs_cvdirect.so -DCASMVS_TRACE -DCASMVS_CV_DIRECT  LDS cost volume with per-channel dword stores (no transpose)
#   libcasmvs_ieeediv.so -DCASMVS_IEEE_DIV  correctly rounded divisions in the plane sweep (A/B of depth-index flips)
# Select one at run time with CASMVS_LIB_PATH=casmvsnet_pl_amd/<name>.so.
cd "$(dirname "$0")/.." && python - <<'PY'
import os
from casmvsnet_pl_amd import build
pkg = build.PKG_DIR
print(build.build_library(extra_flags=["-DCASMVS_TRACE"], lib_path=os.path.join(pkg, "libcasmvs_trace.so"), obj_dir=os.path.join(pkg, "build_trace")))
print(build.build_library(extra_flags=["-DCASMVS_TRACE", "-DCASMVS_CV_DIRECT"], lib_path=os.path.join(pkg, "libcasmvs_cvdirect.so"), obj_dir=os.path.join(pkg, "build_cvdirect")))
print(build.build_library(extra_flags=["-DCASMVS_IEEE_DIV"], lib_path=os.path.join(pkg, "libcasmvs_ieeediv.so"), obj_dir=os.path.join(pkg, "build_ieeediv")))
PY
