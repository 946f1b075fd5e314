"""Build a variant of libcasmvs_hip.so next to the production library (profiling / A-B builds):
   python tools/build_variant.py NAME -DMACRO=VALUE ...   ->  casmvsnet_pl_amd/libcasmvs_NAME.so (objects in build_NAME/, which
.gpurunignore should list: only the .so travels to the GPU box).  Select it with CASMVS_LIB_PATH."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from casmvsnet_pl_amd import build

name, flags = sys.argv[1], sys.argv[2:]
pkg = os.path.dirname(os.path.abspath(build.__file__))
print(build.build_library(force=False, extra_flags=flags, lib_path=os.path.join(pkg, f"libcasmvs_{name}.so"), obj_dir=os.path.join(pkg, f"build_{name}")))
