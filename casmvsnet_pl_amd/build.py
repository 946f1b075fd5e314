"""Builds libcasmvs_hip.so (the C-ABI HIP library, include/casmvs.h) in-tree with hipcc for gfx950.

`python -m casmvsnet_pl_amd.build` or `__graft_entry__.build()`.  hipcc cross-compiles without a
GPU; the .so is git-ignored but travels with the repo snapshot to the GPU box.
"""
import os
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO_ROOT = os.path.dirname(PKG_DIR)
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "libcasmvs_hip.so")
SOURCES = ["abi.hip", "costvol.hip", "depth_ops.hip", "conv3d_mfma.hip", "conv2d_mfma.hip"]
HEADERS = [os.path.join(CSRC, "common.h"), os.path.join(REPO_ROOT, "include", "casmvs.h")]


def _sources():
    return [os.path.join(CSRC, s) for s in SOURCES if os.path.isfile(os.path.join(CSRC, s))]


def is_stale():
    if not os.path.isfile(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(f) > t for f in _sources() + HEADERS)


def build_library(force=False, verbose=False):
    """Compile every HIP source into one shared library.  Returns the library path."""
    if not force and not is_stale():
        return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
           "-I" + os.path.join(REPO_ROOT, "include"), "-I" + CSRC] + _sources() + ["-o", LIB_PATH + ".tmp"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + res.stdout)
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
