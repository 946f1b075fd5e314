"""Builds libcasmvs_hip.so (the C-ABI HIP library, include/casmvs.h) in-tree with hipcc for gfx950, and libcasmvs_io.so
(include/casmvs_io.h: host-side file decoding, plain C++) with g++.

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
SOURCES = ["abi.hip", "costvol.hip", "costvol_lds.hip", "depth_ops.hip", "fusion.hip", "backward.hip", "train.hip", "prob_wgrad.hip", "prob_regress.hip", "conv11_prob_zfused.hip", "fpn_fused.hip", "fpn_fused_sf.hip", "conv0_splitbf16.hip", "conv0_splitf16.hip", "conv0_zmarch.hip", "deconv11_splitf16.hip", "deconv9_splitf16.hip", "conv_ci_splitf16.hip", "conv_s2_splitf16.hip", "conv2d_ci_splitf16.hip", "conv2d_k5s2_splitf16.hip", "debug_disturb.hip", "conv3d_mfma.hip"]
HEADERS = [os.path.join(CSRC, h) for h in ("common.h", "plane_sweep.h", "buffer_ops.h", "softmax_regress.h", "split_f16.h", "fixed_accum.h")] + [os.path.join(REPO_ROOT, "include", "casmvs.h")]


def _sources():
    return [os.path.join(CSRC, s) for s in SOURCES if os.path.isfile(os.path.join(CSRC, s))]


IO_LIB_PATH = os.path.join(PKG_DIR, "libcasmvs_io.so")
IO_SOURCES = [os.path.join(PKG_DIR, "csrc_host", "png_decode.cpp")]
IO_HEADERS = [os.path.join(REPO_ROOT, "include", "casmvs_io.h")]
# no -march: the library travels to hosts with other CPUs (the SIMD it uses is SSE2, part of x86-64)
IO_FLAGS = ["-O3", "-std=c++17", "-fPIC", "-shared", "-pthread", "-Wall", "-Wextra"]

FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC"]


def source_sha16():
    """First 16 hex digits of a sha256 over everything the device code is compiled from (the .hip sources, their headers, the C header, the
    compiler flags): equal hashes = the same kernels, whichever machine linked the .so (hipcc's output is not bit-reproducible across
    builds, so the binary's own hash cannot say that).  Stamps the PMC files and the bench line (bench.py: traffic_source.same_library)."""
    import hashlib
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for path in sorted(_sources() + HEADERS):
        h.update(os.path.basename(path).encode())
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def build_library(force=False, verbose=False, extra_flags=(), lib_path=None, obj_dir=None):
    """Compile every HIP source (one object per source, rebuilt only when it or a header changed, in parallel) and
    link them into one shared library.  Returns the library path.  `extra_flags` / `lib_path` / `obj_dir` build a
    variant next to the production library (tools/build_trace_lib.sh: -DCASMVS_TRACE, -DCASMVS_IEEE_DIV)."""
    lib_path = lib_path or LIB_PATH
    obj_dir = obj_dir or os.path.join(PKG_DIR, "build")
    os.makedirs(obj_dir, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    inc = ["-I" + os.path.join(REPO_ROOT, "include"), "-I" + CSRC]
    hdr_time = max(os.path.getmtime(h) for h in HEADERS)
    jobs, objs = [], []
    for src in _sources():
        obj = os.path.join(obj_dir, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or not os.path.isfile(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_time):
            cmd = [hipcc] + FLAGS + list(extra_flags) + inc + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            jobs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, proc in jobs:
        out, _ = proc.communicate()
        if proc.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n" + out)
    if not jobs and os.path.isfile(lib_path) and os.path.getmtime(lib_path) >= max(os.path.getmtime(o) for o in objs):
        return lib_path
    tmp = f"{lib_path}.{os.getpid()}.tmp"   # per-process name: two ranks building at once never write the same file; os.replace is atomic
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", tmp]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc link failed:\n" + res.stdout)
    os.replace(tmp, lib_path)
    return lib_path


def build_io_library(force=False, verbose=False, lib_path=None):
    """g++ build of libcasmvs_io.so (include/casmvs_io.h: host-side PNG decoding of the input pipeline; no HIP, no torch)."""
    lib_path = lib_path or IO_LIB_PATH
    newest = max(os.path.getmtime(f) for f in IO_SOURCES + IO_HEADERS)
    if not force and os.path.isfile(lib_path) and os.path.getmtime(lib_path) >= newest:
        return lib_path
    # _io.load() builds on first use: several ranks / DataLoader workers of a fresh checkout may get here together.  Each compiles into its OWN
    # temporary file and installs it with an atomic rename - a reader sees the old complete library or a new complete one, never a partial file.
    tmp = f"{lib_path}.{os.getpid()}.tmp"
    cmd = [os.environ.get("CXX", "g++")] + IO_FLAGS + ["-I" + os.path.join(REPO_ROOT, "include")] + IO_SOURCES + ["-o", tmp]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        if os.path.exists(tmp):
            os.unlink(tmp)
        raise RuntimeError("g++ failed on libcasmvs_io.so:\n" + res.stdout)
    os.replace(tmp, lib_path)
    return lib_path


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
    print(build_io_library(force="--force" in sys.argv, verbose=True))
