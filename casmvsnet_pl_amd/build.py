"""Builds libcasmvs_hip.so (the C-ABI HIP library, include/casmvs.h) in-tree with hipcc for gfx950, and libcasmvs_io.so
(include/casmvs_io.h: host-side file decoding, plain C++) with g++.

`python -m casmvsnet_pl_amd.build` or `__graft_entry__.build()`.  hipcc cross-compiles without a
GPU; the .so is git-ignored but travels with the repo snapshot to the GPU box.
"""
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO_ROOT = os.path.dirname(PKG_DIR)
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "libcasmvs_hip.so")
SOURCES = ["abi.hip", "costvol.hip", "costvol_lds.hip", "depth_ops.hip", "fusion.hip", "backward.hip", "train.hip", "prob_wgrad.hip", "prob_regress.hip", "conv11_prob_zfused.hip", "fpn_fused.hip", "fpn_fused_sf.hip", "conv0_splitbf16.hip", "conv0_splitf16.hip", "conv0_zmarch.hip", "deconv11_splitf16.hip", "deconv9_splitf16.hip", "conv_ci_splitf16.hip", "conv_s2_splitf16.hip", "conv2d_ci_splitf16.hip", "conv2d_k5s2_splitf16.hip", "fnet_conv0_mm.hip", "debug_disturb.hip", "conv3d_mfma.hip"]
HEADERS = [os.path.join(CSRC, h) for h in ("common.h", "plane_sweep.h", "buffer_ops.h", "softmax_regress.h", "split_f16.h", "fixed_accum.h")] + [os.path.join(REPO_ROOT, "include", "casmvs.h")]


def _sources():
    return [os.path.join(CSRC, s) for s in SOURCES if os.path.isfile(os.path.join(CSRC, s))]


IO_LIB_PATH = os.path.join(PKG_DIR, "libcasmvs_io.so")
IO_SOURCES = [os.path.join(PKG_DIR, "csrc_host", "png_decode.cpp")]
IO_HEADERS = [os.path.join(REPO_ROOT, "include", "casmvs_io.h")]
# no -march: the library travels to hosts with other CPUs (the SIMD it uses is SSE2, part of x86-64)
IO_FLAGS = ["-O3", "-std=c++17", "-fPIC", "-shared", "-pthread", "-Wall", "-Wextra"]

FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC"]
# (+ -DCASMVS_PACKED_OPSEL_SAFE=1 on every source: _compile_source rewrites the device assembly, and casmvs_packed_opsel_safe() says so)


# ---- the packed-float32 operand-selection fault of gfx950 (tools/probes/pk_fma_opsel_repro.hip) -------------------------
# v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 whose LOW result takes the low half of src0 and the HIGH half of a vector-register src1
# (op_sel:[0,1,..]) read that src1 half as ZERO in lanes 48-63 while another wave of the SIMD issues f16 / bf16 matrix instructions (another
# kernel on another stream).  Every other selection is clean (all 64 + 16 + 16 forms measured).  The three instructions commute in
# src0 / src1, so the library is assembled from device assembly in which every such instruction has the two sources (and their selection
# bits) exchanged: op_sel:[1,0,..], one of the clean forms, the same arithmetic bit for bit.
_PACKED_F32 = re.compile(r"^(\s*)(v_pk_(?:fma|mul|add)_f32)\s+([^;\n]*?)\s*(;.*)?$")
_MODIFIER = re.compile(r"\b(op_sel_hi|op_sel|neg_lo|neg_hi):\[([01,]+)\]")


def _parse_packed(line):
    """(indent, mnemonic, [dst, src0, src1(, src2)], {modifier: [bits]}, clamp, comment) of a packed float32 instruction, else None."""
    m = _PACKED_F32.match(line)
    if not m:
        return None
    indent, op, body, comment = m.groups()
    cut = re.search(r"\s(?:op_sel|neg_lo|neg_hi|clamp)\b", body)
    operands = [o.strip() for o in (body[:cut.start()] if cut else body).split(",")]
    tail = body[cut.start():] if cut else ""
    mods = {k: [int(b) for b in v.split(",")] for k, v in _MODIFIER.findall(tail)}
    return indent, op, operands, mods, bool(re.search(r"\bclamp\b", tail)), comment or ""


def packed_f32_is_unsafe(line):
    """True for an instruction of the faulty class: low result from src0's low half and a VECTOR-register src1's high half."""
    p = _parse_packed(line)
    if not p:
        return False
    _, _, operands, mods, _, _ = p
    sel = mods.get("op_sel", [0] * (len(operands) - 1))
    return sel[0] == 0 and sel[1] == 1 and operands[2][:1] in ("v", "a")   # (accumulation registers as a source: not measured, treated as vector registers)


def rewrite_unsafe_packed(asm_text):
    """Device assembly with src0 / src1 of every unsafe packed float32 instruction exchanged -> (text, number of instructions rewritten)."""
    out, count = [], 0
    for line in asm_text.split("\n"):
        if "v_pk_" in line and packed_f32_is_unsafe(line):
            indent, op, operands, mods, clamp, comment = _parse_packed(line)
            n = len(operands) - 1
            operands[1], operands[2] = operands[2], operands[1]
            mods.setdefault("op_sel", [0] * n)
            for bits in mods.values():
                bits[0], bits[1] = bits[1], bits[0]
            tail = "".join(f" {k}:[{','.join(map(str, mods[k]))}]" for k in ("op_sel", "op_sel_hi", "neg_lo", "neg_hi") if k in mods)
            line = f"{indent}{op} {', '.join(operands)}{tail}{' clamp' if clamp else ''}{(' ' + comment) if comment else ''}"
            count += 1
        out.append(line)
    return "\n".join(out), count


def _llvm_bin(hipcc):
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.realpath(hipcc))), "lib", "llvm", "bin")
    return d if os.path.isdir(d) else "/opt/rocm/lib/llvm/bin"


def _compile_source(hipcc, flags, inc, src, obj, verbose=False):
    """One .hip -> one host object carrying the (rewritten) device code: the steps hipcc runs internally, with the device assembly passed
    through rewrite_unsafe_packed between the compiler and the assembler.  Files next to the object: .s (as compiled), .fixed.s, .hsaco."""
    stem, llvm = obj[:-2], _llvm_bin(hipcc)

    def run(cmd):
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if res.returncode != 0:
            raise RuntimeError(f"build step failed on {src}:\n{' '.join(cmd)}\n{res.stdout}")

    flags = list(flags) + ["-DCASMVS_PACKED_OPSEL_SAFE=1"]
    run([hipcc] + flags + inc + ["--cuda-device-only", "-S", src, "-o", stem + ".s"])
    with open(stem + ".s") as f:
        text, count = rewrite_unsafe_packed(f.read())
    with open(stem + ".fixed.s", "w") as f:
        f.write(text)
    run([os.path.join(llvm, "clang"), "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", stem + ".fixed.s", "-o", stem + ".dev.o"])
    run([os.path.join(llvm, "lld"), "-flavor", "gnu", "-m", "elf64_amdgpu", "--no-undefined", "-shared", "-o", stem + ".hsaco", stem + ".dev.o"])
    run([os.path.join(llvm, "clang-offload-bundler"), "-type=o", "-bundle-align=4096", "-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950",
         "-input=/dev/null", "-input=" + stem + ".hsaco", "-output=" + stem + ".hipfb"])
    run([hipcc] + flags + inc + ["--cuda-host-only", "-Xclang", "-fcuda-include-gpubinary", "-Xclang", stem + ".hipfb", "-c", src, "-o", obj])
    return count


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
        if force or not os.path.isfile(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_time, os.path.getmtime(os.path.abspath(__file__))):
            jobs.append((src, obj))
    with ThreadPoolExecutor(max_workers=max(1, min(len(jobs), os.cpu_count() or 4))) as pool:
        futures = [pool.submit(_compile_source, hipcc, FLAGS + list(extra_flags), inc, src, obj, verbose) for src, obj in jobs]
        rewritten = sum(f.result() for f in futures)   # (raises the first failure)
    if verbose and jobs:
        print(f"{rewritten} packed float32 instructions rewritten in {len(jobs)} sources", file=sys.stderr)
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
