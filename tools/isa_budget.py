"""Instruction budget of a kernel's loops from the compiled device code.  CPU only (hipcc cross-compiles gfx950).

    python tools/isa_budget.py casmvsnet_pl_amd/csrc/costvol_lds.hip 'costvol_lds_kernel<16, 16, 0, 32, 8, 2, 1>' [--loops] [--dump FILE]

Compiles ONE .hip file with the library's flags, finds the kernel whose demangled name contains the pattern, splits its instruction stream at the
backward branches (a loop = label ... s_cbranch back to it) and prints, per loop (innermost first by size) and for the whole kernel, the instruction
count by class:
    valu-fp     v_fma / v_mul / v_add / v_sub / v_pk_* float arithmetic (incl. f64)
    valu-int    integer / logic / shift / lshl_add / mad_u32 address arithmetic
    valu-cmp    v_cmp* / v_cndmask (selects, bounds tests)
    valu-cvt    conversions, v_floor / v_fract / v_rcp / v_trunc
    valu-mov    v_mov / v_readlane / v_readfirstlane / DPP moves / v_perm
    lds         ds_read* / ds_write* / ds_add* / ds_bpermute
    vmem-load   buffer_load / global_load / scratch_load
    vmem-store  buffer_store / global_store / atomics
    salu        s_* arithmetic, moves, compares (scalar unit: co-issues)
    wait        s_waitcnt / s_barrier / s_nop
    branch      s_cbranch / s_branch
    mfma        v_mfma*
A wave issues one instruction every ~5 cycles whatever it is: the COUNT of vector + LDS + memory instructions per iteration is the
loop's run time in issue slots.  Used for the per-(pixel, plane, view) budget of the cost-volume kernels."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from casmvsnet_pl_amd.build import FLAGS  # noqa: E402

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def classify(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith(("ds_",)):
        return "lds"
    if op.startswith(("buffer_load", "global_load", "scratch_load", "flat_load")):
        return "vmem-load"
    if op.startswith(("buffer_store", "global_store", "scratch_store", "flat_store", "buffer_atomic", "global_atomic", "flat_atomic")):
        return "vmem-store"
    if op.startswith(("s_waitcnt", "s_barrier", "s_nop", "s_sleep")):
        return "wait"
    if op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc", "s_swappc")):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    if op.startswith(("v_cmp", "v_cndmask", "v_cmpx")):
        return "valu-cmp"
    if op.startswith(("v_cvt", "v_floor", "v_fract", "v_rcp", "v_trunc", "v_rndne", "v_ceil", "v_rsq", "v_sqrt", "v_exp", "v_log", "v_frexp", "v_ldexp")):
        return "valu-cvt"
    if op.startswith(("v_mov", "v_readlane", "v_readfirstlane", "v_writelane", "v_perm", "v_swap", "v_accvgpr", "v_bfi", "v_permlane")):
        return "valu-mov"
    fp = ("v_fma", "v_mul_f", "v_add_f", "v_sub_f", "v_pk_fma", "v_pk_mul_f", "v_pk_add_f", "v_mac_f", "v_fmac", "v_max_f", "v_min_f", "v_max3_f", "v_med3_f", "v_mad_f", "v_div_",
          "v_mul_legacy", "v_subrev_f", "v_pk_max_f", "v_pk_min_f", "v_fmaak", "v_fmamk")
    if op.startswith(fp) or re.match(r"v_(add|mul|sub|fma|max|min)_f(16|32|64)", op):
        return "valu-fp"
    if op.startswith("v_"):
        return "valu-int"
    return "other"


CLASSES = ["valu-fp", "valu-int", "valu-cmp", "valu-cvt", "valu-mov", "lds", "vmem-load", "vmem-store", "mfma", "salu", "wait", "branch", "other"]


def device_asm(source_path, extra=()):
    with tempfile.TemporaryDirectory() as wd:
        flags = [f for f in FLAGS if f != "-fPIC"] + ["-fPIC"] + list(extra)
        subprocess.run([HIPCC, *flags, "-I" + os.path.join(ROOT, "include"), "-I" + os.path.dirname(os.path.abspath(source_path)), "--cuda-device-only", "-S",
                        os.path.abspath(source_path), "-o", "k.s"], cwd=wd, check=True, capture_output=True, text=True)
        return open(os.path.join(wd, "k.s")).read()


def kernels(asm):
    """mangled name -> list of (label or None, mnemonic, full line)"""
    out = {}
    cur = None
    for line in asm.splitlines():
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = out.setdefault(m.group(1), [])
            continue
        if cur is None:
            continue
        s = line.strip()
        if s.startswith(".end_amdhsa_kernel") or s.startswith(".section") and cur:
            cur = None if s.startswith(".section") else cur
            continue
        m = re.match(r"^(\.LBB\w+):", s)
        if m:
            cur.append((m.group(1), None, s))
            continue
        if not s or s.startswith((".", ";", "//")):
            continue
        cur.append((None, s.split()[0], s))
    return out


def demangle(names):
    res = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return dict(zip(names, res.stdout.splitlines()))


def loops(stream):
    """(start index, end index) of every backward branch's span"""
    pos = {lab: i for i, (lab, _, _) in enumerate(stream) if lab}
    out = []
    for i, (lab, op, line) in enumerate(stream):
        if op and op.startswith(("s_cbranch", "s_branch")):
            tgt = line.split()[-1]
            if tgt in pos and pos[tgt] < i:
                out.append((pos[tgt], i))
    return out


def count(stream):
    c = dict.fromkeys(CLASSES, 0)
    for lab, op, _ in stream:
        if op:
            c[classify(op)] += 1
    return c


def fmt(c):
    vec = sum(c[k] for k in ("valu-fp", "valu-int", "valu-cmp", "valu-cvt", "valu-mov", "lds", "vmem-load", "vmem-store", "mfma"))
    return f"vector-issue {vec:5d} | " + " ".join(f"{k} {c[k]}" for k in CLASSES if c[k])


def main():
    src, pat = sys.argv[1], sys.argv[2]
    extra = [a for a in sys.argv[3:] if a.startswith("-D")]
    asm = device_asm(src, extra)
    ks = kernels(asm)
    names = demangle(list(ks))
    hits = [k for k in ks if pat in names[k]]
    if not hits:
        print("no kernel matches; candidates:")
        for k in ks:
            print("  ", names[k][:200])
        sys.exit(1)
    for k in hits:
        st = ks[k]
        print("==", names[k].split("(")[0])
        print("   whole kernel:", fmt(count(st)))
        ls = sorted(set(loops(st)), key=lambda ab: ab[1] - ab[0])
        for a, b in ls:
            inner = [x for x in ls if x != (a, b) and a <= x[0] and x[1] <= b]
            print(f"   loop {st[a][0]} ({b - a} lines{', contains ' + str(len(inner)) + ' loop(s)' if inner else ''}):", fmt(count(st[a:b + 1])))
        if "--dump" in sys.argv:
            with open(sys.argv[sys.argv.index("--dump") + 1], "w") as f:
                f.write("\n".join(x[2] for x in st))


if __name__ == "__main__":
    main()
