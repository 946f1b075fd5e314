"""Static check of DESIGN.md section 2.0's second hazard rule on the compiled device code.  CPU only (hipcc cross-compiles gfx950).

    python tools/mfma_hazard_lint.py [casmvsnet_pl_amd/csrc/file.hip ...]        (default: every source with f16 matrix instructions)

Rule: no floating-point VALU work between a wave's own f16 / bf16 matrix instructions (at two workgroups per CU one staged value in ~500 tiles came out
wrong when the compiler interleaved int -> float conversions, compares and selects with the wave's v_mfma_f32_16x16x32_f16; integer address arithmetic
between them is what every validated kernel has).  The emulation of tests/hipemu cannot see this - it is a property of the instruction schedule - so the
schedule itself is read: per kernel, the MATRIX PHASES (runs of matrix instructions with at most GAP other instructions between neighbours) and every
vector-ALU instruction inside them that is not integer / move / matrix work.  The kernels that were validated bit-stable at full occupancy on the MI355X
(conv0_sf, conv_ci_sf, conv2d_ci_sf, fpn_tail0_sf) are the reference: their phases DO contain floating-point work - the folds / epilogues of accumulator
chains that finished early, i.e. work that consumes matrix results, with up to ~40 matrix instructions of the wave still to issue - and are bit-stable at
two workgroups per CU; the case that failed had floating-point work INDEPENDENT of the matrix results inside the phase.  The kernels of STRICT (written at
the end of round 3, validated on the MI355X in round 4) keep every floating-point instruction outside their phases (sched_barrier), the stricter form.
Exit status 1 if a kernel listed in STRICT has any.

    python tools/mfma_hazard_lint.py --json [files]      prints {kernel: {opcode: count}} of the flagged instructions of EVERY kernel: the golden file
    tests/golden/mfma_phase_fp_instructions.json pins that table for the production build (tests/test_device_code_lints.py): a compiler or source change that
    puts one more floating-point instruction into a matrix phase fails the CPU suite and sends the kernel back to the full-occupancy bit-stability test."""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from casmvsnet_pl_amd.build import FLAGS  # noqa: E402

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
GAP = int(os.environ.get("MFMA_LINT_GAP", "24"))
STRICT = ("conv0_zm_kernel", "conv0_zw_kernel", "deconv11_sf_kernel", "deconv9_sf_kernel", "conv_s2_sf_kernel", "conv11_prob_zfused_kernel", "conv2d_k5s2_sf_kernel", "fnet_conv0_mm_kernel")   # written with every floating-point instruction outside the matrix phases: must stay so
# vector-ALU work that is NOT floating point: integer arithmetic, logic, shifts, moves, lane exchanges, accumulator moves
INT_OK = re.compile(r"^v_(mov|accvgpr|add_lshl|bfrev|add_u|add_i|add_co|addc|sub_u|sub_i|sub_co|subrev_u|subrev_co|subb|mul_lo|mul_hi|mul_u|mul_i|mad_u|mad_i|mad_u64|lshl|lshr|ashr|and|or|xor|"
                    r"not|bfe|bfi|perm|alignbit|alignbyte|readlane|readfirstlane|writelane|swap|nop|lshlrev|lshrrev|ashrrev|add3|lshl_add|lshl_or|and_or|or3|xad|"
                    r"cmp_[a-z]+_[ui]|cmpx_[a-z]+_[ui]|min_[ui]|max_[ui]|med3_[ui]|bcnt|mbcnt|ffb|pk_(add|sub|lshl|lshr|ashr|mul_lo|mad)_[ui]|cndmask)")


def device_asm(path, workdir):
    csrc = os.path.dirname(os.path.abspath(path))
    subprocess.run([HIPCC, *FLAGS, "-I" + os.path.join(ROOT, "include"), "-I" + csrc, "--save-temps", "-c", os.path.abspath(path), "-o", "x.o"], cwd=workdir, check=True,
                   capture_output=True, text=True)
    name = next(f for f in os.listdir(workdir) if f.endswith("-hip-amdgcn-amd-amdhsa-gfx950.s"))
    return open(os.path.join(workdir, name)).read()


def kernels(asm):
    out, cur = {}, None
    for line in asm.splitlines():
        m = re.match(r"^(_Z\S*):", line)
        if m:
            cur = m.group(1)
            out[cur] = []
            continue
        if cur is None:
            continue
        t = re.sub(r";.*", "", line).strip()
        if t.startswith(".Lfunc_end"):
            cur = None
        elif t and not t.startswith(".") and not t.endswith(":"):
            out[cur].append(t.split()[0])
    return out


def lint(ops):
    """-> (matrix instructions, phases, {opcode: count} of flagged vector-ALU instructions inside the phases, selects inside the phases, the largest number of
    matrix instructions of its phase that were still to be issued when a flagged instruction was)"""
    idx = [i for i, o in enumerate(ops) if o.startswith("v_mfma") and ("f16" in o or "bf16" in o)]
    if not idx:
        return 0, 0, {}, 0, None
    phases, start = [], idx[0]
    for a, b in zip(idx, idx[1:]):
        if b - a - 1 > GAP:
            phases.append((start, a))
            start = b
    phases.append((start, idx[-1]))
    flagged, selects, earliest = {}, 0, None
    for lo, hi in phases:
        seen = 0   # matrix instructions of this phase issued so far
        count = sum(1 for o in ops[lo:hi + 1] if o.startswith("v_mfma"))
        for o in ops[lo:hi + 1]:
            if o.startswith("v_mfma"):
                seen += 1
            elif not o.startswith("v_"):
                continue
            elif o.startswith("v_cndmask"):
                selects += 1
            elif not INT_OK.match(o):
                flagged[o] = flagged.get(o, 0) + 1
                left = count - seen   # matrix instructions still to be issued when this one is
                earliest = left if earliest is None else max(earliest, left)
    return len(idx), len(phases), flagged, selects, earliest


def demangle(names):
    res = subprocess.run([os.environ.get("CXXFILT", "c++filt")], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    out = []
    for r in res:
        r = re.sub(r"^void ", "", r.replace("(anonymous namespace)::", ""))
        cut = r.find(">(") + 1 if ">(" in r else r.find("(")
        out.append(r[:cut] if cut > 0 else r)
    return out


def table(files=None):
    """{kernel: {opcode: count}} of the floating-point vector instructions inside f16 / bf16 matrix phases, for every kernel that has matrix instructions."""
    from concurrent.futures import ThreadPoolExecutor
    files = files or sorted(f for f in glob.glob(os.path.join(ROOT, "casmvsnet_pl_amd", "csrc", "*.hip")) if "mfma_f32_16x16x32" in open(f).read())

    def one(path):
        with tempfile.TemporaryDirectory() as tmp:
            ks = kernels(device_asm(path, tmp))
        out = {}
        for name, key in zip(demangle(list(ks)), ks):
            n, _, flagged, _, _ = lint(ks[key])
            if n and "probe" not in name:
                out[name] = dict(sorted(flagged.items()))
        return out
    res = {}
    with ThreadPoolExecutor(max_workers=min(8, len(files))) as pool:
        for part in pool.map(one, files):
            res.update(part)
    return dict(sorted(res.items()))


def main():
    if "--json" in sys.argv[1:]:
        import json
        print(json.dumps(table([a for a in sys.argv[1:] if a != "--json"] or None), indent=1))
        return 0
    files = sys.argv[1:] or sorted(f for f in glob.glob(os.path.join(ROOT, "casmvsnet_pl_amd", "csrc", "*.hip")) if "mfma_f32_16x16x32" in open(f).read())
    bad = 0
    print(f"matrix phases = runs of f16 / bf16 matrix instructions with <= {GAP} other instructions between neighbours; flagged = vector-ALU instructions inside a phase "
          "that are not integer / move / select work")
    for path in files:
        with tempfile.TemporaryDirectory() as tmp:
            ks = kernels(device_asm(path, tmp))
        names = demangle(list(ks))
        print(os.path.relpath(path, ROOT))
        for name, key in zip(names, ks):
            n, phases, flagged, selects, earliest = lint(ks[key])
            if not n:
                continue
            total = sum(flagged.values())
            bad += total > 0 and name.startswith(STRICT)
            detail = (f"  FLAGGED (the earliest with {earliest} matrix instructions still to issue) " + ", ".join(f"{k} x{v}" for k, v in sorted(flagged.items()))) if total else "  clean"
            print(f"  {name:44s} {n:4d} matrix instructions in {phases:2d} phases, {selects} selects inside{detail}")
    print("the STRICT kernels keep all floating-point vector work outside their matrix phases" if not bad else
          f"{bad} of the STRICT kernels have floating-point vector work inside a matrix phase")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
