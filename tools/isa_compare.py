"""Is the device code of the kernels that exist on both sides unchanged?  CPU only (hipcc cross-compiles gfx950).

    python tools/isa_compare.py casmvsnet_pl_amd/csrc/conv0_splitf16.hip [git-rev, default HEAD]

compiles the working-tree version and the `git show <rev>:<file>` version of ONE .hip file with the library's flags (--save-temps), and compares the
instruction streams kernel by kernel (labels renumbered, comments and directives dropped).  A kernel template that gained a defaulted trailing parameter is
matched by its old name (`..., 0>` -> `...>`).  Used at the end of round 3, when no GPU was left, to add opt-in template parameters to production kernels:
"identical" means the default path runs the same instructions as the build that was validated on the MI355X."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from casmvsnet_pl_amd.build import FLAGS  # noqa: E402

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
CXXFILT = os.environ.get("CXXFILT", "c++filt")


def device_asm(source_path, include_dir, workdir):
    os.makedirs(workdir, exist_ok=True)
    flags = [f for f in FLAGS if f != "-fPIC"] + ["-fPIC"]
    subprocess.run([HIPCC, *flags, "-I" + os.path.join(ROOT, "include"), "-I" + include_dir, "--save-temps", "-c", source_path, "-o", "x.o"], cwd=workdir, check=True,
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
        elif t and not t.startswith("."):
            out[cur].append(re.sub(r"\.LBB\d+_\d+", ".LBB", t))
    return out


def demangle(names):
    res = subprocess.run([CXXFILT], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    out = []
    for r in res:
        r = re.sub(r"^void ", "", r.replace("(anonymous namespace)::", ""))
        cut = r.find(">(") + 1 if ">(" in r else r.find("(")
        out.append(r[:cut] if cut > 0 else r)
    return out


def same_but_defaults(old, new):
    """`new` is `old` with defaulted trailing template arguments (0 / false) appended"""
    return old.endswith(">") and new.startswith(old[:-1]) and re.fullmatch(r"(, (0|false))+>", new[len(old) - 1:]) is not None


def main():
    path = sys.argv[1]
    rev = sys.argv[2] if len(sys.argv) > 2 else "HEAD"
    rel = os.path.relpath(os.path.abspath(path), ROOT)
    csrc = os.path.dirname(os.path.abspath(path))
    with tempfile.TemporaryDirectory() as tmp:
        new = kernels(device_asm(os.path.abspath(path), csrc, os.path.join(tmp, "new")))
        old_src = os.path.join(tmp, os.path.basename(path))
        open(old_src, "w").write(subprocess.run(["git", "show", f"{rev}:{rel}"], cwd=ROOT, check=True, capture_output=True, text=True).stdout)
        old = kernels(device_asm(old_src, csrc, os.path.join(tmp, "old")))   # (the headers it includes are the working tree's)
    old_names, new_names = dict(zip(demangle(list(old)), old)), dict(zip(demangle(list(new)), new))
    bad = 0
    for name, key in old_names.items():
        cand = [n for n in new_names if n == name or same_but_defaults(name, n) or same_but_defaults(n, name)]   # trailing default arguments added OR removed
        if not cand:
            print(f"  {name}: gone")
            bad += 1
            continue
        same = old[key] == new[new_names[cand[0]]]
        bad += not same
        print(f"  {name}: {'identical' if same else 'DIFFERENT'} ({len(old[key])} -> {len(new[new_names[cand[0]]])} instructions)" + ("" if cand[0] == name else f"   [now {cand[0]}]"))
    for n in new_names:
        if n not in old_names and not any(same_but_defaults(o, n) or same_but_defaults(n, o) for o in old_names):
            print(f"  {n}: new ({len(new[new_names[n]])} instructions)")
    print("unchanged" if not bad else f"{bad} kernels differ")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
