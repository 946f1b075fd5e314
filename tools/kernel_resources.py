"""Registers, scratch and static LDS of every kernel of the library, from the compiler's metadata.  CPU only (hipcc cross-compiles gfx950).

    python tools/kernel_resources.py [casmvsnet_pl_amd/csrc/file.hip ...]      (default: every source of the library)

Per kernel: vector registers (architectural + accumulation, the unified count that bounds the waves per SIMD: 512 / count), scalar registers, scratch
bytes per lane (spills: must be 0 in a hot kernel), static LDS.  `waves` = waves per SIMD the register count allows (before LDS and launch bounds)."""
import glob
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from casmvsnet_pl_amd.build import FLAGS  # noqa: E402

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def resources(path):
    """-> [(demangled kernel name, vgprs, sgprs, scratch bytes, static LDS bytes)]"""
    csrc = os.path.dirname(os.path.abspath(path))
    with tempfile.TemporaryDirectory() as tmp:
        subprocess.run([HIPCC, *FLAGS, "-I" + os.path.join(ROOT, "include"), "-I" + csrc, "--save-temps", "-c", os.path.abspath(path), "-o", "x.o"], cwd=tmp, check=True,
                       capture_output=True, text=True)
        asm = open(os.path.join(tmp, next(f for f in os.listdir(tmp) if f.endswith("-hip-amdgcn-amd-amdhsa-gfx950.s")))).read()
    rows = []
    for block in asm.split("  - .agpr_count:")[1:]:
        def field(name):
            m = re.search(r"\.%s:\s+(\S+)" % name, block)
            return m.group(1) if m else "0"
        rows.append((field("name"), int(field("vgpr_count")), int(field("sgpr_count")), int(field("private_segment_fixed_size")), int(field("group_segment_fixed_size"))))
    names = subprocess.run([os.environ.get("CXXFILT", "c++filt")], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.splitlines()
    out = []
    for r, n in zip(rows, names):
        n = re.sub(r"^void ", "", n.replace("(anonymous namespace)::", ""))
        cut = n.find(">(") + 1 if ">(" in n else n.find("(")
        out.append((n[:cut] if cut > 0 else n, *r[1:]))
    return out


def main():
    files = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "casmvsnet_pl_amd", "csrc", "*.hip")))
    with ThreadPoolExecutor(max_workers=min(8, len(files))) as pool:
        results = list(pool.map(resources, files))
    spills = 0
    for path, rows in zip(files, results):
        print(os.path.relpath(path, ROOT))
        for name, vgpr, sgpr, scratch, lds in rows:
            spills += scratch > 0
            print(f"  {name[:70]:70s} vgpr {vgpr:3d} (waves {max(1, min(8, 512 // max(vgpr, 1)))})  sgpr {sgpr:3d}  scratch {scratch:5d} B  static LDS {lds:6d} B" + ("   <-- SPILLS" if scratch else ""))
    print(f"{spills} kernels use scratch")
    return 0


if __name__ == "__main__":
    sys.exit(main())
