"""Static check for the store-data write-after-read hazard on the compiled device code.  CPU only (hipcc cross-compiles gfx950).

    python tools/store_hazard_lint.py [casmvsnet_pl_amd/csrc/file.hip ...]     (default: every .hip source of the library)   [--window N] [--json]

Found in round 5 (profiles/r05_store_data_hazard.md): on the MI355X a `buffer_store_dwordx4 v[a:a+3], voff, s[rsrc], sN offen` whose NEXT instruction is a
VALU write of one of v[a:a+3] stores the NEW value in some lanes - the store reads its 16 bytes of data per lane over several cycles after issue.  The ISA
lists the pair as a software-visible hazard (a vector-memory store of more than 64 bits followed by a write of its data registers needs wait states) and LLVM
pads it - EXCEPT when the store's soffset operand is a scalar register (GCNHazardRecognizer::createsVALUHazard: "this hazard only exists if the instruction
is not using a register in the soffset field"), which is exactly the form of every volume store in this library (buffer addressing with a scalar plane
offset).  Whether a kernel is hit depends on register allocation: the plane sweep ran clean for four rounds until a 16-plane variant put a depth register
next in line.

Reported: every vector-memory store of more than 64 bits (buffer / global / flat / scratch, dwordx3 / dwordx4) that is followed, within WINDOW issue slots of
the same basic block stream (default 2; s_nop N counts N + 1), by a VALU instruction (or v_readlane-class write) whose destination overlaps the store's data
registers.  LDS / vector-memory LOADS into those registers are not VALU writes (their data arrives after many cycles) and are ignored.  Exit status 1 when a
pair is found."""
import glob
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from casmvsnet_pl_amd.build import FLAGS  # noqa: E402

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
STORE = re.compile(r"^(buffer|global|flat|scratch)_store_(dwordx3|dwordx4|b96|b128)\b")


def regs(tok):
    """'v[4:7]' / 'v12' -> set of vector register numbers (empty for anything else)"""
    m = re.match(r"^v\[(\d+):(\d+)\]$", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"^v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def device_asm(path):
    """The device assembly the library is assembled from: the build's own `<source>.fixed.s` when it is newer than the source, the headers and the build script
    (casmvsnet_pl_amd/build.py keeps it beside the object), else a fresh compile passed through the same packed-float32 rewrite."""
    from casmvsnet_pl_amd import build
    fixed = os.path.join(build.PKG_DIR, "build", os.path.basename(path)[:-4] + ".fixed.s")
    if os.path.dirname(os.path.abspath(path)) == build.CSRC and os.path.isfile(fixed):
        newest = max([os.path.getmtime(path), os.path.getmtime(build.__file__)] + [os.path.getmtime(h) for h in build.HEADERS])
        if os.path.getmtime(fixed) >= newest:
            return open(fixed).read()
    with tempfile.TemporaryDirectory() as wd:
        subprocess.run([HIPCC, *[f for f in FLAGS], "-I" + os.path.join(ROOT, "include"), "-I" + os.path.dirname(os.path.abspath(path)), "--cuda-device-only", "-S",
                        os.path.abspath(path), "-o", "k.s"], cwd=wd, check=True, capture_output=True, text=True)
        return build.rewrite_unsafe_packed(open(os.path.join(wd, "k.s")).read())[0]


def kernels(asm):
    out, cur = {}, None
    for line in asm.splitlines():
        m = re.match(r"^(_Z\S*):", line)
        if m:
            cur = out.setdefault(m.group(1), [])
            continue
        if cur is None:
            continue
        t = re.sub(r";.*", "", line).strip()
        if t.startswith(".Lfunc_end"):
            cur = None
        elif t and not t.startswith(".") and not t.endswith(":"):
            cur.append(t)
        elif t.endswith(":"):
            cur.append("LABEL " + t[:-1])
    return out


def find(stream, window):
    """Pairs (store, VALU writer of its data registers within `window` issue slots).  The scan FOLLOWS control flow: a label takes no slot and is fallen through,
    an unconditional branch takes one slot and continues at its target, a conditional branch takes one slot and continues on BOTH sides - a wide store that is
    the last vector-memory instruction of a loop body is checked against the head of the loop its back edge lands on (round-5 advisor finding: the scan used to
    stop at any label or branch)."""
    labels = {ins.split()[1]: i for i, ins in enumerate(stream) if ins.startswith("LABEL ")}
    hits = []
    for i, ins in enumerate(stream):
        if not STORE.match(ins):
            continue
        ops = [o.strip() for o in ins.split(None, 1)[1].split(",")]
        data = regs(ops[0]) if ins.startswith(("buffer", "scratch")) else (regs(ops[1]) if len(ops) > 1 else set())   # global / flat: vaddr, vdata
        if len(data) < 3:
            continue
        work, seen, found = [(i + 1, 0)], set(), None
        while work and found is None:
            j, slots = work.pop()
            while j < len(stream) and slots < window:
                if (j, slots) in seen:
                    break
                seen.add((j, slots))
                nxt = stream[j]
                j += 1
                if nxt.startswith("LABEL "):
                    continue
                op = nxt.split()[0]
                if op in ("s_endpgm", "s_setpc_b64", "s_swappc_b64", "s_trap"):
                    break
                if op == "s_nop":
                    slots += int(nxt.split()[1], 0) + 1
                    continue
                slots += 1
                if op == "s_branch" or op.startswith("s_cbranch"):
                    target = labels.get(nxt.split()[1])
                    if target is not None:
                        work.append((target, slots))
                    if op == "s_branch":
                        break
                    continue
                if op.startswith("v_") and not op.startswith(("v_cmp", "v_cmpx")):
                    dst = regs(nxt.split(None, 1)[1].split(",")[0].strip())
                    if dst & data:
                        found = (i, ins, nxt, slots)
                        break
        if found:
            hits.append(found)
    return hits


def lint(files=None, window=2):
    """{kernel: [pairs]} over the given sources (default: the library's) - what tests/test_device_code_lints.py asserts to be empty"""
    from concurrent.futures import ThreadPoolExecutor
    files = files or sorted(glob.glob(os.path.join(ROOT, "casmvsnet_pl_amd", "csrc", "*.hip")))
    with ThreadPoolExecutor(max_workers=os.cpu_count() or 4) as pool:
        asms = list(pool.map(device_asm, files))
    report = {}
    for asm in asms:
        ks = kernels(asm)
        names = subprocess.run(["c++filt"], input="\n".join(ks), capture_output=True, text=True).stdout.splitlines()
        for k, name in zip(ks, names):
            hits = find(ks[k], window)
            if hits:
                report[re.sub(r"\(.*", "", name.replace("(anonymous namespace)::", ""))] = [{"store": st, "writer": w, "issue_slots_after_the_store": n} for _, st, w, n in hits]
    return report


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    window = int(sys.argv[sys.argv.index("--window") + 1]) if "--window" in sys.argv else 2
    if "--window" in sys.argv:
        args = [a for a in args if a != str(window)]
    files = args or sorted(glob.glob(os.path.join(ROOT, "casmvsnet_pl_amd", "csrc", "*.hip")))
    report = {}
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=os.cpu_count() or 4) as pool:   # hipcc is a subprocess: the files compile side by side
        asms = list(pool.map(device_asm, files))
    for f, asm in zip(files, asms):
        ks = kernels(asm)
        names = subprocess.run(["c++filt"], input="\n".join(ks), capture_output=True, text=True).stdout.splitlines()
        for k, name in zip(ks, names):
            hits = find(ks[k], window)
            if hits:
                report[re.sub(r"\(.*", "", name.replace("(anonymous namespace)::", ""))] = [{"store": s, "writer": w, "issue_slots_after_the_store": n} for _, s, w, n in hits]
    if "--json" in sys.argv:
        print(json.dumps(report, indent=1))
    else:
        for k, hs in report.items():
            print(k)
            for h in hs:
                print(f"    {h['store']}\n        -> {h['writer']}   ({h['issue_slots_after_the_store']} issue slot(s) later)")
        print(f"{sum(len(h) for h in report.values())} store / writer pair(s) in {len(report)} kernel(s) of {len(files)} file(s), window {window}")
    sys.exit(1 if report else 0)


if __name__ == "__main__":
    main()
