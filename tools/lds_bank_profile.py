"""LDS bank-conflict profile of the kernels - on the CPU, from the kernels' own source (no GPU).

    python tools/lds_bank_profile.py run_kernels3 [mode] [--lines]

builds tests/hipemu/<driver>.cpp with the compiler's memory-access hooks (-fsanitize=thread) linked against tests/hipemu/lds_profile.cpp instead of the
sanitizer, runs it, and prints per kernel and source line the LDS wave-instructions of the run with their LDS-array cycles under the bank rules of
MI355X_MICROARCH.md (lane groups, bank = dword mod 32 / 64, broadcasts) next to the conflict-free count.  `cycles / ideal` = 1.00 is conflict-free;
the GPU's counters for the same thing are SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE.  A store additionally moves its registers to the LDS (4 / 6 / 13 cycles per
4- / 8- / 16-byte wave-instruction), which hides that many array cycles: the second ratio of the `writes` line.  The accesses are priced as the source makes them (explicit 8- /
16-byte vectors); the device compiler may merge neighbouring scalar accesses.  Test infrastructure: nothing here is part of the product."""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = os.environ.get("HIPEMU_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
CXX = os.environ.get("HIPEMU_CXX") or os.path.join(LLVM, "clang++")


def build(driver, workdir):
    inc = ["-I" + os.path.join(ROOT, "tests", "hipemu"), "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "casmvsnet_pl_amd", "csrc")]
    obj, hooks, exe = (os.path.join(workdir, n) for n in (driver + ".o", "lds_profile.o", driver + "_ldsprof"))
    subprocess.run([CXX, "-std=c++20", "-O1", "-g", "-fno-pie", "-fsanitize=thread", "-DHIPEMU_LDS_PROFILE", "-pthread", "-DCASMVS_SPLIT_NOASM", *inc, "-x", "c++",
                    "-c", os.path.join(ROOT, "tests", "hipemu", driver + ".cpp"), "-o", obj], check=True, capture_output=True, text=True)
    subprocess.run([CXX, "-std=c++20", "-O2", "-c", os.path.join(ROOT, "tests", "hipemu", "lds_profile.cpp"), "-o", hooks], check=True, capture_output=True, text=True)
    subprocess.run([CXX, "-no-pie", "-pthread", obj, hooks, "-o", exe], check=True, capture_output=True, text=True)
    return exe


def symbolize(exe, addresses):
    """address -> (kernel function, innermost file:line) through llvm-symbolizer's inlined frames."""
    out = subprocess.run([os.path.join(LLVM, "llvm-symbolizer"), "--obj=" + exe, "--inlines", "--functions=short", *["0x%x" % (a - 1) for a in addresses]],
                         check=True, capture_output=True, text=True).stdout
    result = {}
    for a, block in zip(addresses, out.strip().split("\n\n")):
        lines = block.strip().splitlines()
        frames = [(lines[i], lines[i + 1]) for i in range(0, len(lines) - 1, 2)]
        # the innermost frame inside a kernel source (the buffer helpers of buffer_ops.h / the emulator's header are inlined into it)
        inner = next((loc for _, loc in frames if ".hip:" in loc), next((loc for _, loc in frames if "hip_runtime.h" not in loc and "buffer_ops.h" not in loc),
                                                                        frames[0][1] if frames else "?"))
        kernel = next((fn for fn, _ in frames if fn.endswith("_kernel") or "_kernel<" in fn), frames[-1][0] if frames else "?")
        m = re.match(r"(.*?):(\d+):\d+$", inner)
        result[a] = (kernel, os.path.basename(m.group(1)) + ":" + m.group(2) if m else inner)
    return result


def profile(driver, mode="quick", workdir=None, exe=None):
    """-> {(kernel, 'file:line', 'R'|'W', bytes): [wave-instructions, cycles, ideal, worst, cycles and ideal with the store transfer as the floor]}"""
    with tempfile.TemporaryDirectory() as tmp:
        workdir = workdir or tmp
        exe = exe or build(driver, workdir)
        report = os.path.join(workdir, driver + ".lds")
        run = subprocess.run([exe, mode], env=dict(os.environ, HIPEMU_LDS_REPORT=report), capture_output=True, text=True)
        if run.returncode != 0 or "ALL OK" not in run.stdout:
            raise RuntimeError(run.stdout[-2000:] + run.stderr[-2000:])
        rows = []
        for line in open(report):
            m = re.match(r"LDS 0x([0-9a-f]+) ([RW]) (\d+) n=(\d+) cycles=(\d+) ideal=(\d+) worst=(\d+) lanes=\d+ eff=(\d+) eff_ideal=(\d+)", line)
            if m:
                rows.append((int(m.group(1), 16), m.group(2), *(int(m.group(i)) for i in range(3, 10))))
        glb = []
        global EPOCH_LINES
        EPOCH_LINES = {m.group(1): (int(m.group(2)), int(m.group(3))) for m in (re.match(r"GLBSUM ([RW]) epoch_lines64=(\d+) epoch_lines128=(\d+)", ln)
                                                                                 for ln in open(report)) if m}
        for line in open(report):
            m = re.match(r"GLB 0x([0-9a-f]+) ([RW]) (\d+) n=(\d+) lanes=(\d+) bytes=(\d+) lines64=(\d+) lines128=(\d+)", line)
            if m:
                glb.append((int(m.group(1), 16), m.group(2), *(int(m.group(i)) for i in range(3, 9))))
        sym = symbolize(exe, sorted({r[0] for r in rows} | {r[0] for r in glb}))
    global GLOBAL_TABLE
    GLOBAL_TABLE = collections.OrderedDict()
    for addr, rw, nbytes, n, lanes, nb, l64, l128 in glb:
        kernel, where = sym[addr]
        t = GLOBAL_TABLE.setdefault((kernel, where, rw, nbytes), [0, 0, 0, 0, 0])
        for i, v in enumerate((n, lanes, nb, l64, l128)):
            t[i] += v
    table = collections.OrderedDict()
    for addr, rw, nbytes, n, cyc, ideal, worst, eff, eff_ideal in rows:
        kernel, where = sym[addr]
        t = table.setdefault((kernel, where, rw, nbytes), [0, 0, 0, 0, 0, 0])
        t[0] += n
        t[1] += cyc
        t[2] += ideal
        t[3] = max(t[3], worst)
        t[4] += eff
        t[5] += eff_ideal
    return table


EPOCH_LINES = {}   # of the last profile(): {'R'|'W': (distinct 64-byte lines, distinct 128-byte lines) per (workgroup, barrier epoch), summed over the run}
GLOBAL_TABLE = collections.OrderedDict()   # of the last profile(): {(kernel, 'file:line', 'R'|'W', bytes): [wave-instructions, lanes, bytes, 64-byte lines, 128-byte lines]}


def global_per_kernel(table=None):
    tot = collections.OrderedDict()
    for (kernel, _, rw, _), vals in (GLOBAL_TABLE if table is None else table).items():
        t = tot.setdefault(kernel, {"R": [0] * 5, "W": [0] * 5})[rw]
        for i, v in enumerate(vals):
            t[i] += v
    return tot


def per_kernel(table):
    tot = collections.OrderedDict()
    for (kernel, _, rw, _), (n, cyc, ideal, _, eff, eff_ideal) in table.items():
        t = tot.setdefault(kernel, {"R": [0, 0, 0, 0, 0], "W": [0, 0, 0, 0, 0]})[rw]
        for i, v in enumerate((n, cyc, ideal, eff, eff_ideal)):
            t[i] += v
    return tot


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    driver = args[0] if args else "run_kernels3"
    mode = args[1] if len(args) > 1 else "quick"
    table = profile(driver, mode)
    print(f"LDS bank profile of tests/hipemu/{driver}.cpp ({mode}): wave-instructions, LDS-array cycles vs conflict-free cycles (1.00 = no bank conflict)")
    gtot = global_per_kernel()
    if EPOCH_LINES:
        print("whole run, distinct lines per (workgroup, barrier epoch) - re-use inside an epoch counted once, as an L1 would: loads %d x 64 B / %d x 128 B, stores %d / %d"
              % (*EPOCH_LINES.get("R", (0, 0)), *EPOCH_LINES.get("W", (0, 0))))
    ltot = per_kernel(table)
    for kernel in list(ltot) + [k for k in gtot if k not in ltot]:
        t = ltot.get(kernel, {"R": [0] * 5, "W": [0] * 5})
        print(f"{kernel}")
        for rw, name in (("R", "buffer loads "), ("W", "buffer stores")):
            n, lanes, nb, l64, l128 = gtot.get(kernel, {"R": [0] * 5, "W": [0] * 5})[rw]
            if n:
                print(f"    {name} {n:8d} wave-instructions  {nb:10d} bytes in {l64:8d} 64-byte lines ({nb / (64 * l64):.2f} of the lines' bytes used) / "
                      f"{l128:8d} 128-byte lines ({nb / (128 * l128):.2f}); {l64 / n:.1f} lines per instruction")
                if "--lines" in sys.argv:
                    for (k, where, r, nbytes), (n2, lanes2, nb2, a64, a128) in GLOBAL_TABLE.items():
                        if k == kernel and r == rw:
                            print(f"        {where:32s} {r}{nbytes:<3d} n={n2:7d}  {lanes2 / n2:5.1f} lanes, {a64 / n2:5.2f} 64-byte / {a128 / n2:5.2f} 128-byte lines per instruction, {nb2 / (64 * a64):.2f} / {nb2 / (128 * a128):.2f} used")
        for rw, name in (("R", "reads "), ("W", "writes")):
            n, cyc, ideal, eff, eff_ideal = t[rw]
            if n:
                print(f"    {name} {n:8d} wave-instructions  {cyc:9d} cycles  {ideal:9d} conflict-free  -> {cyc / ideal:.2f}x" +
                      (f"   (with the stores' register transfer as the floor: {eff / eff_ideal:.2f}x)" if rw == "W" else ""))
        if "--lines" in sys.argv:
            for (k, where, rw, nbytes), (n, cyc, ideal, worst, _, _) in sorted(table.items(), key=lambda kv: (kv[0][0] != kernel, -(kv[1][1] - kv[1][2]))):
                if k == kernel:
                    print(f"        {where:32s} {rw}{nbytes:<3d} n={n:7d}  {cyc / n:5.2f} cycles per instruction ({ideal / n:4.2f} conflict-free), worst {worst}")


if __name__ == "__main__":
    main()
