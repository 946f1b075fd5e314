"""A/B of library builds on the WHOLE forward, torch-free: runs tools/notorch/step_runner.py for each build in turn (A B A B ...,
so that box drift hits both), and prints per build the mean / best ms per step and the per-stage HIP-event times, then every stage's delta
against the first build.   python tools/notorch/ab_step.py [--rounds 3] [--batch 8] libA.so libB.so [libC.so ...]
A variant build: python tools/build_variant.py NAME -DMACRO=VALUE  ->  casmvsnet_pl_amd/libcasmvs_NAME.so"""
import argparse
import os
import re
import subprocess
import sys

ap = argparse.ArgumentParser()
ap.add_argument("libs", nargs="+")
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--f32-layers", default="", help="passed to step_runner.py for every library EXCEPT the first (the baseline): e.g. conv9,conv11")
args = ap.parse_args()
here = os.path.dirname(os.path.abspath(__file__))
res = {lib: {"ms": [], "stages": []} for lib in args.libs}
for _ in range(args.rounds):
    for lib in args.libs:
        extra = ["--f32-layers", args.f32_layers] if (args.f32_layers and lib is not args.libs[0]) else []
        out = subprocess.run([sys.executable, os.path.join(here, "step_runner.py"), "--lib", lib, "--batch", str(args.batch), "--steps", str(args.steps)] + extra,
                             capture_output=True, text=True)
        m = re.search(r"step ([0-9.]+) ms", out.stdout)
        if out.returncode != 0 or not m:
            print(f"{lib}: FAILED\n{out.stdout[-500:]}\n{out.stderr[-1500:]}")
            sys.exit(1)
        res[lib]["ms"].append(float(m.group(1)))
        line = next(l for l in out.stdout.splitlines() if l.startswith("stages"))
        res[lib]["stages"].append({k: float(v) for k, v in re.findall(r"(\w+) ([0-9.]+)", line.split(":", 1)[1])})
base = args.libs[0]
names = list(res[base]["stages"][0])
mean = lambda xs: sum(xs) / len(xs)
for lib in args.libs:
    r = res[lib]
    print(f"{os.path.basename(lib):32s} step mean {mean(r['ms']):.3f} ms  best {min(r['ms']):.3f}  ({args.batch / mean(r['ms']) * 1e3:.0f} depth maps/s)  runs {['%.3f' % x for x in r['ms']]}")
for lib in args.libs[1:]:
    print(f"-- {os.path.basename(lib)} against {os.path.basename(base)} (stage means, ms; negative = faster)")
    for n in names:
        a, b = mean([s[n] for s in res[base]["stages"]]), mean([s[n] for s in res[lib]["stages"]])
        if abs(b - a) > 0.003:
            print(f"   {n:14s} {a:.3f} -> {b:.3f}  ({b - a:+.3f}, {(b / a - 1) * 100:+.1f} %)")
    print(f"   {'step':14s} {mean(res[base]['ms']):.3f} -> {mean(res[lib]['ms']):.3f}  ({(mean(res[lib]['ms']) / mean(res[base]['ms']) - 1) * 100:+.2f} %)")
