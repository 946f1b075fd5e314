"""Turns the output of `tools/gpu_run.sh <tag> sq` (two rocprofv3 --pmc passes of SQ counters over the torch-free step runner) into the per-kernel table of
wave states:   python tools/summarize_sq.py gpurun_out/<tag> profiles/r05_sq_counters_step.md
Fractions of SQ_WAVE_CYCLES (MI355X_MICROARCH.md): WAIT_ANY = parked at s_waitcnt / barrier, WAIT_INST_ANY = issue stall, ACTIVE_INST_ANY = issuing (disjoint);
ACTIVE_INST_VALU / LDS / VMEM / SCA = which unit the issuing cycles went to.  One row per (kernel, grid size), sorted by time per step."""
import collections
import csv
import os
import sys

src, out = sys.argv[1], sys.argv[2]
STEPS = int(os.environ.get("SQ_STEPS", "5"))   # launches of a once-per-step kernel in the profiled command: 3 steps + 1 warm-up + the runner's instrumented step


def short(k):
    return k.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0]


def collect(name):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.Counter()
    for r in csv.DictReader(open(os.path.join(src, name, "pmc_counter_collection.csv"))):
        k = (short(r["Kernel_Name"]), int(r["Grid_Size"]))
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[(k, r["Counter_Name"])] += 1
    return agg, n


state, n_state = collect("sq_state")
units, _ = collect("sq_units")
dur = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(os.path.join(src, "sq_state", "pmc_kernel_trace.csv"))):
    k = (short(r["Kernel_Name"]), int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]))
    dur[k][0] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    dur[k][1] += 1
rows = []
for k, c in state.items():
    wc = c.get("SQ_WAVE_CYCLES", 0.0)
    if wc <= 0 or k not in dur:
        continue
    u = units.get(k, {})
    uw = u.get("SQ_WAVE_CYCLES", 0.0) or 1.0
    launches = dur[k][1]
    rows.append((k[0], k[1], launches / STEPS, dur[k][0] / launches, c.get("SQ_WAIT_ANY", 0) / wc, c.get("SQ_WAIT_INST_ANY", 0) / wc, c.get("SQ_ACTIVE_INST_ANY", 0) / wc,
                 u.get("SQ_ACTIVE_INST_VALU", 0) / uw, u.get("SQ_ACTIVE_INST_LDS", 0) / uw, u.get("SQ_ACTIVE_INST_VMEM", 0) / uw, u.get("SQ_ACTIVE_INST_SCA", 0) / uw))
rows.sort(key=lambda r: -r[2] * r[3])
collected = open(os.path.join(src, "collected.txt")).read().strip() if os.path.isfile(os.path.join(src, "collected.txt")) else "unknown"
with open(out, "w") as f:
    f.write("# SQ counters of every kernel of the benched step\n\n"
            f"`rocprofv3 --kernel-trace --pmc` in two passes over `python tools/notorch/step_runner.py --batch 8 --steps 3 --warmup 1` (`tools/gpu_run.sh <tag> sq`, collected {collected}).  "
            "Fractions of SQ_WAVE_CYCLES (MI355X_MICROARCH.md: WAIT_ANY = parked at s_waitcnt / barrier, WAIT_INST_ANY = issue stall, ACTIVE_INST_ANY = issuing; disjoint).  One row per "
            "(kernel, grid size); `us` = mean duration under the counters; sorted by time per step.\n\n"
            "| kernel | grid threads | launches per step | us | parked | issue-stalled | issuing | VALU | LDS | VMEM | scalar |\n|---|---|---|---|---|---|---|---|---|---|---|\n")
    for r in rows:
        if r[2] * r[3] < 5:
            continue
        f.write(f"| `{r[0]}` | {r[1]} | {r[2]:.1f} | {r[3]:.0f} | " + " | ".join(f"{x:.2f}" for x in r[4:]) + " |\n")
print("wrote", out, len(rows), "kernels")
