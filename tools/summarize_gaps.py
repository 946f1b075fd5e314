"""Gaps between consecutive kernels of one step from a rocprofv3 kernel trace (CSV): what a fused multi-layer kernel could save at most.

    python tools/summarize_gaps.py <dir with *_kernel_trace.csv> [--steps-back 1] [--segment conv3:conv9]

Takes the LAST complete step of the run (a step = the kernels between two launches of the first FeatureNet kernel), prints every kernel's duration and the idle
time before the next kernel starts (start[i + 1] - end[i]: dispatch latency + drain / ramp seen from the timestamps), and the sums over the whole step and
over CostRegNet's quarter- / eighth-resolution layers (conv3 .. conv9) of every level.  DESIGN.md 7.2 quotes these sums."""
import csv
import glob
import os
import sys


def load(d):
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no *kernel_trace.csv under {d}")
    rows = []
    with open(sorted(files)[-1]) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    return rows


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name[:name.index("(")] if "(" in name else name


def main():
    rows = [r for r in load(sys.argv[1]) if "copyBuffer" not in r[2] and "fillBuffer" not in r[2]]
    names = [short(r[2]) for r in rows]
    period = next(p for p in range(8, len(names) // 2 + 1) if names[-p:] == names[-2 * p:-p])   # the step = the shortest repeating tail of the launch sequence
    back = int(sys.argv[sys.argv.index("--steps-back") + 1]) if "--steps-back" in sys.argv else 1   # 1: the step before the last (the runner's last step carries HIP events)
    end = len(rows) - back * period
    step = rows[end - period:end]
    t_busy = sum(e - s for s, e, _ in step)
    gaps = [step[i + 1][0] - step[i][1] for i in range(len(step) - 1)]
    print(f"{len(rows)} kernel launches, {period} per step; step {back} from the end:")
    print(f"{'#':>3s} {'kernel':72s} {'us':>9s} {'gap after, us':>14s}")
    for i, (s, e, n) in enumerate(step):
        print(f"{i:3d} {short(n)[:72]:72s} {(e - s) / 1e3:9.1f} {(gaps[i] / 1e3 if i < len(gaps) else 0):14.1f}")
    wall = step[-1][1] - step[0][0]
    print(f"step wall {wall / 1e3:.1f} us = kernels {t_busy / 1e3:.1f} us + gaps {sum(gaps) / 1e3:.1f} us ({100.0 * sum(gaps) / wall:.1f} %), mean gap {sum(gaps) / len(gaps) / 1e3:.2f} us")


if __name__ == "__main__":
    main()
