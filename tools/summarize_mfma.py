"""Turns the output of `tools/gpu_run.sh <tag> mfma` (three rocprofv3 --pmc passes over the torch-free step runner) into the per-kernel table of matrix-pipe and
LDS counters:   python tools/summarize_mfma.py gpurun_out/<tag> profiles/r06_mfma_lds_counters.md
  matrix pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES)   (the counter is per SIMD, the CU cycles per CU: both summed over the chip)
  co-issue         = SQ_VALU_MFMA_COEXEC_CYCLES / SQ_VALU_MFMA_BUSY_CYCLES      (cycles in which a vector instruction was issued while the matrix pipe was busy)
  executed TFLOP/s = 512 x MFMA_MOPS_{F16, F32} / kernel time                    (what the instruction stream executed: for the split-f16 layers 4 per algorithmic FLOP)
  LDS bank conflict = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE                   (extra LDS-array cycles over all LDS-array cycles)
One row per (kernel, grid size), sorted by time per step."""
import collections
import csv
import os
import sys

src, out = sys.argv[1], sys.argv[2]
STEPS = int(os.environ.get("SQ_STEPS", "5"))


def short(k):
    return k.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0]


def collect(name):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(os.path.join(src, name, "pmc_counter_collection.csv"))):
        agg[(short(r["Kernel_Name"]), int(r["Grid_Size"]))][r["Counter_Name"]] += float(r["Counter_Value"])
    return agg


busy, ops, lds = collect("mfma_busy"), collect("mfma_ops"), collect("lds_bank")
dur = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(os.path.join(src, "mfma_ops", "pmc_kernel_trace.csv"))):
    k = (short(r["Kernel_Name"]), int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]))
    dur[k][0] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    dur[k][1] += 1
rows = []
for k, (us, n) in dur.items():
    b, o, l = busy.get(k, {}), ops.get(k, {}), lds.get(k, {})
    cu = b.get("SQ_BUSY_CU_CYCLES", 0.0)
    mb = b.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    t = us * 1e-6   # total seconds of the n launches under the counters
    f16, f32 = 512.0 * o.get("SQ_INSTS_VALU_MFMA_MOPS_F16", 0.0), 512.0 * o.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0.0)
    rows.append((k[0], k[1], n / STEPS, us / n, mb / (4.0 * cu) if cu else 0.0, b.get("SQ_VALU_MFMA_COEXEC_CYCLES", 0.0) / mb if mb else 0.0,
                 f16 / t / 1e12 if t else 0.0, f32 / t / 1e12 if t else 0.0,
                 l.get("SQ_LDS_BANK_CONFLICT", 0.0) / l["SQ_LDS_IDX_ACTIVE"] if l.get("SQ_LDS_IDX_ACTIVE") else 0.0))
rows.sort(key=lambda r: -r[2] * r[3])
collected = open(os.path.join(src, "collected_mfma.txt")).read().strip() if os.path.isfile(os.path.join(src, "collected_mfma.txt")) else "unknown"
with open(out, "w") as f:
    f.write("# Matrix-pipe and LDS counters of every kernel of the benched step\n\n"
            f"`rocprofv3 --kernel-trace --pmc` in three passes over `python tools/notorch/step_runner.py --batch 8 --steps 3 --warmup 1` (`tools/gpu_run.sh <tag> mfma`, collected {collected}).  "
            "`matrix pipe busy` = SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CU_CYCLES); `co-issue` = SQ_VALU_MFMA_COEXEC_CYCLES / SQ_VALU_MFMA_BUSY_CYCLES; executed TFLOP/s = 512 x "
            "SQ_INSTS_VALU_MFMA_MOPS_{F16,F32} / kernel time (dense peaks: 2 500 f16, 157.3 float32; a split-f16 layer executes 4 f16 FLOPs per algorithmic float32 FLOP); `LDS conflicts` = "
            "SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE.  One row per (kernel, grid size); `us` = mean duration under the counters; sorted by time per step.\n\n"
            "| kernel | grid threads | launches per step | us | matrix pipe busy | co-issue | executed f16 TFLOP/s (of 2 500) | executed f32-MFMA TFLOP/s (of 157.3) | LDS conflicts |\n|---|---|---|---|---|---|---|---|---|\n")
    for r in rows:
        if r[2] * r[3] < 5:
            continue
        f.write(f"| `{r[0]}` | {r[1]} | {r[2]:.1f} | {r[3]:.0f} | {r[4]:.2f} | {r[5]:.2f} | {r[6]:.0f} ({r[6] / 2500:.2f}) | {r[7]:.1f} ({r[7] / 157.3:.2f}) | {r[8]:.2f} |\n")
print("wrote", out, len(rows), "kernels")
