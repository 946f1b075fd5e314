"""Turns the output directory of `tools/gpu_run.sh <tag> pmc` (gpurun_out/<tag>) into the committed evidence files:
   profiles/<prefix>_kernel_stats.csv       rocprofv3 --kernel-trace --stats summary (as produced)
   profiles/<prefix>_bench.json             the bench line of the same build (batch 2) + batch 1 line
   profiles/<prefix>_pmc_traffic.md/.json   per-kernel HBM-side bytes per launch from the two PMC passes
usage: PMC_BATCH=8 PMC_DATE=$(cat gpurun_out/<tag>/collected.txt) python tools/summarize_profile.py gpurun_out/<tag> r04"""
import collections, csv, json, os, shutil, sys

src, prefix = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "profiles")
os.makedirs(out, exist_ok=True)
if os.path.isfile(os.path.join(src, "prof", "stats_kernel_stats.csv")):
    shutil.copy(os.path.join(src, "prof", "stats_kernel_stats.csv"), os.path.join(out, prefix + "_kernel_stats.csv"))
for a, b in (("bench.json", "_bench.json"), ("bench_batch1.json", "_bench_batch1.json"), ("prof_bench.json", "_bench_under_rocprof.json"),
             ("mfma_rate.txt", "_mfma_rate.txt"), ("parity_report.json", "_parity_report.json")):
    p = os.path.join(src, a)
    if os.path.isfile(p) and os.path.getsize(p) > 0:
        shutil.copy(p, os.path.join(out, prefix + b))


def short(k):
    k = k.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    return k.split("(")[0]


def collect(name):
    rows = list(csv.DictReader(open(os.path.join(src, name, "pmc_counter_collection.csv"))))
    agg = collections.OrderedDict()
    for r in rows:
        d = agg.setdefault((short(r["Kernel_Name"]), int(r["Grid_Size"])), [0.0, 0])
        d[0] += float(r["Counter_Value"])
        d[1] += 1
    return {k: (v / n, n) for k, (v, n) in agg.items()}


fetch, write = collect("pmc_fetch"), collect("pmc_write")
dur = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(os.path.join(src, "pmc_fetch", "pmc_kernel_trace.csv"))):
    k = (short(r["Kernel_Name"]), int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]))
    dur[k][0] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    dur[k][1] += 1
table = []
for k in fetch:
    if not any(s in k[0] for s in ("conv16", "deconv16", "prob_valu", "prob_zwalk", "costvol", "softmax", "hypotheses", "nchw_to", "fpn_lateral", "fpn_tail0",
                                   "conv0_sf", "conv0_sb", "conv0_zm", "conv0_zw", "conv_ci_sf", "conv_s2_sf", "conv11_prob_zfused", "conv2d_ci_sf", "deconv9_sf", "deconv11_sf")):
        continue
    f_kb, n = fetch[k]
    w_kb = write.get(k, (0.0, 0))[0]
    # MI355X_MICROARCH.md (HBM / rocprofv3): on gfx950 FETCH_SIZE reports 1/2 of the bytes of wide (16 B/lane)
    # streaming reads -> doubled; WRITE_SIZE is taken as reported (it matched the algorithmic bytes of every kernel)
    table.append({"kernel": k[0], "grid_threads": k[1], "launches": n, "avg_us_under_pmc": dur[k][0] / max(dur[k][1], 1),
                  "fetch_kb_raw": f_kb, "read_mb_corrected": 2 * f_kb / 1e3, "write_mb": w_kb / 1e3})
table.sort(key=lambda r: -(r["read_mb_corrected"] + r["write_mb"]) * r["launches"])
import hashlib
lib = os.path.join(root, "casmvsnet_pl_amd", "libcasmvs_hip.so")
sha_file = os.path.join(src, "library_sha256.txt")   # written on the GPU box by the pmc stage: the library the passes really ran
if os.path.isfile(sha_file):
    lib_sha = open(sha_file).read().strip()[:16]
else:
    lib_sha = hashlib.sha256(open(lib, "rb").read()).hexdigest()[:16] if os.path.isfile(lib) else None
src_sha_file = os.path.join(src, "source_sha16.txt")   # sources + flags the box's library was compiled from (casmvsnet_pl_amd/build.py: source_sha16)
src_sha = open(src_sha_file).read().strip() if os.path.isfile(src_sha_file) else None
env_file = os.path.join(src, "summarize_env.txt")
collected = open(os.path.join(src, "collected.txt")).read().strip() if os.path.isfile(os.path.join(src, "collected.txt")) else "unknown"
meta = {"collected": os.environ.get("PMC_DATE", collected), "batch": int(os.environ.get("PMC_BATCH", "8")) or None,
        "command": os.environ.get("PMC_CMD_NOTE", open(env_file).read().strip() if os.path.isfile(env_file) else ""), "library_sha16": lib_sha, "source_sha16": src_sha}
json.dump({"meta": meta, "kernels": table}, open(os.path.join(out, prefix + "_pmc_traffic.json"), "w"), indent=1)
with open(os.path.join(out, prefix + "_pmc_traffic.md"), "w") as f:
    f.write("# HBM-side traffic per launch (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, two separate passes)\n\n"
            f"Command: `{meta['command']}` (batch {meta['batch']}, 640x512, 3 views), collected {meta['collected']}, library sha256[:16] {meta['library_sha16']}, source sha256[:16] {meta['source_sha16']}.\n"
            "FETCH_SIZE is doubled (gfx950 rocprofv3 tallies the 128-byte requests of 16 B/lane reads at 64 B: "
            "MI355X_MICROARCH.md, HBM section); WRITE_SIZE as reported.  Both count L2 misses, i.e. include "
            "Infinity-Cache hits.  One row per (kernel, grid size) = per cascade level.\n"
            "`tools/gpu_run.sh <tag> pmc` collects these passes over the torch-free step runner; bench.py reads the newest "
            "profiles/r*_pmc_traffic.json and reports roofline.traffic from it only when the library hash above is the running library's.\n\n"
            "| kernel | grid threads | launches | avg us (under PMC) | read MB | write MB |\n|---|---|---|---|---|---|\n")
    for r in table:
        f.write(f"| `{r['kernel']}` | {r['grid_threads']} | {r['launches']} | {r['avg_us_under_pmc']:.1f} | "
                f"{r['read_mb_corrected']:.1f} | {r['write_mb']:.1f} |\n")
print("wrote", [p for p in sorted(os.listdir(out)) if p.startswith(prefix)])
