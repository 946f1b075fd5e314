"""Per-kernel means of the counters of rocprofv3 --pmc passes: python tools/summarize_pmc.py <dir> <pass> [<pass> ...]"""
import collections
import csv
import glob
import os
import re
import sys

root = sys.argv[1]
for ps in sys.argv[2:]:
    files = glob.glob(os.path.join(root, ps, "**", "*counter_collection.csv"), recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in files:
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "costvol" not in k and "homo_warp" not in k and os.environ.get("PMC_ALL") is None:
                continue
            if os.environ.get("PMC_FILTER") and not re.search(os.environ["PMC_FILTER"], k):
                continue
            k = re.sub(r"\(anonymous namespace\)::|void |\(.*", "", k)
            if os.environ.get("PMC_BY_GRID"):   # the same kernel at several launch shapes (cascade levels): one row per grid
                k += " grid " + r.get("Grid_Size", "?")
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("==", ps)
    for k in sorted(acc):
        n = max(len(v) for v in acc[k].values())
        print(f"  {k}  [{n} dispatches]")
        print("     " + "  ".join(f"{c}={sum(v)/len(v):.4g}" for c, v in sorted(acc[k].items())))
