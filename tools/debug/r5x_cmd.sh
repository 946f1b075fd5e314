#!/bin/bash
# the bench line with the two-stream measurements; and the headline launched as 2 streams x batch 4
timeout 500 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train-step > $GRAFT_REPO_ROOT/gpurun_out/r5x/bench_1s.json 2> $GRAFT_REPO_ROOT/gpurun_out/r5x/bench_1s.err; tail -2 $GRAFT_REPO_ROOT/gpurun_out/r5x/bench_1s.err
timeout 500 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train-step --streams 2 --batch 4 --no-events > $GRAFT_REPO_ROOT/gpurun_out/r5x/bench_2s.json 2> $GRAFT_REPO_ROOT/gpurun_out/r5x/bench_2s.err; tail -2 $GRAFT_REPO_ROOT/gpurun_out/r5x/bench_2s.err
python - <<'PY'
import json, os
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r5x/"
for f in ("bench_1s.json", "bench_2s.json"):
    d = json.loads(open(root + f).read().strip().splitlines()[-1])
    print(f, "value", round(d["value"], 1), "ms", round(d["ms_per_step"], 3), "median", round(d["median_ms_per_step"], 3), d["config"]["launch"])
    for k in ("two_streams", "single_stream"):
        if k in d: print("   ", k, round(d[k]["value"], 1), round(d[k]["ms_per_step"], 3))
    if "batch1" in d:
        print("    batch1", round(d["batch1"]["value"], 1), {k: round(v["value"], 1) for k, v in d["batch1"].items() if isinstance(v, dict) and "value" in v})
PY
