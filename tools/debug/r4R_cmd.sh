#!/bin/bash
# batch-1 / batch-2 per-layer breakdown (bench.py --batch N: stage_ms_per_step of the instrumented pass)
for b in 1 2; do
  timeout 300 python bench.py --batch $b --steps 20 --warmup 5 --no-cpu-baseline --no-batch1 --no-train-step > $OUT/bench_b$b.json 2> $OUT/bench_b$b.err
  python - <<PY
import json
d = json.load(open("$OUT/bench_b$b.json"))
print("== batch $b:", d["value"], "depth-maps/s", d["ms_per_step"], "ms")
print(json.dumps(d["stage_ms_per_step"]))
PY
done
for b in 1 2 4; do python tools/notorch/step_runner.py --batch $b --steps 20 --warmup 5 | head -4; done
