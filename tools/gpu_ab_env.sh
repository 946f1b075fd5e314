#!/bin/bash
# Generic A/B: gpu_ab_env.sh VAR "v1 v2 ..." [repeats]  -> gpurun_out/ab_<VAR>/<value>_<rep>.json (bench lines)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
VAR=$1; VALS=$2; REPS=${3:-2}
mkdir -p gpurun_out/ab_$VAR
for r in $(seq 1 $REPS); do for v in $VALS; do
  env $VAR=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/ab_$VAR/${v}_$r.json 2>/dev/null
done; done
