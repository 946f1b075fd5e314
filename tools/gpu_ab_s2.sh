#!/bin/bash
# A/B of the stride-2 tile / staging variants: prints conv1 / conv3 / conv5 times of the bench line.
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out/ab_s2
for t in 0 1 2 3; do for v in 0 1; do
  if [ $v = 1 ]; then export CASMVS_S2_VEC4=1; else unset CASMVS_S2_VEC4; fi
  CASMVS_S2_TILE=$t python bench.py --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/ab_s2/t${t}_v${v}.json 2>/dev/null
done; done
