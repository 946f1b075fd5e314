#!/bin/bash
# A/B ablation of the dominant kernel (conv16 PX = CostRegNet.conv0): staging-only vs MFMA-only.
cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out/ablate
for a in 0 1 2; do
  CASMVS_ABLATE=$a python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/ablate/abl$a.json 2>/dev/null
  python - <<PY
import json
j=json.load(open('gpurun_out/ablate/abl$a.json'))
s=j['stage_ms_per_step']
print('ABL=$a conv0 ms (L2,L1,L0):', [s[f'costreg_{l}/conv0'] for l in (2,1,0)], 'step', round(j['ms_per_step'],3))
PY
done
