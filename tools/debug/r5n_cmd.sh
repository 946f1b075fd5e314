#!/bin/bash
# eight wait states behind every float32 matrix instruction of the PX layer form: the co-residency fault at 1000 rounds, and what the wait states cost
for v in hip ab1; do
  echo "== library $v"
  LD_PRELOAD=$GRAFT_REPO_ROOT/casmvsnet_pl_amd/libcasmvs_$v.so timeout 280 tools/probes/bin/coresidency_lib_victim $([ $v = hip ] && echo 100 || echo 1000) px 2>&1 | grep -E "beside"
  CASMVS_LIB_PATH=$GRAFT_REPO_ROOT/casmvsnet_pl_amd/libcasmvs_$v.so timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-batch1 --no-train-step --conv0-mode f32 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('all-float32 forward:', round(d['value'],1), 'depth-maps/s', round(d['ms_per_step'],3), 'ms; conv0', [d['stage_ms_per_step'][f'costreg_{l}/conv0'] for l in (2,1,0)], 'conv11', [d['stage_ms_per_step'][f'costreg_{l}/conv11'] for l in (2,1,0)])"
  CASMVS_LIB_PATH=$GRAFT_REPO_ROOT/casmvsnet_pl_amd/libcasmvs_$v.so timeout 200 python bench.py --mode train --steps 20 --warmup 5 2>/dev/null | grep -o '"train_step_ms": [0-9.]*'
done
