#!/bin/bash
# A/B of staging / buffering variants: per-layer times from bench.py's HIP events.
cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out/ab
run () { name=$1; shift; env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/ab/$name.json 2>/dev/null; echo "== $name"; python tools/show_bench.py gpurun_out/ab/$name.json | grep -E "ms/step|^[012] "; }
run p8 CASMVS_NO_DB=1 CASMVS_NO_VEC4=1
run p16 CASMVS_NO_DB=1 CASMVS_NO_VEC4=1 CASMVS_ABLATE=16
run p32 CASMVS_NO_DB=1 CASMVS_NO_VEC4=1 CASMVS_ABLATE=32
