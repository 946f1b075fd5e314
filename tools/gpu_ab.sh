#!/bin/bash
# A/B of staging / buffering variants: per-layer times from bench.py's HIP events.
cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out/ab
run () { name=$1; shift; env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/ab/$name.json 2>/dev/null; echo "== $name"; python tools/show_bench.py gpurun_out/ab/$name.json | grep -E "ms/step|^[012] "; }
run base A=1
run dbci1 CASMVS_DB_CI=1
run dbci2 CASMVS_DB_CI=2
run dbci3 CASMVS_DB_CI=3
