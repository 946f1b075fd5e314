#!/bin/bash
# ONE parameterised GPU-box runner (replaces the per-call gpu_r3*.sh / gpu_r4a.sh scripts of earlier rounds):
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/gpu_run.sh <tag> <stage> [<stage> ...]'
# Every stage writes under gpurun_out/<tag>/ (the only directory that travels back) and prints a short tail.  Stages are ordered by the
# caller; each is bounded by its own `timeout` so that a hung kernel cannot become a gpurun strike.
#   native       torch-free C-ABI check binaries (tools/native/*.cpp -> tools/probes/bin/) + the step runner: seconds of GPU time
#   step         tools/notorch/step_runner.py --batch 8 (ms per step, per-stage HIP-event times; no torch)
#   probes       tools/probes/*.hip stand-alone programs that exist as binaries (mfma co-residency reproducer, ...)
#   bench        python bench.py --steps 20 --warmup 5 (the driver's line)               BENCH_ARGS="..." adds flags
#   configs      bench lines of the other BASELINE configs (gwc8, 1152x864 V5, 768x576 V7), no CPU baseline
#   suite        python -m pytest tests -m gpu
#   smoke        __graft_entry__.smoke()
#   train        bench.py --mode train (+ --zero-fill-grads)
#   files        tools/gpu_files_throughput.py (files -> depth maps, native decoder, batch-8 graph)
#   pmc          FETCH_SIZE / WRITE_SIZE passes + kernel stats over the step runner (layout of tools/summarize_profile.py)
#   prof         rocprofv3 --kernel-trace --stats over bench.py --steps 10 --no-cpu-baseline
#   costvol      tools/gpu_costvol_probe.py at batch 1 and 8 with dirtied caches (all BASELINE configs)
#   trainprof    rocprofv3 --kernel-trace --stats over bench.py --mode train --steps 10 (per-kernel times of the training step)
#   sq           SQ wave-state counters of every kernel of the step (two PMC passes over the step runner; tools/summarize_sq.py <dir> -> profiles/*_sq_counters_step.md)
#   mfma         matrix-pipe busy cycles, MFMA operation counts and LDS bank conflicts of every kernel of the step (three PMC passes; tools/summarize_mfma.py)
#   coresidency  the packed-float32 op_sel fault: stand-alone reproducer matrix, the library's float32 kernels beside f16 / bf16 neighbours
#                (tools/native/coresidency_lib_victim.cpp), concurrent split-f16 forwards on 2 streams (tools/gpu_mixed_streams.py)
#   cmd          runs "$GPU_RUN_CMD" (one-off experiments without a new script)
TAG=${1:-run}; shift
ROOTDIR=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOTDIR/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOTDIR
export TMPDIR=/tmp
PASSED=""
ALL=""

stage_native () {
  [ -x tools/probes/bin/deconv9_check ] || bash tools/native/build.sh > $OUT/native_build.txt 2>&1
  { for pair in ${NATIVE_CHECKS:-"zmarch:conv0_zm_check:2" "deconv11:deconv11_check:2" "deconv9:deconv9_check:2" "conv_s2:conv_s2_check:8" "conv11_prob:conv11_prob_check:8" "conv2d_k5s2:conv2d_k5s2_check:8"}; do
      name=${pair%%:*}; rest=${pair#*:}; c=${rest%%:*}; arg=${rest#*:}
      [ -x tools/probes/bin/$c ] || continue
      timeout 90 tools/probes/bin/$c $arg; rc=$?; echo "-- $c: exit $rc"; [ $rc -eq 0 ] && PASSED="$PASSED $name"
    done
    for c in prob_wgrad_check fusion_check; do [ -x tools/probes/bin/$c ] && timeout 30 tools/probes/bin/$c; done
    echo "== passed their native checks:$PASSED"; } > $OUT/native.txt 2>&1
  echo "$PASSED" > $OUT/native_passed.txt
  tail -40 $OUT/native.txt
}

stage_step () { timeout 90 python tools/notorch/step_runner.py --batch ${STEP_BATCH:-8} $STEP_ARGS > $OUT/step.txt 2>&1; tail -30 $OUT/step.txt; }

stage_probes () {
  for b in ${PROBE_BINS:-mfma_coresidency_repro}; do
    [ -x tools/probes/bin/$b ] && { timeout 120 tools/probes/bin/$b $PROBE_ARGS > $OUT/probe_$b.txt 2>&1; echo "-- $b: exit $?" >> $OUT/probe_$b.txt; tail -25 $OUT/probe_$b.txt; }
  done
}

stage_bench () {
  nproc > $OUT/host.txt; cat /sys/fs/cgroup/cpu.max >> $OUT/host.txt 2>/dev/null; lscpu | grep -E "Model name|^CPU\(s\)" | cut -c1-200 >> $OUT/host.txt
  # stdout = the compact line the driver parses (<= 4 kB: bench.py compact_line), bench_full.json = the full object (tools/show_bench.py reads either)
  timeout 600 python bench.py --steps 20 --warmup 5 --full-out $OUT/bench_full.json $BENCH_ARGS > $OUT/bench.json 2> $OUT/bench.err
  echo "bench exit: $?"; grep -v "^bench_full: " $OUT/bench.err | tail -3; wc -c $OUT/bench.json; cat $OUT/bench.json
}

stage_configs () {
  for cfg in dtu_640x512_v3_gwc8 dtu_1152x864_v5_var blended_768x576_v7_var; do
    timeout 400 python bench.py --config $cfg --steps 10 --warmup 3 --no-cpu-baseline --full-out $OUT/bench_full_$cfg.json $CONFIG_ARGS > $OUT/bench_$cfg.json 2> $OUT/bench_$cfg.err
    cut -c1-1200 $OUT/bench_$cfg.json
  done
}

stage_suite () {
  timeout ${SUITE_TIMEOUT:-1500} python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider $SUITE_ARGS > $OUT/pytest_gpu.log 2>&1
  echo "pytest exit: $?" >> $OUT/pytest_gpu.log; tail -15 $OUT/pytest_gpu.log
}

stage_smoke () { timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; echo "smoke exit: $?" >> $OUT/smoke.txt; tail -4 $OUT/smoke.txt; }

stage_train () {
  timeout 400 python bench.py --mode train --steps 20 --warmup 5 --full-out "" > $OUT/bench_train.json 2> $OUT/bench_train.err
  for v in ${TRAIN_VARIANTS:-"--zero-fill-grads"}; do
    n=$(echo $v | tr -d ' -'); timeout 400 python bench.py --mode train --steps 20 --warmup 5 --full-out "" $v > $OUT/bench_train_$n.json 2> $OUT/bench_train_$n.err
  done
  grep -ho '"train_step_ms": [0-9.]*' $OUT/bench_train*.json
}

stage_files () {
  FT_BATCH=8 FT_GRAPH=1 timeout 300 python tools/gpu_files_throughput.py 49 16 32 > $OUT/files_b8_graph.txt 2>&1
  FT_BATCH=8 FT_GRAPH=1 FT_THREADED=1 timeout 300 python tools/gpu_files_throughput.py 49 16 32 > $OUT/files_b8_graph_stager.txt 2>&1
  for f in $OUT/files_b8_graph*.txt; do tail -n 5 $f; done
}

stage_pmc () {
  BATCH=${PMC_BATCH:-8}
  RUN="python $ROOTDIR/tools/notorch/step_runner.py --batch $BATCH --steps 3 --warmup 1 $STEP_ARGS"
  run_pmc () { name=$1; shift
    (cd /tmp && timeout 120 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o pmc -- $RUN > $OUT/$name.log 2>&1)
    find $OUT/$name -type f -size +8M -delete 2>/dev/null; }
  run_pmc pmc_fetch FETCH_SIZE
  run_pmc pmc_write WRITE_SIZE
  (cd /tmp && timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o stats -- $RUN > $OUT/prof.log 2>&1)
  find $OUT/prof -name "*.db" -delete 2>/dev/null; find $OUT/prof -type f -size +4M -delete 2>/dev/null
  date -u +%Y-%m-%dT%H:%MZ > $OUT/collected.txt
  sha256sum casmvsnet_pl_amd/libcasmvs_hip.so | cut -c1-64 > $OUT/library_sha256.txt
  python -c "import importlib.util as u; s = u.spec_from_file_location('b', 'casmvsnet_pl_amd/build.py'); m = u.module_from_spec(s); s.loader.exec_module(m); print(m.source_sha16())" > $OUT/source_sha16.txt
  echo "PMC_CMD_NOTE=\"$RUN\" PMC_BATCH=$BATCH PMC_DATE=$(cat $OUT/collected.txt)" > $OUT/summarize_env.txt
  ls $OUT/pmc_fetch $OUT/pmc_write $OUT/prof | head -12; tail -2 $OUT/pmc_fetch.log
}

stage_prof () {
  (cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bench -o stats -- python $ROOTDIR/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-batch1 --full-out $OUT/bench_under_rocprof_full.json > $OUT/bench_under_rocprof.json 2> $OUT/prof_bench.log)
  find $OUT/prof_bench -name "*.db" -delete 2>/dev/null; find $OUT/prof_bench -type f -size +4M -delete 2>/dev/null
  cut -c1-300 $OUT/bench_under_rocprof.json
}

stage_costvol () {
  for b in 1 8; do CV_PROBE_DIRTY=512 timeout 300 python tools/gpu_costvol_probe.py $b > $OUT/costvol_probe_b$b.txt 2>&1; tail -30 $OUT/costvol_probe_b$b.txt; done
}

stage_trainprof () {
  ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_train -o stats -- python $ROOTDIR/bench.py --mode train --steps 10 --warmup 3 --full-out "" $TRAIN_ARGS > $OUT/bench_train_prof.json 2> $OUT/bench_train_prof.err )
  find $OUT/prof_train -name "*.db" -delete 2>/dev/null; find $OUT/prof_train -type f -size +4M -delete 2>/dev/null
  f=$(find $OUT/prof_train -name "*kernel_stats.csv" | head -1); head -${PROF_TOP:-30} $f | cut -c1-220
}

stage_sq () {
  RUN="python $ROOTDIR/tools/notorch/step_runner.py --batch ${PMC_BATCH:-8} --steps 3 --warmup 1 $STEP_ARGS"
  run_sq () { name=$1; shift
    (cd /tmp && timeout 150 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o pmc -- $RUN > $OUT/$name.log 2>&1)
    find $OUT/$name -type f -size +8M -delete 2>/dev/null; }
  run_sq sq_state SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
  run_sq sq_units SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA
  date -u +%Y-%m-%dT%H:%MZ > $OUT/collected.txt
  ls $OUT/sq_state $OUT/sq_units | head -8; tail -2 $OUT/sq_state.log
}

stage_mfma () {   # matrix-pipe and LDS counters of every kernel of the step (three PMC passes over the step runner; tools/summarize_mfma.py <dir> -> profiles/*_mfma_lds_counters.md)
  RUN="python $ROOTDIR/tools/notorch/step_runner.py --batch ${PMC_BATCH:-8} --steps 3 --warmup 1 $STEP_ARGS"
  run_m () { name=$1; shift
    (cd /tmp && timeout 150 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o pmc -- $RUN > $OUT/$name.log 2>&1)
    find $OUT/$name -type f -size +8M -delete 2>/dev/null; }
  run_m mfma_busy SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES
  run_m mfma_ops SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA
  run_m lds_bank SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES
  date -u +%Y-%m-%dT%H:%MZ > $OUT/collected_mfma.txt
  ls $OUT/mfma_busy $OUT/mfma_ops $OUT/lds_bank | head -8; tail -2 $OUT/mfma_busy.log
}

stage_coresidency () {
  [ -x tools/probes/bin/pk_fma_opsel_repro ] && timeout 300 tools/probes/bin/pk_fma_opsel_repro ${OPSEL_ROUNDS:-4} > $OUT/packed_opsel_matrix.txt 2>&1
  grep -c "f16mfma [1-9]" $OUT/packed_opsel_matrix.txt | sed 's/^/forms that fail beside f16 matrix instructions: /'
  [ -x tools/probes/bin/coresidency_lib_victim ] && timeout 500 tools/probes/bin/coresidency_lib_victim ${CORES_ROUNDS:-500} px,ci,ciw,s2,t2 2>&1 | grep -E "beside|REPRODUCED|not reproduced" > $OUT/coresidency_lib_victim.txt
  tail -1 $OUT/coresidency_lib_victim.txt
  { timeout 400 python tools/gpu_mixed_streams.py 2 1 ${MIXED_ROUNDS:-200}; timeout 300 python tools/gpu_mixed_streams.py 2 4 60; timeout 300 python tools/gpu_mixed_streams.py 4 2 60; } 2>&1 | grep -v Warning > $OUT/mixed_streams.txt
  grep -E "differ|depth maps/s" $OUT/mixed_streams.txt
}

stage_cmd () { bash -c "$GPU_RUN_CMD" > $OUT/cmd.txt 2>&1; echo "cmd exit: $?" >> $OUT/cmd.txt; tail -${CMD_TAIL:-60} $OUT/cmd.txt; }

for s in "$@"; do
  echo "===== stage $s ($(date -u +%H:%M:%S))"
  case $s in
    native|step|probes|bench|configs|suite|smoke|train|trainprof|files|pmc|sq|mfma|prof|costvol|coresidency|cmd) stage_$s ;;
    *) echo "unknown stage $s" ;;
  esac
done
echo "===== done ($(date -u +%H:%M:%S))"
