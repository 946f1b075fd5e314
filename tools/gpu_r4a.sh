#!/bin/bash
# First gpurun call of round 4: everything added at the end of round 3 WITHOUT a GPU (the budget was spent) gets its
# measurement here - the parity suite on the re-linked library, the paired-tap fusion kernel against the one-tap-per-load
# kernel (bit equality + time), files -> depth maps with the native PNG decoder / one intra-op thread / stager thread, the
# headline bench line.     /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tools/gpu_r4a.sh'
TAG=${1:-r4a}
ROOTDIR=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOTDIR/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOTDIR
export TMPDIR=/tmp
# the check binaries are git-ignored: a fresh clone that skipped `bash tools/native/build.sh` builds them here (about a minute of box time)
[ -x tools/probes/bin/conv11_prob_check ] || bash tools/native/build.sh > $OUT/native_build.txt 2>&1
# torch-free first (seconds): the kernel written blind at the end of round 3, then the checks of the kernels that are already defaults
PASSED=""
{ for pair in "zmarch:conv0_zm_check 2" "fnet_conv0:fnet_conv0_check" "deconv11:deconv11_check 2" "deconv9:deconv9_check 2" "tail:conv11_prob_check 2"; do
    name=${pair%%:*}; c=${pair#*:}
    timeout 90 tools/probes/bin/$c; rc=$?; echo "-- $c: exit $rc"; [ $rc -eq 0 ] && PASSED="$PASSED $name"
  done
  timeout 30 tools/probes/bin/prob_wgrad_check; timeout 30 tools/probes/bin/fusion_check; timeout 60 python tools/notorch/step_runner.py --batch 8
  # a blind kernel enters the forward only after it passed on its own (a hang there would be a strike); `tail` uses deconv11's packed image (host code) and supersedes its kernel
  echo "== passed their native checks:$PASSED"
  ALL=""
  for x in $PASSED; do
    echo "== --experimental $x"; timeout 60 python tools/notorch/step_runner.py --batch 8 --experimental $x | tail -4
    [ $x = zmarch ] && for y in zmarch32 xshift zmarch,xshift; do echo "== --experimental $y"; timeout 60 python tools/notorch/step_runner.py --batch 8 --experimental $y | tail -4; done   # conv0_zm_check covers the shifted grids too
    ALL="$ALL,$x"
  done
  ALL=${ALL#,}
  echo "$PASSED" | grep -qw tail && ALL=$(echo "$ALL" | sed 's/deconv11,//; s/,deconv11$//; s/^deconv11$//')
  [ -n "$ALL" ] && { echo "== --experimental $ALL"; timeout 60 python tools/notorch/step_runner.py --batch 8 --experimental $ALL | tail -4; }; } > $OUT/native.txt 2>&1
nproc > $OUT/host.txt; cat /sys/fs/cgroup/cpu.max >> $OUT/host.txt 2>/dev/null; lscpu | grep -E "Model name|^CPU\(s\)" | cut -c1-200 >> $OUT/host.txt
# the order is by value per GPU-minute (the whole script is ~30 of the round's 90): the headline line, the experimental set beside it, files -> maps, then the suite
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
[ -n "$ALL" ] && timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-batch1 --experimental $ALL > $OUT/bench_experimental.json 2> $OUT/bench_experimental.err
FT_BATCH=8 FT_GRAPH=1 timeout 300 python tools/gpu_files_throughput.py 49 16 32 > $OUT/files_b8_graph.txt 2>&1
FT_BATCH=8 FT_GRAPH=1 FT_THREADED=1 timeout 300 python tools/gpu_files_throughput.py 49 16 32 > $OUT/files_b8_graph_stager.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest exit: $?" >> $OUT/pytest_gpu.log
# the gated parity cases of the experimental kernels, only when every one of them passed its native check
[ "$(echo $PASSED | wc -w)" -eq 5 ] && { CASMVS_TEST_EXPERIMENTAL=1 timeout 600 python -m pytest tests/test_gpu_experimental.py -m gpu -q --timeout 300 -p no:cacheprovider > $OUT/pytest_experimental.log 2>&1; echo "pytest exit: $?" >> $OUT/pytest_experimental.log; }
# the training step with dropped vs zero-filled gradients (bench.py --mode train; the first form was written without a GPU run)
timeout 400 python bench.py --mode train --steps 20 --warmup 5 > $OUT/bench_train.json 2> $OUT/bench_train.err
timeout 400 python bench.py --mode train --steps 20 --warmup 5 --zero-fill-grads > $OUT/bench_train_zero_fill.json 2> $OUT/bench_train_zero_fill.err
timeout 400 python bench.py --mode train --steps 20 --warmup 5 --wgrad-layout 1 > $OUT/bench_train_wgrad_layout1.json 2> $OUT/bench_train_wgrad_layout1.err
grep -ho '"train_step_ms": [0-9.]*' $OUT/bench_train.json $OUT/bench_train_zero_fill.json $OUT/bench_train_wgrad_layout1.json
cat $OUT/native.txt; tail -3 $OUT/pytest_gpu.log; tail -3 $OUT/pytest_experimental.log 2>/dev/null; tail -4 $OUT/files_b8_graph*.txt; cut -c1-300 $OUT/bench.json; cut -c1-300 $OUT/bench_experimental.json 2>/dev/null
