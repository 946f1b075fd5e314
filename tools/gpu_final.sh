#!/bin/bash
# Evidence run for profiles/ (one gpurun call): parity tests, bench lines (default + the other BASELINE configs + the
# view-sharded mode at world size 1), rocprofv3 kernel stats, two SEPARATE PMC passes (FETCH_SIZE, WRITE_SIZE) for the
# HBM traffic of the hot kernels, the cost-volume probe, the micro-probes.
TAG=${1:-final}
ROOTDIR=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOTDIR/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOTDIR
export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -E "Marketing Name|Compute Unit|gfx" | head -8 > $OUT/rocminfo.txt
nproc > $OUT/nproc.txt; lscpu | grep -E "Model name|^CPU\(s\)" >> $OUT/nproc.txt
python -c "
from casmvsnet_pl_amd import ops
print('mfma 16x16x4 blocks 2048 TFLOP/s %.1f' % ops.selftest_mfma_rate(1, 2048, 4096))" > $OUT/mfma_rate.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest exit: $?" >> $OUT/pytest_gpu.log
cp gpurun_out/parity_report.json $OUT/ 2>/dev/null
# PMC passes FIRST, so that the bench line below reads the traffic figures of THIS build (bench.py: pmc_traffic)
BENCH="python $ROOTDIR/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-events --streams 1 --no-batch1"
run_pmc () { name=$1; shift
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o pmc -- $BENCH > $OUT/$name.log 2>&1)
  find $OUT/$name -type f -size +8M -delete 2>/dev/null
}
run_pmc pmc_fetch FETCH_SIZE
run_pmc pmc_write WRITE_SIZE
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o stats -- python $ROOTDIR/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-events --streams 1 --no-batch1 > $OUT/prof_bench.json 2> $OUT/prof.err)
find $OUT/prof -name "*.db" -delete 2>/dev/null; find $OUT/prof -type f -size +4M -delete 2>/dev/null
python tools/summarize_profile.py $OUT ${PREFIX:-r02} > $OUT/summarize.log 2>&1
timeout 900 python bench.py --steps 20 --warmup 5 --stock-pytorch > $OUT/bench.json 2> $OUT/bench.err
echo "bench exit: $?" >> $OUT/bench.err
for cfg in dtu_640x512_v3_gwc8 dtu_1152x864_v5_var blended_768x576_v7_var; do
  timeout 400 python bench.py --config $cfg --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_$cfg.json 2>> $OUT/bench.err
done
for cfg in dtu_1152x864_v5_var blended_768x576_v7_var; do
  timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29641 bench.py --gpus 1 \
     --mode view_sharded --config $cfg --batch 1 --steps 10 --warmup 3 --no-cpu-baseline --no-events 2>> $OUT/bench.err | grep "^{" > $OUT/bench_viewsharded_$cfg.json
done
timeout 300 python tools/gpu_costvol_probe.py 512 640 3 1 2>/dev/null > $OUT/costvol_probe_b1.txt
timeout 300 python tools/gpu_costvol_probe.py 512 640 3 2 2>/dev/null > $OUT/costvol_probe_b2.txt
for p in lds_probe valu_probe clock_probe store_probe; do timeout 120 tools/probes/bin/$p > $OUT/$p.txt 2>/dev/null; done
timeout 200 python tools/gpu_streams_probe.py 2>/dev/null > $OUT/streams_probe.txt
# cost-volume kernels as they run inside the forward (caches dirtied by another writer between launches), and a training step
CV_PROBE_DIRTY=512 CV_PROBE_IMPLS=gather,lds timeout 300 python tools/gpu_costvol_probe.py 512 640 3 2 2>/dev/null > $OUT/costvol_probe_b2_dirty.txt
timeout 300 python tools/gpu_train_step.py 2>/dev/null | grep -E "train step|eval forward" > $OUT/train_step.txt
cat $OUT/mfma_rate.txt; tail -4 $OUT/pytest_gpu.log; python tools/show_bench.py $OUT/bench.json | head -8; tail -3 $OUT/bench.err; cat $OUT/summarize.log | tail -3
for f in $OUT/bench_*.json; do python -c "
import json,sys
j=json.load(open('$f')); print('$(basename $f)', round(j['value'],1), j['unit'], j['config']['launch'][:40], j.get('single_stream',{}).get('value'), j.get('batch1',{}).get('value'))" 2>/dev/null; done
