#!/bin/bash
# Round-3 evidence run for profiles/ (one gpurun call): parity tests, PMC passes (FETCH_SIZE, WRITE_SIZE: separate passes, no tracing
# domains beside --kernel-trace), rocprofv3 kernel stats, the bench lines (default = single stream, batch 8; the other BASELINE configs;
# view-sharded at world size 1; --mode train), files -> depth maps, the layer probes.
TAG=${1:-final3}
PREFIX=${PREFIX:-r03}
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
timeout 300 python __graft_entry__.py --smoke > $OUT/smoke.log 2>&1; echo "smoke exit: $?" >> $OUT/smoke.log
# PMC passes FIRST, so that the bench line below reads the traffic figures of THIS build (bench.py: pmc_traffic)
export PMC_BATCH=${BENCH_BATCH:-8}
export PMC_CMD_NOTE="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-events --no-batch1 --batch $PMC_BATCH"
export PMC_DATE=$(date -u +%Y-%m-%dT%H:%MZ)
BENCH="python $ROOTDIR/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-events --no-batch1 --batch $PMC_BATCH"
run_pmc () { name=$1; shift
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o pmc -- $BENCH > $OUT/$name.log 2>&1)
  find $OUT/$name -type f -size +8M -delete 2>/dev/null
}
run_pmc pmc_fetch FETCH_SIZE
run_pmc pmc_write WRITE_SIZE
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o stats -- python $ROOTDIR/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-events --no-batch1 --batch $PMC_BATCH > $OUT/prof_bench.json 2> $OUT/prof.err)
find $OUT/prof -name "*.db" -delete 2>/dev/null; find $OUT/prof -type f -size +4M -delete 2>/dev/null
python tools/summarize_profile.py $OUT $PREFIX > $OUT/summarize.log 2>&1
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
echo "bench exit: $?" >> $OUT/bench.err
cp $OUT/bench.json profiles/${PREFIX}_bench.json 2>/dev/null
for cfg in dtu_640x512_v3_gwc8 dtu_1152x864_v5_var blended_768x576_v7_var; do
  timeout 400 python bench.py --config $cfg --steps 10 --warmup 3 --no-cpu-baseline --batch 4 > $OUT/bench_$cfg.json 2>> $OUT/bench.err
done
for cfg in dtu_1152x864_v5_var blended_768x576_v7_var; do
  timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29641 bench.py --gpus 1 \
     --mode view_sharded --config $cfg --batch 1 --steps 10 --warmup 3 --no-cpu-baseline --no-events 2>> $OUT/bench.err | grep "^{" > $OUT/bench_viewsharded_$cfg.json
done
timeout 400 python bench.py --mode train --steps 10 --warmup 3 > $OUT/bench_train.json 2>> $OUT/bench.err
timeout 300 python tools/gpu_files_throughput.py 49 64 128 2>/dev/null > $OUT/files_throughput.txt
timeout 100 python tools/gpu_fpn_probe.py 2>/dev/null | grep "^N" > $OUT/fpn_probe.txt
timeout 200 python tools/gpu_conv0_probe.py 2>/dev/null | grep -v amdgpu > $OUT/conv0_probe.txt
LAYER_PROBE_ITEMS=bottom,deconv timeout 200 python tools/gpu_layer_probe.py 2>/dev/null | grep -v amdgpu > $OUT/layer_probe.txt
for d in k0:1024 k1:1024 none; do DISTURB=$d timeout 100 python tools/debug/disturber.py 2>&1 | grep "^disturber" >> $OUT/mfma_coresidency_recheck.txt; done
cat $OUT/mfma_rate.txt; tail -4 $OUT/pytest_gpu.log; tail -3 $OUT/smoke.log; python tools/show_bench.py $OUT/bench.json | head -8; tail -3 $OUT/bench.err; cat $OUT/summarize.log | tail -3
for f in $OUT/bench_*.json; do python -c "
import json,sys
j=json.load(open('$f')); print('$(basename $f)', round(j['value'],1), j['unit'], str(j['config'].get('launch'))[:40], j.get('two_streams_float32',{}).get('value'), j.get('batch1',{}).get('value'))" 2>/dev/null; done
cat $OUT/files_throughput.txt | tail -6; cat $OUT/mfma_coresidency_recheck.txt
