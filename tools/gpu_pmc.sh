#!/bin/bash
# PMC counter passes (separate from kernel-trace/stats, as gpurun requires) + MFMA rate probe.
TAG=${1:-pmc}
ROOTDIR=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOTDIR/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOTDIR
export TMPDIR=/tmp
(cd /tmp && rocprofv3 -L > $OUT/counters_list.txt 2>&1)
BENCH="python $ROOTDIR/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-events"
run_pmc () { # name counters...
  name=$1; shift
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o pmc -- $BENCH > $OUT/$name.log 2>&1)
  find $OUT/$name -type f -size +8M -delete 2>/dev/null
}
run_pmc p1 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS
run_pmc p2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_WAVES
run_pmc p3 FETCH_SIZE GRBM_GUI_ACTIVE
run_pmc p4 WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
ls -R $OUT | head -40; tail -3 $OUT/p1.log
