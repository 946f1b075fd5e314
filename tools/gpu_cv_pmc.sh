#!/bin/bash
# PMC passes over the cost-volume kernels (tools/gpu_costvol_probe.py, 3 levels x {nchw, nhwc}, smooth depth).
TAG=${1:-cvpmc}
ROOTDIR=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOTDIR/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOTDIR
export TMPDIR=/tmp
export CV_PROBE_QUICK=1
CMD="python $ROOTDIR/tools/gpu_costvol_probe.py"
run_pmc () { name=$1; shift
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o pmc -- $CMD > $OUT/$name.log 2>&1)
  find $OUT/$name -type f -size +8M -delete 2>/dev/null
}
run_pmc p1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS
run_pmc p2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INST_LEVEL_VMEM
run_pmc p3 TA_TA_BUSY_sum TA_BUFFER_TOTAL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_WAVEFRONTS_sum TA_FLAT_WAVEFRONTS_sum
run_pmc p4 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
run_pmc p5 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCP_TCC_READ_REQ_LATENCY_sum
run_pmc p6 GRBM_GUI_ACTIVE TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_LATENCY_sum MemUnitStalled
ls $OUT; tail -3 $OUT/p1.log
