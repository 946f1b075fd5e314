#!/bin/bash
# PMC passes over single CostRegNet layers (tools/gpu_layer_probe.py): HBM-side traffic and SQ counters of the kernels
# named by PMC_FILTER (regex on the kernel name; default: the `prob` head).
TAG=${1:-layerpmc}
ROOTDIR=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOTDIR/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOTDIR
export TMPDIR=/tmp
export LAYER_PROBE_ITEMS=${LAYER_PROBE_ITEMS:-prob}
export LAYER_PROBE_REPS=2
CMD=${PMC_CMD:-"python $ROOTDIR/tools/gpu_layer_probe.py 512 640 ${PMC_BATCH:-2}"}   # PMC_CMD: any other command (e.g. tools/gpu_conv0_probe.py)
run_pmc () { name=$1; shift
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o pmc -- $CMD > $OUT/$name.log 2>&1)
  find $OUT/$name -type f -size +8M -delete 2>/dev/null
}
run_pmc fetch FETCH_SIZE
run_pmc write WRITE_SIZE
run_pmc p1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS
run_pmc p2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA
run_pmc p3 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_SMEM SQ_WAVES SQ_INST_CYCLES_SALU
run_pmc p4 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum TCP_TCC_READ_REQ_sum
run_pmc p5 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum
PMC_BY_GRID=1 PMC_ALL=1 PMC_FILTER=${PMC_FILTER:-prob_zwalk} python $ROOTDIR/tools/summarize_pmc.py $OUT fetch write p1 p2 p3 p4 p5 | tee $OUT/summary.txt
