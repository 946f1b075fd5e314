#!/bin/bash
# PMC passes over the cost-volume kernels (tools/gpu_costvol_probe.py; CV_PROBE_IMPLS selects the kernels).
TAG=${1:-cvpmc}
ROOTDIR=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOTDIR/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOTDIR
export TMPDIR=/tmp
export CV_PROBE_IMPLS=${CV_PROBE_IMPLS:-gather,lds}
export CV_PROBE_REPS=1
CMD="python $ROOTDIR/tools/gpu_costvol_probe.py 512 640 3 1"
run_pmc () { name=$1; shift
  (cd /tmp && timeout 120 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o pmc -- $CMD > $OUT/$name.log 2>&1)
  find $OUT/$name -type f -size +8M -delete 2>/dev/null
}
run_pmc p1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS
run_pmc p2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA
run_pmc p3 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_SMEM SQ_WAVES SQ_INST_CYCLES_SALU
# (TA / TCP / TCC counters: more than 2 per block and pass are rejected by the hardware and the profiled
#  process then hangs until the timeout - do not add them here without checking the block limits)
python $ROOTDIR/tools/summarize_pmc.py $OUT p1 p2 p3 | tee $OUT/summary.txt
