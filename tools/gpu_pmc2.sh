#!/bin/bash
# LDS / instruction-fetch counters of the MFMA loop (conv0 kernel), ABL=2 and normal.
# (historic: the ablation modes now exist only in the -DCASMVS_TRACE build, tools/build_trace_lib.sh; point
#  casmvsnet_pl_amd/_lib.LIB_PATH at libcasmvs_trace.so as tools/gpu_trace2.py does to reproduce)
TAG=${1:-pmc3}
ROOTDIR=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOTDIR/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOTDIR
export TMPDIR=/tmp
BENCH="python $ROOTDIR/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-events"
run_pmc () { name=$1; shift
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o pmc -- $BENCH > $OUT/$name.log 2>&1)
  find $OUT/$name -type f -size +8M -delete 2>/dev/null
}
export CASMVS_NO_DB=1
export CASMVS_ABLATE=2
run_pmc a1 SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_BANK_CONFLICT SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_UNALIGNED_STALL
run_pmc a2 SQ_IFETCH SQ_IFETCH_LEVEL SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY
run_pmc a3 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_CYCLES
ls $OUT
