#!/bin/bash
# round 4, call N: SQ counters of the fused tail and of conv0 (two PMC passes each over the torch-free check binaries)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4N
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
            "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES"; do
  n=$(echo $pass | cut -d' ' -f1)
  timeout 120 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $O/tail_$n -o pmc -- $R/tools/probes/bin/conv11_prob_check 8 > $O/tail_$n.log 2>&1; echo "tail $n: exit $?"
  timeout 120 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $O/conv0_$n -o pmc -- $R/tools/probes/bin/conv0_zm_check 8 > $O/conv0_$n.log 2>&1; echo "conv0 $n: exit $?"
done
find $O -type f -size +6M -delete
ls $O | head -20; tail -3 $O/tail_SQ_WAVE_CYCLES.log
