#!/bin/bash
# round 4, call O: SQ counters of every kernel of the benched step (two PMC passes over the torch-free step runner)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4O
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
            "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAVES SQ_VALU_MFMA_BUSY_CYCLES"; do
  n=$(echo $pass | cut -d' ' -f1)
  timeout 120 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $O/step_$n -o pmc -- python $R/tools/notorch/step_runner.py --batch 8 --steps 3 --warmup 1 > $O/step_$n.log 2>&1; echo "step $n: exit $?"
done
find $O -type f -size +8M -delete
ls $O
