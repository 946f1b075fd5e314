#!/bin/bash
# Round 3, call i: split-bf16 conv0: lane probe, layer parity, layer timing vs the float32-MFMA kernel.
TAG=${1:-r3i}
ROOTDIR=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOTDIR/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOTDIR
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 300 -p no:cacheprovider -k "mfma_bf16 or splitbf16" > $OUT/pytest_sel.log 2>&1
echo "pytest exit: $?" >> $OUT/pytest_sel.log
tail -15 $OUT/pytest_sel.log
timeout 300 python tools/gpu_conv0_probe.py > $OUT/conv0_probe.txt 2>&1
grep -v amdgpu.ids $OUT/conv0_probe.txt
