#!/bin/bash
# co-residency experiments 11-14 on the PX kernel (csrc/conv3d_mfma.hip CASMVS_PX_EXP): what the wrong accumulator element holds and when it goes wrong
for v in ${R5Q_LIBS:-ab1 ab2 ab3 ab4}; do
  echo "== library $v"
  CORES_DUMP=${R5Q_DUMP:-2} LD_PRELOAD=$GRAFT_REPO_ROOT/casmvsnet_pl_amd/libcasmvs_$v.so timeout 120 tools/probes/bin/coresidency_lib_victim ${R5Q_ROUNDS:-60} px 2>&1 | grep -vE "^\s*$" | head -${R5Q_LINES:-70}
done
