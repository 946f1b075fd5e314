#!/bin/bash
# the stand-alone reproducer of the packed-multiply-add fault + the library kernel with (exp 15) no packed form / (exp 16) the op_sel form on every column tile
timeout 250 tools/probes/bin/pk_fma_opsel_repro ${R5R_ROUNDS:-30}
R5Q_LIBS="${R5R_LIBS:-ab1 ab2}" R5Q_ROUNDS=${R5R_LIB_ROUNDS:-100} R5Q_DUMP=1 R5Q_LINES=40 bash tools/debug/r5q_cmd.sh
