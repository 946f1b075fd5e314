#!/bin/bash
# the library assembled with the packed-float32 sources exchanged (casmvsnet_pl_amd/build.py rewrite_unsafe_packed): every float32 victim form beside every neighbour
timeout 500 tools/probes/bin/coresidency_lib_victim ${R5V_ROUNDS:-500} px,ci,ciw,s2,t2 2>&1 | grep -E "beside|REPRODUCED|not reproduced"
