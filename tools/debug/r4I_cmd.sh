#!/bin/bash
# round 4, call I: first GPU run of conv_s1z_sf_kernel (conv2 input-stationary along z) against the tile kernel
timeout 200 tools/probes/bin/conv_s2_check 8 | grep "s1 \|ALL\|FAIL"
timeout 200 tools/probes/bin/conv_s2_check 1 | grep "s1 B=1 [0-9]*x[0-9]*x[0-9][0-9][0-9]\|s1 B=1 24\|FAIL"
