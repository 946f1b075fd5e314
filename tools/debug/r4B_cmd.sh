#!/bin/bash
# round 4, call B: first GPU run of conv11_prob_zfused_kernel against deconv11_sf + prob_zwalk
timeout 200 tools/probes/bin/conv11_prob_check 8; echo "-- exit $?"
timeout 100 tools/probes/bin/conv11_prob_check 1 | grep "B=1 in [0-9]*x[0-9][0-9]*x[0-9][0-9][0-9]\|B=1 in 24\|FAIL"
