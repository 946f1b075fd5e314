#!/bin/bash
# round 4, call E: phase timeline of conv11_prob_zfused_kernel (all waves in step)
LD_PRELOAD=$PWD/casmvsnet_pl_amd/libcasmvs_zftrace.so timeout 60 tools/probes/bin/zf_trace 8
