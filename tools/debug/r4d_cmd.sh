echo "== production library: wide forms"
timeout 120 tools/probes/bin/coresidency_lib_victim 100 px,ciw,s2w
echo "== trace library, CASMVS_NO_DB=1 (PX single-buffered), CASMVS_DB_CI=0 (CI single-buffered)"
LD_PRELOAD=$PWD/casmvsnet_pl_amd/libcasmvs_trace.so CASMVS_NO_DB=1 CASMVS_DB_CI=0 timeout 120 tools/probes/bin/coresidency_lib_victim 100 px,ciw
echo "== trace library, defaults (control)"
LD_PRELOAD=$PWD/casmvsnet_pl_amd/libcasmvs_trace.so timeout 120 tools/probes/bin/coresidency_lib_victim 100 px
