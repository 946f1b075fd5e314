timeout 120 tools/probes/bin/coresidency_lib_victim 200
echo "== pytest non-finite"
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "non_finite" -p no:cacheprovider 2>&1 | tail -5
for cfg in "864 1152 5" "576 768 7"; do
  for b in 1 4; do
    echo "== costvol probe $cfg batch $b (dirty 512)"
    CV_PROBE_DIRTY=512 CV_PROBE_REPS=6 CV_PROBE_IMPLS=gather,lds,pairs timeout 300 python tools/gpu_costvol_probe.py $cfg $b 2>&1 | grep -v "homo_warp"
  done
done
echo "== gwc8 640x512 V3 batch 1 / 4"
for b in 1 4; do CV_PROBE_G=8 CV_PROBE_DIRTY=512 CV_PROBE_REPS=6 CV_PROBE_IMPLS=gather,lds timeout 300 python tools/gpu_costvol_probe.py 512 640 3 $b 2>&1; done
