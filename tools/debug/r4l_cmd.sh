timeout 120 tools/probes/bin/conv0_zm_check 8 | grep "cin=32\|ALL\|FAIL"
timeout 60 python tools/notorch/step_runner.py --batch 8 | tail -4
