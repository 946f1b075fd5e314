#!/bin/bash
# concurrent forwards with the split-f16 layers on 2 / 3 / 4 streams: bit equality with the single-stream forward, and the rate
timeout 400 python tools/gpu_mixed_streams.py 2 1 200 2>&1 | grep -v Warning
timeout 300 python tools/gpu_mixed_streams.py 3 1 60 2>&1 | grep -v Warning
timeout 300 python tools/gpu_mixed_streams.py 2 4 60 2>&1 | grep -v Warning
timeout 300 python tools/gpu_mixed_streams.py 4 2 60 2>&1 | grep -v Warning
