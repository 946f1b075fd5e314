"""tools/notorch/step_runner.py drives the whole forward through the C ABI without torch.  On a machine without a GPU its dry
run (HIPMINI_FAKE=1: "device" arrays in host memory) must get through every host-side call - weight packing of all 13 + 3 x 11
layers and their split-f16 images, workspace queries, the camera rig - and stop at the first kernel launch with the library's
own "no device" error: the script stays in step with the library's signatures.  CPU only (skipped where a GPU exists: host
pointers must not reach a kernel)."""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(torch.cuda.is_available(), reason="dry run is for machines without a GPU")
def test_step_runner_dry_run_reaches_the_first_launch():
    env = dict(os.environ, HIPMINI_FAKE="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "notorch", "step_runner.py"), "--batch", "1", "--hw", "64", "96"],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0
    assert "featurenet" in out.stderr and "no ROCm-capable device" in out.stderr, out.stderr[-1500:]
    # the float32 A/B switch of the layers that have an f16 form
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "notorch", "step_runner.py"), "--batch", "1", "--hw", "64", "96",
                          "--f32-layers", "conv9,conv11"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "featurenet" in out.stderr and "no ROCm-capable device" in out.stderr, out.stderr[-1500:]


def test_step_runner_does_not_import_torch():
    import re
    for rel in ("tools/notorch/step_runner.py", "tools/notorch/hipmini.py", "casmvsnet_pl_amd/_lib.py"):   # the runner loads _lib.py as a stand-alone module
        src = open(os.path.join(ROOT, rel)).read()
        assert not re.search(r"^\s*(import|from)\s+torch\b", src, re.M), rel
