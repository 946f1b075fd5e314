"""TEST INFRASTRUCTURE ONLY - writes tests/golden/fusion_*.npz by RUNNING the unmodified reference.

    python oracle/make_fusion_golden.py        (build container only: needs /root/reference)

What is executed: `check_geo_consistency`, `xy_ref2src`, `xy_src2ref` of the unmodified /root/reference/eval.py (:113-182),
imported as a module through oracle/reference_loader.load_reference_eval():
  * `numba.jit` is the identity (oracle/shims/numba) - the decorated functions run as the numpy code they are written in
    (numba's fastmath only frees the order of float32 sums; here numpy / BLAS fix one);
  * `cv2.remap` / `cv2.resize` are oracle/fusion_restatement.py's restatement of OpenCV (oracle/shims/cv2): opencv-python is
    not installable offline.
So these fixtures PIN the reference's own arithmetic of the fusion step - projection both ways, the pixel (< 1 px) and
relative-depth (< 1 %) tests, the masking - and leave exactly two things unpinned: OpenCV's `remap` and `resize`
(INTER_LINEAR) themselves.  The scan loop of eval.py:245-353 lives in the script's `__main__` block and cannot be imported;
its sums / divisions are restated in fusion_restatement.fuse_reference_view and are not covered by a fixture.
Inputs are NOT stored: they are regenerated from the seed by oracle/fusion_scene.py and guarded by a checksum.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import fusion_scene  # noqa: E402
from oracle.reference_loader import load_reference_eval  # noqa: E402

CASES = {"fusion_48x64_s3": dict(H=48, W=64, S=3, seed=11), "fusion_64x96_s4": dict(H=64, W=96, S=4, seed=12)}


def run_reference(case):
    ev = load_reference_eval()
    Ps, depths, images, proba = fusion_scene.scene(**case)
    H, W = depths[0].shape
    out = {"meta_hws_seed": np.array([case["H"], case["W"], case["S"], case["seed"]]),
           "chk_inputs": np.array(fusion_scene.checksum(Ps + depths + images + [proba]))}
    xy_ref = np.mgrid[:H, :W][::-1].astype(np.float32)                                   # eval.py:164
    for s in range(1, len(Ps)):
        out[f"xy_src_{s}"] = ev.xy_ref2src(xy_ref, depths[0], Ps[0], depths[s], Ps[s], (W, H)).astype(np.float32)
        d, m, im = ev.check_geo_consistency(depths[0], Ps[0], depths[s], Ps[s], images[0], images[s], (W, H))
        out[f"depth_ref_reproj_{s}"], out[f"mask_geo_{s}"], out[f"image_src2ref_{s}"] = d.astype(np.float32), m, im
    return out


if __name__ == "__main__":
    for name, case in CASES.items():
        out = run_reference(case)
        path = os.path.join(ROOT, "tests", "golden", name + ".npz")
        np.savez_compressed(path, **out)
        print(path, {k: (v.shape, str(v.dtype)) for k, v in out.items() if k.startswith(("depth", "mask", "xy"))}.__len__(), "arrays;",
              "consistent fraction", float(np.mean([out[k].mean() for k in out if k.startswith("mask_geo")])))
