"""TEST INFRASTRUCTURE ONLY - import shim so that the *unmodified* reference files that `import cv2`
(/root/reference/eval.py:3, datasets/{dtu,blendedmvs,tanks}.py, utils/visualization.py) can be imported and RUN in the build
container, where opencv-python is not installed (no network).  Never imported by the product package.

Only the calls those files make on the paths the tests exercise are provided:
  * resize(..., interpolation=INTER_NEAREST)  - datasets' ground-truth depth / mask pyramids (dtu.py:95-124,
    blendedmvs.py:110-116): OpenCV's resizeNN index rule restated here: dst size = cvRound(src * f) when only fx / fy are
    given, source index = min(floor(dst_index * (1 / scale)), src - 1) with scale = fx (or dsize / ssize), in double;
  * resize(float32, None, fx=4, fy=4, INTER_LINEAR) and remap(..., INTER_LINEAR) - eval.py:159-167,281-282: these two are
    NOT OpenCV: they forward to oracle/fusion_restatement.py (this repo's restatement of OpenCV's published algorithm).
    A fixture generated through this shim therefore pins the reference's OWN arithmetic around those calls (the numpy code
    of xy_ref2src / xy_src2ref / check_geo_consistency), and leaves the two OpenCV functions themselves unpinned;
  * imread(path, 0) for single-channel PNGs (dtu.py:113).
"""
import numpy as np

INTER_NEAREST = 0
INTER_LINEAR = 1
IMREAD_GRAYSCALE = 0
COLORMAP_JET, COLORMAP_BONE = 2, 1   # default arguments of utils/visualization.py:7,21 (never called by the tests)


def _nearest_index(n_dst, n_src, scale):
    idx = np.floor(np.arange(n_dst, dtype=np.float64) * (1.0 / scale)).astype(np.int64)
    return np.minimum(idx, n_src - 1)


def resize(src, dsize, fx=0.0, fy=0.0, interpolation=INTER_LINEAR):
    src = np.asarray(src)
    h, w = src.shape[:2]
    if dsize is None or tuple(dsize) == (0, 0):
        ow, oh = int(np.rint(w * fx)), int(np.rint(h * fy))     # saturate_cast<int>(ssize * f): cvRound
        sx, sy = float(fx), float(fy)
    else:
        ow, oh = int(dsize[0]), int(dsize[1])
        sx, sy = ow / w, oh / h
    if interpolation == INTER_NEAREST:
        return src[_nearest_index(oh, h, sy)][:, _nearest_index(ow, w, sx)].copy()
    if interpolation == INTER_LINEAR and src.dtype == np.float32 and src.ndim == 2 and (ow, oh) == (4 * w, 4 * h):
        from oracle.fusion_restatement import resize_linear_x4
        return resize_linear_x4(src)
    raise NotImplementedError(f"cv2 shim: resize {src.dtype} {src.shape} -> {(ow, oh)} interpolation={interpolation}")


def remap(src, map1, map2, interpolation=INTER_LINEAR, **kwargs):
    if interpolation != INTER_LINEAR or kwargs:
        raise NotImplementedError("cv2 shim: remap supports INTER_LINEAR with the default constant-0 border only")
    from oracle import fusion_restatement as FR
    src = np.asarray(src)
    if src.dtype == np.float32 and src.ndim == 2:
        return FR.remap_linear_f32(src, np.asarray(map1), np.asarray(map2))
    if src.dtype == np.uint8 and src.ndim == 3:
        return FR.remap_linear_u8(src, np.asarray(map1), np.asarray(map2))
    raise NotImplementedError(f"cv2 shim: remap of {src.dtype} {src.shape}")


def imread(filename, flags=1):
    from PIL import Image
    im = Image.open(filename)
    if flags == 0:
        if im.mode != "L":
            raise NotImplementedError("cv2 shim: imread(.., 0) of a non-grayscale file (OpenCV's fixed-point BGR2GRAY is not restated)")
        return np.asarray(im).copy()
    return np.asarray(im.convert("RGB"))[:, :, ::-1].copy()   # BGR like OpenCV


def imwrite(filename, img):
    from PIL import Image
    a = np.asarray(img)
    if a.dtype != np.uint8:
        a = np.clip(np.rint(a), 0, 255).astype(np.uint8)      # saturate_cast<uchar>: round half to even
    Image.fromarray(a[:, :, ::-1] if a.ndim == 3 else a).save(filename)
    return True
