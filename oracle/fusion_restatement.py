"""ORACLE - TEST INFRASTRUCTURE ONLY.  Never imported by the product package `casmvsnet_pl_amd`.

CPU (numpy) restatement of the reference's depth filtering / fusion step for ONE reference view:
`eval.py:113-182` (xy_ref2src, xy_src2ref, check_geo_consistency) and `eval.py:273-318` (confidence mask,
geometric mask, depth / colour averaging, back-projection to world points), function by function.

Pinning: PARITY UNPINNED for the two OpenCV calls.  The reference's arithmetic lives in numba-compiled numpy code
(restated here with plain numpy in float32, the dtypes numba infers from the float32 inputs) and in two OpenCV
functions - `cv2.remap(..., INTER_LINEAR)` and `cv2.resize(..., fx=4, fy=4, INTER_LINEAR)` - and neither cv2 nor
numba is installed in the build container or on the GPU box (opencv-python / numba are pip dependencies of
the reference, absent offline), so the reference itself cannot be executed for this step.  OpenCV's published
algorithm (modules/imgproc/src/imgwarp.cpp `remap` / `remapBilinear`, resize.cpp `resizeGeneric_` - OpenCV 4.x) is
restated from its documentation and source layout as recalled:
  * remap with float maps converts every coordinate to fixed point with INTER_BITS = 5 fractional bits
    (sx = cvRound(x * 32), integer part sx >> 5 saturated to int16, fraction sx & 31) and takes the four weights from
    a 32 x 32 table: float weights (1-fy)(1-fx), (1-fy)fx, fy(1-fx), fy fx for CV_32F sources; 15-bit fixed-point
    weights (rounded, largest weight adjusted so that the four sum to 32768) and a rounding shift for CV_8U sources;
    BORDER_CONSTANT 0 for taps outside the image;
  * resize INTER_LINEAR samples at (dst + 0.5) / scale - 0.5 with the taps clamped to the image, horizontal pass then
    vertical pass in float.
Where numba's `fastmath` / BLAS leave the order of a float32 sum open (the 3x4 matrix products), this file fixes one
order - ((m0*X + m1*Y) + m2*Z) + m3, each product and sum rounded to float32 - and the HIP kernel follows it.
"""
import numpy as np

INTER_BITS = 5
INTER_TAB_SIZE = 1 << INTER_BITS
REMAP_COEF_BITS = 15
REMAP_COEF_SCALE = 1 << REMAP_COEF_BITS
f32 = np.float32


def relative_transform(P_to, P_from):
    """(P_to @ inv(P_from))[:3] in float32 (eval.py:120 / :135: `np.linalg.inv` of a float32 matrix)."""
    return (P_to.astype(f32) @ np.ascontiguousarray(np.linalg.inv(P_from.astype(f32))))[:3].astype(f32)


def _project(M, X, Y, Z):
    """rows of M (3,4) applied to (X, Y, Z, 1): ((m0*X + m1*Y) + m2*Z) + m3 in float32."""
    return [((M[i, 0] * X + M[i, 1] * Y) + M[i, 2] * Z) + M[i, 3] for i in range(3)]


def xy_ref2src(xy_ref, depth_ref, M_ref2src):
    """eval.py:113-127.  xy_ref (2,H,W) float32, depth_ref (H,W) -> xy_src (2,H,W)."""
    X, Y, Z = xy_ref[0] * depth_ref, xy_ref[1] * depth_ref, depth_ref          # :117  (x, y, 1) * depth
    q = _project(M_ref2src, X, Y, Z)                                           # :122
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.stack([q[0] / q[2], q[1] / q[2]]).astype(f32)                # :123


def _fixed_point_coords(map_x, map_y):
    """cv::remap's conversion of float maps: 5 fractional bits, integer part saturated to int16."""
    with np.errstate(invalid="ignore"):
        sx = np.rint(map_x.astype(np.float64) * INTER_TAB_SIZE)   # cvRound: round half to even
        sy = np.rint(map_y.astype(np.float64) * INTER_TAB_SIZE)
    # cvRound of NaN / out-of-int-range is INT_MIN on x86 (cvtsd2si)
    sx = np.where(np.isfinite(sx) & (np.abs(sx) < 2 ** 31), sx, -2.0 ** 31).astype(np.int64)
    sy = np.where(np.isfinite(sy) & (np.abs(sy) < 2 ** 31), sy, -2.0 ** 31).astype(np.int64)
    ix = np.clip(sx >> INTER_BITS, -32768, 32767)
    iy = np.clip(sy >> INTER_BITS, -32768, 32767)
    return ix, iy, (sx & (INTER_TAB_SIZE - 1)), (sy & (INTER_TAB_SIZE - 1))


def _taps(src, ix, iy):
    """The 2x2 neighbourhood with BORDER_CONSTANT 0: four arrays shaped like ix."""
    H, W = src.shape[:2]
    out = []
    for dy in (0, 1):
        for dx in (0, 1):
            x, y = ix + dx, iy + dy
            ok = (x >= 0) & (x < W) & (y >= 0) & (y < H)
            v = src[np.clip(y, 0, H - 1), np.clip(x, 0, W - 1)]
            out.append(np.where(ok[..., None] if src.ndim == 3 else ok, v, 0))
    return out


def remap_linear_f32(src, map_x, map_y):
    """cv2.remap(src float32 (H,W), map_x, map_y, INTER_LINEAR), borderMode constant 0  (eval.py:159-162)."""
    ix, iy, fx, fy = _fixed_point_coords(map_x, map_y)
    ax, ay = (fx.astype(f32) / f32(INTER_TAB_SIZE)), (fy.astype(f32) / f32(INTER_TAB_SIZE))
    w = [(f32(1) - ay) * (f32(1) - ax), (f32(1) - ay) * ax, ay * (f32(1) - ax), ay * ax]   # BilinearTab_f = vy[k1] * vx[k2]
    t = _taps(src.astype(f32), ix, iy)
    return (((t[0] * w[0] + t[1] * w[1]) + t[2] * w[2]) + t[3] * w[3]).astype(f32)


def _int_weights(fx, fy):
    """BilinearTab_i: the float weights scaled by 2^15 and rounded to int16, the largest one adjusted so that the
    four sum to 32768 (initInterTab2D's fix-up for a 2x2 kernel)."""
    ax, ay = fx.astype(f32) / f32(INTER_TAB_SIZE), fy.astype(f32) / f32(INTER_TAB_SIZE)
    wf = np.stack([(f32(1) - ay) * (f32(1) - ax), (f32(1) - ay) * ax, ay * (f32(1) - ax), ay * ax])
    wi = np.rint(wf.astype(np.float64) * REMAP_COEF_SCALE).astype(np.int64)
    diff = wi.sum(0) - REMAP_COEF_SCALE
    k = np.where(diff < 0, wi.argmax(0), wi.argmin(0))   # diff < 0: raise the largest; diff > 0: lower the smallest
    np.put_along_axis(wi, k[None], np.take_along_axis(wi, k[None], 0) - diff[None], 0)
    return wi


def remap_linear_u8(src, map_x, map_y):
    """cv2.remap(src uint8 (H,W,3), ...): fixed-point weights, (sum + 2^14) >> 15  (eval.py:164-167)."""
    ix, iy, fx, fy = _fixed_point_coords(map_x, map_y)
    wi = _int_weights(fx, fy)
    t = _taps(src.astype(np.int64), ix, iy)
    acc = sum(t[k] * wi[k][..., None] for k in range(4))
    return np.clip((acc + (1 << (REMAP_COEF_BITS - 1))) >> REMAP_COEF_BITS, 0, 255).astype(np.uint8)


def xy_src2ref(xy_ref, xy_src, depth_ref, depth_src2ref, M_src2ref):
    """eval.py:130-153 -> depth_ref_reproj (H,W) float32, mask_geo (H,W) bool."""
    X, Y, Z = xy_src[0] * depth_src2ref, xy_src[1] * depth_src2ref, depth_src2ref   # :134
    r = _project(M_src2ref, X, Y, Z)                                                # :137
    depth_ref_reproj = r[2].astype(f32)                                             # :138
    with np.errstate(divide="ignore", invalid="ignore"):
        dx = (r[0] / r[2]).astype(f32) - xy_ref[0]                                  # :139, :143
        dy = (r[1] / r[2]).astype(f32) - xy_ref[1]
        mask_pixel = (dx * dx + dy * dy) < f32(1)                                   # :144
        mask_depth = np.abs((depth_ref_reproj - depth_ref) / depth_ref) < f32(0.01)  # :147
    return depth_ref_reproj, mask_pixel & mask_depth


def check_geo_consistency(depth_ref, P_world2ref, depth_src, P_world2src, image_src):
    """eval.py:156-182 -> depth_ref_reproj (masked), mask_geo, image_src2ref (masked)."""
    H, W = depth_ref.shape
    xy_ref = np.mgrid[:H, :W][::-1].astype(f32)                                     # :164
    xy_src = xy_ref2src(xy_ref, depth_ref.astype(f32), relative_transform(P_world2src, P_world2ref))
    depth_src2ref = remap_linear_f32(depth_src, xy_src[0], xy_src[1])               # :169-172
    image_src2ref = remap_linear_u8(image_src, xy_src[0], xy_src[1])                # :174-177
    depth_ref_reproj, mask_geo = xy_src2ref(xy_ref, xy_src, depth_ref.astype(f32), depth_src2ref,
                                            relative_transform(P_world2ref, P_world2src))
    depth_ref_reproj = np.where(mask_geo, depth_ref_reproj, f32(0))                 # :183
    image_src2ref = np.where(mask_geo[..., None], image_src2ref, 0).astype(np.uint8)  # :184
    return depth_ref_reproj, mask_geo, image_src2ref


def resize_linear_x4(src):
    """cv2.resize(src float32 (h,w), None, fx=4, fy=4, INTER_LINEAR)  (eval.py:281-282)."""
    h, w = src.shape

    def coeffs(n_dst, n_src):
        f = (np.arange(n_dst, dtype=np.float64) + 0.5) * 0.25 - 0.5   # (dst + 0.5) * scale - 0.5, scale = 1 / 4 in double
        s = np.floor(f).astype(np.int64)
        a = (f - s).astype(f32)
        lo = s < 0
        s, a = np.where(lo, 0, s), np.where(lo, f32(0), a)
        hi = s >= n_src - 1
        s, a = np.where(hi, n_src - 1, s), np.where(hi, f32(0), a)
        return s, np.minimum(s + 1, n_src - 1), (f32(1) - a).astype(f32), a.astype(f32)
    x0, x1, a0, a1 = coeffs(4 * w, w)
    y0, y1, b0, b1 = coeffs(4 * h, h)
    rows = (src[:, x0] * a0 + src[:, x1] * a1).astype(f32)                          # hresize
    return (rows[y0] * b0[:, None] + rows[y1] * b1[:, None]).astype(f32)            # vresize


def fuse_reference_view(depth_ref, image_ref, proba_ref_quarter, P_world2ref, depth_srcs, image_srcs, P_world2srcs,
                        conf=0.999, min_geo_consistent=5):
    """eval.py:273-318 for one reference view -> dict(depth_refined float32 (H,W), image_refined float64 (H,W,3),
    mask_geo_sum int64, mask_final bool, xyz_world float32 (H,W,3) for EVERY pixel (the reference keeps mask_final's)."""
    H, W = depth_ref.shape
    mask_conf = resize_linear_x4(proba_ref_quarter.astype(f32)) > conf              # :281-283
    reprojs, images, masks = [depth_ref.astype(f32)], [image_ref], []
    for depth_src, image_src, P_src in zip(depth_srcs, image_srcs, P_world2srcs):   # :291-309
        d, m, im = check_geo_consistency(depth_ref, P_world2ref, depth_src.astype(f32), P_src, image_src)
        reprojs.append(d)
        images.append(im)
        masks.append(m)
    mask_geo_sum = np.sum(masks, 0)                                                 # :310
    mask_geo_final = mask_geo_sum >= min_geo_consistent                             # :311
    depth_refined = (np.sum(reprojs, 0) / (mask_geo_sum + 1)).astype(f32)           # :312-313 (float64 quotient)
    image_refined = np.sum(images, 0) / np.expand_dims(mask_geo_sum + 1, -1)        # :314-315 (float64)
    mask_final = mask_conf & mask_geo_final                                         # :319
    xy = np.mgrid[:H, :W][::-1]                                                     # :322 (int64)
    X, Y, Z = xy[0] * depth_refined, xy[1] * depth_refined, 1 * depth_refined       # :323 int64 * float32 -> float64
    Minv = np.linalg.inv(P_world2ref.astype(f32)).astype(np.float64)               # :326 float32 inverse, float64 product
    xyz_world = np.stack([((Minv[i, 0] * X + Minv[i, 1] * Y) + Minv[i, 2] * Z) + Minv[i, 3] for i in range(3)], -1)
    return dict(depth_refined=depth_refined, image_refined=image_refined, mask_geo_sum=mask_geo_sum,
                mask_final=mask_final, xyz_world=xyz_world.astype(f32))


def fuse_scan(views, metas, conf=0.999, min_geo_consistent=5, skip=1):
    """eval.py:255-326 (the scan loop) in memory, numpy: refined depth / 8-bit refined image of an already processed view
    replace its prediction when it is used again (eval.py:263-265, :284-293; the refined image goes through cv2.imwrite /
    cv2.imread = an 8-bit round trip with round-half-even), masked world points and truncated colours are collected
    (:313-321, :338).  views: vid -> dict(depth, image, proba, P)."""
    depth_refined, image_refined, vs, v_colors = {}, {}, [], []

    def current(vid):
        if vid in image_refined:
            return depth_refined[vid], image_refined[vid]
        return views[vid]["depth"], views[vid]["image"]

    for ref_vid, src_vids in metas:
        if ref_vid not in views or any(s not in views for s in src_vids):
            continue
        depth_ref, image_ref = current(ref_vid)
        srcs = [current(s) for s in src_vids]
        for s, (d, _) in zip(src_vids, srcs):
            depth_refined.setdefault(s, d)
        r = fuse_reference_view(depth_ref, image_ref, views[ref_vid]["proba"], views[ref_vid]["P"], [d for d, _ in srcs],
                                [i for _, i in srcs], [views[s]["P"] for s in src_vids], conf=conf, min_geo_consistent=min_geo_consistent)
        depth_refined[ref_vid] = r["depth_refined"]
        image_refined[ref_vid] = np.clip(np.rint(r["image_refined"]), 0, 255).astype(np.uint8)
        m = r["mask_final"]
        vs.append(r["xyz_world"][m][::skip])
        v_colors.append(r["image_refined"][m][::skip])
    if not vs:
        return np.zeros((0, 3), f32), np.zeros((0, 3), np.uint8), depth_refined
    return np.vstack(vs).astype(f32), np.vstack(v_colors).astype(np.uint8), depth_refined
