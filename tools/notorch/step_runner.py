"""The engine's whole forward (FeatureNet -> 3 x (hypotheses, cost volume, CostRegNet + regression)) through the C ABI, WITHOUT
torch: numpy + ctypes (tools/notorch/hipmini.py).  Same library calls in the same order as casmvsnet_pl_amd/mvsnet.py's eval path
(CascadeMVSNet.forward with the default split-f16 layer set), random weights of the model's shapes, the synthetic DTU-like rig of
casmvsnet_pl_amd/synthetic.py restated in numpy.  Prints ms per step, depth maps/s and the HIP-event time of every stage.

Why: a python + torch process needs 1-2 minutes to start on a fresh GPU box (most of what a short gpurun call is charged
for); this one starts in a second, so an A/B of two library builds on the WHOLE step - or a rocprofv3 / PMC pass over it - costs
~20 s of GPU time.   python tools/notorch/step_runner.py [--lib path/to/libcasmvs_x.so] [--batch 8] [--steps 10] [--warmup 3]
It is a measurement tool: parity is established by tests/ (torch, oracle), not here; it checks only that the depths are finite and
inside the hypothesis range."""
import argparse
import ctypes
import math
import os
import sys

import numpy as np

ap = argparse.ArgumentParser()
ap.add_argument("--lib", default=None, help="another build of libcasmvs_hip.so (tools/build_variant.py)")
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--views", type=int, default=3)
ap.add_argument("--hw", type=int, nargs=2, default=(512, 640))
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--warmup", type=int, default=3)
ap.add_argument("--all-f32", action="store_true", help="every layer on the float32 MFMA kernels (conv0_mode / ci_mode / tail_mode = f32)")
ap.add_argument("--f32-k5s2", action="store_true", help="A/B: FeatureNet's conv1.0 / conv2.0 (5 x 5 stride 2) on the float32 MFMA kernel")
ap.add_argument("--feature-split", type=int, default=1, help="A/B: FeatureNet over this many groups of images in sequence (cache blocking)")
ap.add_argument("--nchw-feats", action="store_true", help="A/B: FeatureNet also stores the (N, C, h, w) maps of levels 0 / 1 (nothing in the forward reads them; the engine's call drops them)")
ap.add_argument("--two-layer-conv0", action="store_true", help="A/B: FeatureNet's conv0.0 / conv0.1 as two float32-MFMA launches (rounds 1-5) instead of the fused f16 kernel")
ap.add_argument("--f32-layers", default="", help="A/B: comma list of CostRegNet layers kept on the float32 MFMA kernel although they have an f16 form: conv0, conv1, conv2, conv3, conv4, conv6, conv9, conv11")
args = ap.parse_args()
if args.lib:
    os.environ["CASMVS_LIB_PATH"] = os.path.abspath(args.lib)
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import importlib.util

# casmvsnet_pl_amd/__init__ imports torch; _lib.py itself needs only ctypes: load it as a stand-alone module
spec = importlib.util.spec_from_file_location("casmvs_lib", os.path.join(ROOT, "casmvsnet_pl_amd", "_lib.py"))
_lib = importlib.util.module_from_spec(spec)
spec.loader.exec_module(_lib)
import hipmini as hip
from hipmini import DeviceArray

lib = _lib.load()
g = np.random.default_rng(0)
B, V = args.batch, args.views
H, W = args.hw
N_DEPTHS, RATIOS = (8, 32, 48), (1.0, 2.0, 4.0)
DEPTH_MIN, DEPTH_INTERVAL = 425.0, 2.65
CONV_S1, CONV_S2, CONV_T2 = _lib.CONV_S1, _lib.CONV_S2, _lib.CONV_T2
K3, K5S2, K1, K1UP = _lib.CONV2D_K3, _lib.CONV2D_K5S2, _lib.CONV2D_K1, _lib.CONV2D_K1_UP


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what}: {lib.casmvs_last_error().decode()}")


def hp(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def rand_w(shape, fan_in):
    return (g.standard_normal(shape) * math.sqrt(2.0 / fan_in)).astype(np.float32)


def rand_abn(c):   # folded eval-mode ABN: scale, shift
    return (0.8 + 0.4 * g.random(c)).astype(np.float32), (0.1 * g.standard_normal(c)).astype(np.float32)


def pack_f32(kind, cin, cout, w, sc, sh, three_d):
    n = (lib.casmvs_conv3d_packed_floats if three_d else lib.casmvs_conv2d_packed_floats)(kind, cin, cout)
    assert n, (kind, cin, cout)
    out = np.empty(n, np.float32)
    check((lib.casmvs_conv3d_pack_f32 if three_d else lib.casmvs_conv2d_pack_f32)(kind, cin, cout, hp(w), hp(sc), hp(sh), hp(out)), "pack")
    return DeviceArray.from_numpy(out)


def pack_bytes(nbytes, fn, *a):
    out = np.empty(nbytes, np.uint8)
    check(fn(*a, hp(out)), fn.__name__)
    return DeviceArray.from_numpy(out)


# ---- FeatureNet (mvsnet.py:103-198) ------------------------------------------------------------------------------------
FEATURE = (("conv0.0", K3, 3, 8, 3, True), ("conv0.1", K3, 8, 8, 3, True), ("conv1.0", K5S2, 8, 16, 5, True), ("conv1.1", K3, 16, 16, 3, True),
           ("conv1.2", K3, 16, 16, 3, True), ("conv2.0", K5S2, 16, 32, 5, True), ("conv2.1", K3, 32, 32, 3, True), ("conv2.2", K3, 32, 32, 3, True),
           ("toplayer", K1, 32, 32, 1, False), ("lat1", K1UP, 16, 32, 1, False), ("lat0", K1UP, 8, 32, 1, False), ("smooth1", K3, 32, 16, 3, False),
           ("smooth0", K3, 32, 8, 3, False))
fw, feat_packed = {}, []
for name, kind, cin, cout, k, abn in FEATURE:
    w = rand_w((cout, cin, k, k), cin * k * k)
    sc, sh = rand_abn(cout) if abn else (None, (0.05 * g.standard_normal(cout)).astype(np.float32))
    fw[name] = (w, sc, sh)
    feat_packed.append(pack_f32(kind, cin, cout, w, sc, sh, False))
# the composed full-resolution tail (mvsnet.compose_fpn_tail) in numpy
Ws, Wl = fw["smooth0"][0].astype(np.float64), fw["lat0"][0].astype(np.float64)[:, :, 0, 0]
bl, bs = fw["lat0"][2].astype(np.float64), fw["smooth0"][2].astype(np.float64)
w40 = np.concatenate([np.einsum("omyx,mi->oiyx", Ws, Wl), Ws], axis=1).astype(np.float32)
tap_bias = np.einsum("omyx,m->oyx", Ws, bl)
valid = {0: (1, 2), 1: (0, 1, 2), 2: (0, 1)}
bias9 = np.stack([np.stack([bs + sum(tap_bias[:, ky, kx] for ky in valid[r] for kx in valid[c]) for c in range(3)]) for r in range(3)]).astype(np.float32)
tail_sf = pack_bytes(lib.casmvs_fpn_tail0_splitf16_packed_bytes(), lib.casmvs_fpn_tail0_splitf16_pack, hp(np.ascontiguousarray(w40)))
tail_f32 = pack_f32(K3, 40, 8, np.ascontiguousarray(w40), None, None, False)
bias9_d = DeviceArray.from_numpy(bias9)
ci2d = []
for name in ("conv1.1", "conv1.2", "conv2.1", "conv2.2", "smooth1"):
    w, sc, sh = fw[name]
    cout, cin = w.shape[:2]
    ci2d.append(pack_bytes(lib.casmvs_conv2d_ci_splitf16_packed_bytes(cin, cout), lib.casmvs_conv2d_ci_splitf16_pack, cin, cout, hp(w), hp(sc), hp(sh)))
for name in ("conv1.0", "conv2.0"):   # the 5 x 5 stride-2 layers on the f16 cores (ABI 3: ci_layers[5], [6])
    w, sc, sh = fw[name]
    cout, cin = w.shape[:2]
    ci2d.append(pack_bytes(lib.casmvs_conv2d_k5s2_splitf16_packed_bytes(cin, cout), lib.casmvs_conv2d_k5s2_splitf16_pack, cin, cout, hp(w), hp(sc), hp(sh)))

w0_, sc0_, sh0_ = fw["conv0.0"]
w1_, sc1_, sh1_ = fw["conv0.1"]
ci2d.append(None if args.two_layer_conv0 else pack_bytes(lib.casmvs_fnet_conv0_mm_packed_bytes(), lib.casmvs_fnet_conv0_mm_pack, hp(w0_), hp(sc0_), hp(sh0_), hp(w1_), hp(sc1_), hp(sh1_)))   # ABI 6: ci_layers[7]

# ---- CostRegNet per level (mvsnet.py:201-326) ----------------------------------------------------------------------------
COSTREG = (("conv0", CONV_S1, None, 8), ("conv1", CONV_S2, 8, 16), ("conv2", CONV_S1, 16, 16), ("conv3", CONV_S2, 16, 32), ("conv4", CONV_S1, 32, 32),
           ("conv5", CONV_S2, 32, 64), ("conv6", CONV_S1, 64, 64), ("conv7", CONV_T2, 64, 32), ("conv9", CONV_T2, 32, 16), ("conv11", CONV_T2, 16, 8),
           ("prob", CONV_S1, 8, 1))
costreg, costreg_w = [], []
for l in range(3):
    c_in0 = 8 * 2 ** l
    packed, split = [], [None] * 8
    costreg_w.append({})
    for name, kind, cin, cout in COSTREG:
        cin = c_in0 if cin is None else cin
        w = rand_w((cin, cout, 3, 3, 3) if kind == CONV_T2 else (cout, cin, 3, 3, 3), cin * 27 / (8 if kind == CONV_T2 else 1))
        sc, sh = rand_abn(cout) if name != "prob" else (None, np.zeros(1, np.float32))
        packed.append(pack_f32(kind, cin, cout, w, sc, sh, True))
        costreg_w[l][name] = (w, sc, sh)
        if name == "conv0":
            split[0] = pack_bytes(lib.casmvs_conv0_splitf16_packed_bytes(cin), lib.casmvs_conv0_splitf16_pack, cin, hp(w), hp(sc), hp(sh))
        if name in ("conv2", "conv4", "conv6"):
            split[1 + ("conv2", "conv4", "conv6").index(name)] = pack_bytes(lib.casmvs_conv_ci_splitf16_packed_bytes(cin, cout), lib.casmvs_conv_ci_splitf16_pack,
                                                                             cin, cout, hp(w), hp(sc), hp(sh))
        if name in ("conv1", "conv3"):
            split[6 + ("conv1", "conv3").index(name)] = pack_bytes(lib.casmvs_conv_s2_splitf16_packed_bytes(cin, cout), lib.casmvs_conv_s2_splitf16_pack,
                                                                   cin, cout, hp(w), hp(sc), hp(sh))
        if name == "conv9":
            split[4] = pack_bytes(lib.casmvs_deconv9_splitf16_packed_bytes(), lib.casmvs_deconv9_splitf16_pack, hp(w), hp(sc), hp(sh))
        if name == "conv11":
            split[5] = pack_bytes(lib.casmvs_deconv11_splitf16_packed_bytes(), lib.casmvs_deconv11_splitf16_pack, hp(w), hp(sc), hp(sh))
    costreg.append((packed, split))

F32_LAYERS = set(filter(None, args.f32_layers.split(",")))
SPLIT_ORDER = ("conv0", "conv2", "conv4", "conv6", "conv9", "conv11", "conv1", "conv3")   # casmvs_costreg_regress_f32: split_layers[0..7]
assert F32_LAYERS <= set(SPLIT_ORDER), F32_LAYERS

# ---- inputs: images, the DTU-like rig of synthetic.dtu_like_cameras / make_inputs --------------------------------------------
imgs = DeviceArray.from_numpy(g.standard_normal((B * V, 3, H, W)).astype(np.float32))


def rig(b):
    f = 1446.0 * W / 640.0
    K0 = np.array([[f, 0, W / 2.0], [0, f, H / 2.0], [0, 0, 1.0]])
    dirs = [(1, 0), (-1, 0), (0, 1), (0, -1), (0.7071, 0.7071), (-0.7071, -0.7071), (0.7071, -0.7071), (-0.7071, 0.7071)]
    depth_mid, baseline = 680.0, 60.0 * (1.0 + 0.1 * b)
    views = []
    for v in range(V):
        R, c = np.eye(3), np.zeros(3)
        if v:
            dx, dy = dirs[(v - 1) % 8]
            s = baseline * (1.0 + 0.15 * ((v - 1) // 8) + 0.07 * (v - 1))
            c = np.array([dx * s, dy * s, 0.0])
            ax, ay = -math.atan2(c[1], depth_mid), math.atan2(c[0], depth_mid)
            Rx = np.array([[1, 0, 0], [0, math.cos(ax), -math.sin(ax)], [0, math.sin(ax), math.cos(ax)]])
            Ry = np.array([[math.cos(ay), 0, math.sin(ay)], [0, 1, 0], [-math.sin(ay), 0, math.cos(ay)]])
            R = Rx @ Ry
        t = -R @ c
        mats = []
        for l in range(3):
            K = K0.copy()
            K[:2] /= 2 ** l
            P = np.eye(4)
            P[:3, :3], P[:3, 3] = K @ R, K @ t
            mats.append(P.astype(np.float32))
        views.append(np.stack(mats))
    ref_inv = np.linalg.inv(views[0].astype(np.float64))
    return np.stack([(views[v].astype(np.float64) @ ref_inv)[:, :3] for v in range(1, V)]).astype(np.float32)   # (V-1, levels, 3, 4)


proj = np.stack([rig(b) for b in range(B)])                                  # (B, V-1, levels, 3, 4)
proj_l = [DeviceArray.from_numpy(np.ascontiguousarray(proj[:, :, l])) for l in range(3)]

# ---- device buffers ---------------------------------------------------------------------------------------------------------
N = B * V
feat = [DeviceArray((N, 8 * 2 ** l, H >> l, W >> l)) for l in range(3)]
feat_cl = [DeviceArray((N, H >> l, W >> l, 8 * 2 ** l)) for l in range(3)]
feat_ws = DeviceArray(lib.casmvs_featurenet_workspace_bytes(N, H, W), np.uint8)
levels = []
for l in range(3):
    D, h, w, C = N_DEPTHS[l], H >> l, W >> l, 8 * 2 ** l
    interval = DEPTH_INTERVAL * RATIOS[l]
    levels.append(dict(D=D, h=h, w=w, C=C, dv=DeviceArray((B, D, h, w)), vol=DeviceArray((B, C, D, h, w)), cost=DeviceArray((B, D, h, w)),
                       depth=DeviceArray((B, h, w)), conf=DeviceArray((B, h, w)), ws=DeviceArray(lib.casmvs_costreg_workspace_bytes(B, D, h, w), np.uint8),
                       interval=DeviceArray.from_numpy(np.full(B, interval, np.float32)), half=DeviceArray.from_numpy(np.full(B, D / 2 * interval, np.float32)),
                       lds=bool(lib.casmvs_costvol_lds_preferred(C, w, D, V - 1, 1))))
dmin = DeviceArray.from_numpy(np.full(B, DEPTH_MIN, np.float32))
stream = hip.stream_create()
st = ctypes.c_void_p(stream)
arr13 = (ctypes.c_void_p * 13)(*[p.ptr for p in feat_packed])
ci5 = None if args.all_f32 else (ctypes.c_void_p * 8)(*[None if (p is None or (i in (5, 6) and args.f32_k5s2)) else p.ptr for i, p in enumerate(ci2d)])
STAGES = ["feature"] + [f"{s}_{l}" for l in (2, 1, 0) for s in ("hypotheses", "costvol", "costreg")]
events = {s: (hip.Event(), hip.Event()) for s in STAGES}


def run_stage(name, fn, timed):
    if timed:
        events[name][0].record(stream)
    fn()
    if timed:
        events[name][1].record(stream)


def feature_stage():
    # --feature-split K: FeatureNet over K groups of images, one after the other through the same workspace (cache blocking: a group's intermediates
    # - 8 channels at full resolution = 10.5 MB per image - stay in the 256 MB Infinity Cache between the layer that writes and the one that reads them)
    K = args.feature_split
    assert N % K == 0, (N, K)
    n = N // K
    at = lambda a, i: ctypes.c_void_p(a.ptr + i * n * (a.nbytes // N))
    for i in range(K):
        check(lib.casmvs_featurenet_forward_fused_f32(
            arr13, (tail_f32 if args.all_f32 else tail_sf).p, 0 if args.all_f32 else 1, bias9_d.p, ci5, at(imgs, i), at(feat[0], i) if args.nchw_feats else None,
            at(feat[1], i) if args.nchw_feats else None, at(feat[2], i), at(feat_cl[0], i), at(feat_cl[1], i), at(feat_cl[2], i), feat_ws.p, n, H, W,
            ctypes.c_float(0.01), None, st), "featurenet")


def step(timed=False):
    run_stage("feature", feature_stage, timed)
    prev = None
    for l in (2, 1, 0):
        L = levels[l]
        D, h, w, C = L["D"], L["h"], L["w"], L["C"]
        if prev is None:
            fn = lambda: check(lib.casmvs_depth_hypotheses_f32(None, dmin.p, L["interval"].p, None, L["dv"].p, B, D, h, w, 0, 0, st), "hypotheses")
        else:
            fn = lambda: check(lib.casmvs_depth_hypotheses_f32(prev["depth"].p, None, L["interval"].p, L["half"].p, L["dv"].p, B, D, h, w, prev["h"], prev["w"], st), "hypotheses")
        run_stage(f"hypotheses_{l}", fn, timed)
        cv = lib.casmvs_costvol_var_lds_f32 if L["lds"] else lib.casmvs_costvol_var_nhwc_f32
        run_stage(f"costvol_{l}", lambda: check(cv(feat_cl[l].p, proj_l[l].p, L["dv"].p, L["vol"].p, B, V, C, h, w, D, st), "costvol"), timed)
        packed, split = costreg[l]
        arr11 = (ctypes.c_void_p * 11)(*[p.ptr for p in packed])
        sp = None if args.all_f32 else (ctypes.c_void_p * 8)(*[None if (s is None or n in F32_LAYERS) else s.ptr for n, s in zip(SPLIT_ORDER, split)])
        arith = 0 if (args.all_f32 or "conv0" in F32_LAYERS) else 2
        run_stage(f"costreg_{l}", lambda: check(lib.casmvs_costreg_regress_f32(
            arr11, sp, arith, L["vol"].p, L["dv"].p, L["cost"].p, L["depth"].p, L["conf"].p, None, L["ws"].p, B, C, D, h, w,
            ctypes.c_float(0.01), None, st), "costreg_regress"), timed)
        prev = L


for _ in range(args.warmup):
    step()
hip.synchronize()
t0, t1 = hip.Event(), hip.Event()
t0.record(stream)
for _ in range(args.steps):
    step()
t1.record(stream)
t1.synchronize()
ms = t1.ms_since(t0) / args.steps
step(timed=True)
hip.synchronize()
stage_ms = {s: events[s][1].ms_since(events[s][0]) for s in STAGES}
d0 = levels[0]["depth"].numpy()
lo, hi = DEPTH_MIN - 200.0, DEPTH_MIN + 192 * DEPTH_INTERVAL + 200.0
ok = bool(np.isfinite(d0).all() and d0.min() > lo and d0.max() < hi)
print(f"{hip.device_name()}  lib {os.path.basename(_lib.LIB_PATH)}  batch {B} x {V} views {W}x{H}  {'all-f32' if args.all_f32 else 'split-f16 layer set'}"
      + (f"  float32: {sorted(F32_LAYERS)}" if F32_LAYERS else ""))
d0_sum = float(np.float64(d0).sum())
print(f"depth_0 checksum {d0_sum:.6f} (compare runs of two builds / layer sets: the same weights and inputs, results equal to ~1e-5 relative)")
print(f"step {ms:.3f} ms  = {B / ms * 1e3:.1f} depth maps/s (kernel by kernel on one stream, {args.steps} steps after {args.warmup})")
print("stages (one instrumented step, ms): " + "  ".join(f"{s} {stage_ms[s]:.3f}" for s in STAGES))
print(f"sum of stages {sum(stage_ms.values()):.3f} ms; depth_0 range [{d0.min():.1f}, {d0.max():.1f}] mean {d0.mean():.1f}  {'ok' if ok else 'OUT OF RANGE'}")
sys.exit(0 if ok else 1)
