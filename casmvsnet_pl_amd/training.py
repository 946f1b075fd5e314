"""Training through the HIP engine: what `train.py:99-127` needs - a train-mode forward of
`CascadeMVSNet` whose graph autograd can walk, with every tensor-sized operation executed by libcasmvs_hip.so.

reference op (train mode)                              here
nn.Conv2d / Conv3d / ConvTranspose3d forward           the inference MFMA kernels, un-folded (scale 1, slope 1): `conv`
  ... gradient w.r.t. the input                        the SAME forward kernels with adjoint weights (transposed / mirrored through
                                                       the packing index, or the strided <-> transposed kind); the three layer shapes
                                                       without such a twin (Conv2d k5 s2, the 8-channel 1x1 lateral) use
                                                       casmvs_conv_dgrad_direct_f32
  ... gradient w.r.t. the weight / bias                casmvs_conv_wgrad_f32 (matrix cores) / casmvs_channel_sums_f64
ABN / InPlaceABN in train mode (modules.py:14,27)      casmvs_channel_sums_f64 -> casmvs_abn_train_finish_f32 -> casmvs_abn_apply_f32;
                                                       backward casmvs_abn_backward_{sums_f64,finish_f32,apply_f32}; running statistics updated
F.interpolate(x2, bilinear, align_corners) + lateral   casmvs_upsample2x_add_f32 / casmvs_upsample2x_backward_f32
homo_warp + variance volume (mvsnet.py:137-167)        fused forward kernel; casmvs_costvol_var_backward_f32
softmax + depth regression (mvsnet.py:175-177)         autograd.softmax_depth_regression

torch is the autograd tape and the allocator; the skip additions of the U-Net and the gradient accumulation are torch adds.  Nothing here runs on CPU tensors.  DESIGN.md 2.6 has the kernels and the measured step
times (18 ms per step at the reference's default training configuration, 15 ms captured as one hipGraph).
"""
import ctypes
import weakref

import torch

from . import _lib, ops
from . import autograd as A
from ._lib import CONV_S1, CONV_S2, CONV_T2, CONV2D_K3, CONV2D_K5S2, CONV2D_K1

_3D = (CONV_S1, CONV_S2, CONV_T2)


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream(t):
    from . import streams
    return streams.launch_stream(t)


# ---- device-side packing of a layer image --------------------------------------------------------------------------------
# casmvs_conv{2,3}d_pack_f32 is host code (it runs once per checkpoint in the inference engine).  In training the weights
# change every step: the packing is a fixed gather, so it is derived ONCE per layer shape by packing a weight tensor
# whose values are their own indices, and applied on the device as one index_select per forward.
_PACK_MAPS = {}


def _pack_map(kind, cin, cout, has_bias, device, adjoint=False):
    """int32 gather index of the packed image of layer (kind, cin -> cout) over the source [weight, bias, 0, 1].
    adjoint: the layer's weight is `w.transpose(0, 1).flip(spatial dims)` of the tensor handed to device_pack (shape
    (cin, cout, k...): the input-gradient layer of a stride-1 convolution) - the permutation is folded into the index."""
    key = (kind, cin, cout, has_bias, str(device), adjoint)
    m = _PACK_MAPS.get(key)
    if m is not None:
        return m
    three_d = kind in _3D
    k = {CONV2D_K3: 3, CONV2D_K5S2: 5, CONV2D_K1: 1}.get(kind, 3)
    shape = ((cin, cout) if kind == CONV_T2 else (cout, cin)) + ((3, 3, 3) if three_d else (k, k))
    n = 1
    for s in shape:
        n *= s
    if n + 2 >= (1 << 24):
        raise RuntimeError("training: layer too large for the index-valued packing probe")
    probe_w = (torch.arange(n, dtype=torch.float32) + 2.0).reshape(shape)         # weight i -> value i + 2
    probe_shift = -(torch.arange(cout, dtype=torch.float32) + 1.0) if has_bias else None   # bias c -> value -(c + 1)
    packed = (ops.conv3d_pack if three_d else ops.conv2d_pack)(kind, probe_w, None, probe_shift)
    # source vector: [weights (n), bias (cout or 0), 0.0, 1.0]
    nb = cout if has_bias else 0
    idx = torch.empty(packed.numel(), dtype=torch.int64)
    v = packed.round().to(torch.int64)
    widx = v[v >= 2] - 2
    if adjoint:   # flat index in the logical (transposed, mirrored) weight -> flat index in the tensor as stored
        if kind == CONV_T2:
            raise RuntimeError("training: no adjoint packing for ConvTranspose3d")
        stored = (cin, cout) + tuple(shape[2:])
        perm = torch.arange(n, dtype=torch.int64).reshape(stored).transpose(0, 1)
        if shape[2] > 1:
            perm = perm.flip(tuple(range(2, len(shape))))
        widx = perm.reshape(-1)[widx]
    idx[v >= 2] = widx
    idx[v == 1] = n + nb + 1
    idx[v == 0] = n + nb
    idx[v < 0] = n + (-v[v < 0] - 1)
    m = _PACK_MAPS[key] = idx.to(torch.int32).to(device)
    return m


class PackPlan:
    """Every packed layer image of a training step in ONE gather launch (casmvs_pack_gather_batch_f32).  The weights change every step, so every
    convolution's operand image (and, for the input gradients, the image of the adjoint layer) is re-derived from the parameters each step: 88 launches of
    ~4.7 us.  The plan records the (weight, bias, kind, adjoint) requests of a model's first training step together with persistent output buffers; from
    the second step on `begin_step` fills ALL of them with one launch at the start of the forward and `device_pack` hands the buffers out, as long as the
    parameter's storage and version are the ones the launch saw (anything else - a new request, a parameter replaced or modified after `begin_step` - takes
    the single-image launch and, for a new request, joins the plan).  Capturable: the batched launch is part of the captured step.
    A plan serves the parameters of ITS model only (`owned`: the storages of model.parameters() at the last begin_step): a convolution of another model - or
    a stray device_pack call - that runs while this plan is the active one packs on its own and never joins the table; entries whose parameter the model no
    longer holds (replaced by .to(), a re-initialised layer, another device) are dropped at the next begin_step together with their buffers (a captured
    graph that still read them would read a freed parameter as well)."""

    def __init__(self, model=None):
        import weakref
        self.model = None if model is None else weakref.ref(model)
        self.owned = None      # data_ptr -> device of the model's parameters at the last begin_step (None: no model given, every request is served)
        self.entries = {}      # (weight ptr, bias ptr, kind, adjoint) -> [weight, bias, index, out, version seen by the last batched launch or None,
        #                          steps since the last request]
        self.table = None      # device copy of the segment array; None = rebuild before the next launch
        self.table_keys = ()   # the entries the table lists
        self.retired = []      # earlier tables: a captured hipGraph may still read them (a few KB each)
        self.n_blocks = 0
        self.launches = self.hits = self.misses = 0   # batched launches; images handed out without / with a launch of their own (tests)

    @staticmethod
    def _key(weight, bias, kind, adjoint):
        return (weight.data_ptr(), 0 if bias is None else bias.data_ptr(), int(kind), bool(adjoint))

    def serves(self, weight):
        return self.owned is None or self.owned.get(weight.data_ptr()) == weight.device

    def lookup(self, kind, weight, bias, adjoint, idx):
        key = self._key(weight, bias, kind, adjoint)
        e = self.entries.get(key)
        if e is None:
            out = torch.empty(idx.numel(), dtype=torch.float32, device=weight.device)
            self.entries[key] = [weight, bias, idx, out, None, 0]
            self.misses += 1
            return out, False
        e[5] = 0
        fresh = e[4] is not None and e[4] == (weight._version, -1 if bias is None else bias._version)
        self.hits += fresh
        self.misses += not fresh
        return e[3], fresh

    def begin_step(self):
        """One launch for every recorded image; nothing on the first step (no requests yet)."""
        import numpy as np
        model = None if self.model is None else self.model()
        if model is not None:
            self.owned = {p.data_ptr(): p.device for p in model.parameters()}
            for k in [k for k, e in self.entries.items() if not self.serves(e[0])]:
                del self.entries[k]
        # an image nobody asked for during two whole steps (its parameter was replaced: .to(), a re-initialised layer) is no longer packed; its buffer
        # stays (a captured graph may read it) and it rejoins the table when it is requested again
        for e in self.entries.values():
            e[5] += 1
        keys = tuple(k for k, e in self.entries.items() if e[5] <= 2)
        if keys != self.table_keys:
            if self.table is not None:
                self.retired.append(self.table)
            self.table, self.table_keys = None, keys
        if not keys:
            return
        entries = [self.entries[k] for k in keys]
        dev = entries[0][0].device   # one device: entries on another one were dropped above
        if self.table is None:
            seg = np.zeros(len(entries), dtype=np.dtype([("w", "<u8"), ("b", "<u8"), ("idx", "<u8"), ("out", "<u8"), ("n_w", "<i4"), ("n_b", "<i4"),
                                                          ("n_out", "<i4"), ("first", "<i4")]))
            block = 0
            for i, (w, bz, idx, out, _, _) in enumerate(entries):
                seg[i] = (w.data_ptr(), 0 if bz is None else bz.data_ptr(), idx.data_ptr(), out.data_ptr(), w.numel(), 0 if bz is None else bz.numel(),
                          idx.numel(), block)
                block += (idx.numel() + 255) // 256
            self.table = torch.from_numpy(seg.view(np.uint8).copy()).to(dev)
            self.n_blocks = block
        with torch.cuda.device(dev):
            rc = _lib.load().casmvs_pack_gather_batch_f32(ctypes.c_void_p(self.table.data_ptr()), len(entries), self.n_blocks, _stream(self.table))
        _lib.check(rc, "casmvs_pack_gather_batch_f32")
        for e in entries:
            e[4] = (e[0]._version, -1 if e[1] is None else e[1]._version)
        self.launches += 1


# The plan of the training step in flight (set by cascade_forward_train; its backward runs under the same plan), as a WEAK reference: the plan lives on its
# model (pack_plan_of), so the module global neither keeps a dropped model's weights and packed images alive nor serves them to another model's step
# (PackPlan.serves also checks ownership).  release_plan() clears it once a step's graph is consumed (train_steps does, after optimizer.step()).
_ACTIVE_PLAN = None


def _active_plan():
    return None if _ACTIVE_PLAN is None else _ACTIVE_PLAN()


def set_active_plan(plan):
    """Make `plan` (a PackPlan, or None) the plan device_pack consults; held weakly."""
    global _ACTIVE_PLAN
    _ACTIVE_PLAN = None if plan is None else weakref.ref(plan)


def release_plan():
    """Forget the training step in flight: later device_pack calls pack on their own until the next train-mode forward."""
    global _ACTIVE_PLAN
    _ACTIVE_PLAN = None


def pack_plan_of(model):
    """The model's PackPlan (created on first use; kept on the module object, so it goes away with it)."""
    plan = model.__dict__.get("_casmvs_pack_plan")
    if plan is None:
        plan = model.__dict__["_casmvs_pack_plan"] = PackPlan(model)
    return plan


def device_pack(kind, weight, bias=None, adjoint=False):
    """Packed layer image (the operand casmvs_conv{2,3}d_forward_f32 takes) of `weight` [+ `bias`] with scale 1, on the
    device: one gather launch (casmvs_pack_gather_f32) - or none, when the step's PackPlan has filled it already.
    adjoint: pack `weight.transpose(0, 1).flip(spatial)` instead."""
    if (kind == CONV_T2) != adjoint:
        cin, cout = weight.shape[:2]
    else:
        cout, cin = weight.shape[:2]
    idx = _pack_map(kind, cin, cout, bias is not None, weight.device, adjoint)
    w = weight.detach().contiguous().float()
    bz = None if bias is None else bias.detach().contiguous().float()
    plan = _active_plan()
    planned = plan is not None and w.data_ptr() == weight.data_ptr() and (bias is None or bz.data_ptr() == bias.data_ptr()) and plan.serves(weight)
    if planned:
        out, fresh = plan.lookup(kind, weight.detach(), None if bias is None else bias.detach(), adjoint, idx)
        if fresh:
            return out
    else:
        out = torch.empty(idx.numel(), dtype=torch.float32, device=weight.device)
    with torch.cuda.device(weight.device):
        rc = _lib.load().casmvs_pack_gather_f32(_ptr(w), _ptr(bz), _ptr(idx), _ptr(out), w.numel(), 0 if bz is None else bz.numel(),
                                                idx.numel(), _stream(w))
    _lib.check(rc, "casmvs_pack_gather_f32")
    return out


def _forward_kernel_supports(kind, cin, cout):
    lib = _lib.load()
    fn = lib.casmvs_conv3d_packed_floats if kind in _3D else lib.casmvs_conv2d_packed_floats
    return fn(kind, cin, cout) > 0


def _conv_raw(kind, weight, bias, x, adjoint=False):
    """conv (no activation) through the inference MFMA kernels.  adjoint: the layer whose weight is
    `weight.transpose(0, 1).flip(spatial)` (the input gradient of a stride-1 convolution)."""
    cout = weight.shape[1] if (kind == CONV_T2) != adjoint else weight.shape[0]
    packed = device_pack(kind, weight, bias, adjoint)
    if kind in _3D:
        return ops.conv3d_forward(kind, packed, x, cout, None, slope=1.0)
    return ops.conv2d_forward(kind, packed, x, cout, slope=1.0)


PROB_WGRAD_KERNEL = True   # False: the generic matrix-core kernel also for `prob` (A/B runs)


def conv_wgrad(kind, x, grad_out, weight_shape):
    """Gradient w.r.t. the weight (torch layout of `kind`), casmvs_conv_wgrad_f32."""
    lib = _lib.load()
    x, grad_out = x.contiguous(), grad_out.contiguous()
    if kind in _3D:
        B, cin, D, H, W = x.shape
    else:
        B, cin, H, W = x.shape
        D = 1
    cout = grad_out.shape[1]
    if (PROB_WGRAD_KERNEL and kind == CONV_S1 and cin == 8 and cout == 1 and lib.casmvs_prob_wgrad_supported(B, D, H, W)
            and x.data_ptr() % 16 == 0 and grad_out.data_ptr() % 16 == 0):
        # the `prob` layer: one output channel would be padded to a 16-row matrix tile; its own vector-ALU kernel is 3.3x faster
        ws = torch.empty(lib.casmvs_prob_wgrad_workspace_bytes(B, D, H, W), dtype=torch.uint8, device=x.device)
        gw = torch.empty(weight_shape, dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            rc = lib.casmvs_prob_wgrad_f32(_ptr(x), _ptr(grad_out), _ptr(gw), ctypes.c_void_p(ws.data_ptr()), B, D, H, W, _stream(x))
        _lib.check(rc, "casmvs_prob_wgrad_f32")
        return gw
    nbytes = lib.casmvs_conv_wgrad_workspace_bytes(kind, B, cin, cout, D, H, W)
    if nbytes == 0:
        raise RuntimeError(f"conv_wgrad: unsupported kind={kind} input {tuple(x.shape)}")
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    gw = torch.empty(weight_shape, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        rc = lib.casmvs_conv_wgrad_f32(kind, _ptr(x), _ptr(grad_out), _ptr(gw), ctypes.c_void_p(ws.data_ptr()), B, cin, cout, D, H, W, _stream(x))
    _lib.check(rc, "casmvs_conv_wgrad_f32")
    return gw


def conv_dgrad(kind, weight, grad_out, x_shape):
    """Gradient w.r.t. the input: a forward launch of the adjoint layer, or the direct kernel."""
    grad_out = grad_out.contiguous()
    if kind == CONV_T2:                      # ConvTranspose3d (cin, cout, k): adjoint = strided conv cout -> cin, same tensor
        cin, cout = weight.shape[:2]
        adj_kind, adj_w, a_in, a_out = CONV_S2, weight, cout, cin
    elif kind == CONV_S2:                    # strided conv (cout, cin, k): adjoint = ConvTranspose3d (cin_T = cout, cout_T = cin)
        cout, cin = weight.shape[:2]
        adj_kind, adj_w, a_in, a_out = CONV_T2, weight, cout, cin
    elif kind in (CONV_S1, CONV2D_K3, CONV2D_K1):   # stride 1: swap the channel roles, mirror the taps
        cout, cin = weight.shape[:2]
        adj_kind, adj_w, a_in, a_out = kind, None, cout, cin     # the permutation is part of the packing index
    else:
        cout, cin = weight.shape[:2]
        adj_kind = None
    if adj_kind is not None and _forward_kernel_supports(adj_kind, a_in, a_out):
        if adj_w is None:
            return _conv_raw(adj_kind, weight, None, grad_out, adjoint=True)
        return _conv_raw(adj_kind, adj_w.contiguous(), None, grad_out)
    # Conv2d k5 s2, 1x1 / 3x3 with a channel count the MFMA forms do not take as an output
    if kind == CONV_T2:
        raise RuntimeError("conv_dgrad: ConvTranspose3d shape without an adjoint kernel")
    gin = torch.empty(x_shape, dtype=torch.float32, device=grad_out.device)
    if kind in _3D:
        B, _, D, H, W = x_shape
    else:
        B, _, H, W = x_shape
        D = 1
    w = weight.detach().contiguous().float()
    with torch.cuda.device(grad_out.device):
        rc = _lib.load().casmvs_conv_dgrad_direct_f32(kind, _ptr(w), _ptr(grad_out), _ptr(gin), B, cin, cout, D, H, W, _stream(grad_out))
    _lib.check(rc, "casmvs_conv_dgrad_direct_f32")
    return gin


def channel_sums(x):
    """x (N,C,...) -> (sum, sum of squares) per channel as float64 (C,) tensors (casmvs_channel_sums_f64)."""
    x = x.contiguous()
    N, C = x.shape[:2]
    n = x.numel() // (N * C)
    lib = _lib.load()
    blocks = lib.casmvs_channel_sums_blocks(N, n)
    part = torch.empty((C, blocks, 2), dtype=torch.float64, device=x.device)
    with torch.cuda.device(x.device):
        rc = lib.casmvs_channel_sums_f64(_ptr(x), _ptr(part), N, C, n, _stream(x))
    _lib.check(rc, "casmvs_channel_sums_f64")
    s = part.sum(1)
    return s[:, 0], s[:, 1]


class _Conv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, kind):
        x = x.contiguous().float()
        ctx.kind = kind
        ctx.has_bias = bias is not None
        ctx.save_for_backward(x, weight)
        return _conv_raw(kind, weight, bias, x)

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        gy = gy.contiguous().float()
        gx = conv_dgrad(ctx.kind, weight, gy, tuple(x.shape)) if ctx.needs_input_grad[0] else None
        gw = conv_wgrad(ctx.kind, x, gy, tuple(weight.shape)) if ctx.needs_input_grad[1] else None
        gb = channel_sums(gy)[0].float() if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        return gx, gw, gb, None


def conv(x, weight, bias, kind):
    """Differentiable convolution of one of the model's layer kinds (forward, input, weight and bias gradients in HIP)."""
    if not x.is_cuda:
        raise RuntimeError("casmvsnet_pl_amd.training runs on the MI355X only; there is no CPU fallback")
    return _Conv.apply(x, weight, bias, kind)


class _ABNTrain(torch.autograd.Function):
    """y = leaky_relu(batch_norm(x) with BATCH statistics); updates the running statistics in place like F.batch_norm.
    `weight` is the module's parameter; abs_eps >= 0 selects InPlaceABN's gamma = |weight| + abs_eps (inplace_abn.py).
    Per layer: channel sums -> the elementwise apply, whose workgroups derive the statistics, the folded scale / shift and (one of them) the running
    statistics from the partial sums themselves; nothing of it is a torch operation."""

    @staticmethod
    def forward(ctx, x, weight, beta, running_mean, running_var, momentum, eps, slope, abs_eps):
        x = x.contiguous().float()
        N, C = x.shape[:2]
        n = x.numel() // (N * C)
        M = N * n
        lib = _lib.load()
        blocks = lib.casmvs_channel_sums_blocks(N, n)
        part = torch.empty((C, blocks, 2), dtype=torch.float64, device=x.device)
        vec = torch.empty((4, C), dtype=torch.float32, device=x.device)   # scale, shift, mean, rstd
        y = torch.empty_like(x)
        w, bta = weight.detach().contiguous().float(), beta.detach().contiguous().float()
        track = running_mean is not None and running_var is not None
        with torch.cuda.device(x.device):
            st = _stream(x)
            rc = lib.casmvs_channel_sums_f64(_ptr(x), _ptr(part), N, C, n, st)
            _lib.check(rc, "casmvs_channel_sums_f64")
            # statistics -> folded scale / shift, running statistics AND the elementwise pass in one launch (every workgroup reduces the partial sums)
            rc = lib.casmvs_abn_train_apply_f32(_ptr(x), _ptr(part), blocks, float(M), _ptr(w), _ptr(bta), float(abs_eps), float(eps),
                                                float(momentum), _ptr(running_mean) if track else None,
                                                _ptr(running_var) if track else None, _ptr(vec[0]), _ptr(vec[1]), _ptr(vec[2]),
                                                _ptr(vec[3]), _ptr(y), N, C, n, float(slope), st)
            _lib.check(rc, "casmvs_abn_train_apply_f32")
        if track:   # the kernel wrote the buffers behind torch's back: bump their version counters (packed-weight caches key on them)
            torch.autograd.graph.increment_version(running_mean)
            torch.autograd.graph.increment_version(running_var)
        ctx.save_for_backward(x, y, w, vec)
        ctx.slope, ctx.M, ctx.abs_eps = float(slope), M, float(abs_eps)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, y, w, vec = ctx.saved_tensors
        gy = gy.contiguous().float()
        N, C = x.shape[:2]
        n = x.numel() // (N * C)
        lib = _lib.load()
        blocks = lib.casmvs_channel_sums_blocks(N, n)
        part = torch.empty((C, blocks, 2), dtype=torch.float64, device=x.device)
        out = torch.empty((2, C), dtype=torch.float32, device=x.device)   # grad_weight, grad_bias
        gx = torch.empty_like(x)
        scale, mean, rstd = vec[0], vec[2], vec[3]
        with torch.cuda.device(x.device):
            st = _stream(x)
            rc = lib.casmvs_abn_backward_sums_f64(_ptr(gy), _ptr(y), _ptr(x), _ptr(mean), _ptr(rstd), _ptr(part), N, C, n, ctx.slope, st)
            _lib.check(rc, "casmvs_abn_backward_sums_f64")
            rc = lib.casmvs_abn_backward_apply_fused_f32(_ptr(gy), _ptr(y), _ptr(x), _ptr(part), blocks, float(ctx.M), _ptr(w), ctx.abs_eps,
                                                         _ptr(scale), _ptr(mean), _ptr(rstd), _ptr(out[0]), _ptr(out[1]), _ptr(gx), N, C, n,
                                                         ctx.slope, st)
            _lib.check(rc, "casmvs_abn_backward_apply_fused_f32")
        return gx, out[0], out[1], None, None, None, None, None, None


def abn_train(norm, x):
    """Train-mode forward of an ABN-like module (`weight`, `bias`, `running_mean`, `running_var`, `eps`, `momentum`,
    leaky-relu slope): inplace_abn.ABN (gamma = weight) or InPlaceABN (gamma = |weight| + eps, see inplace_abn.py)."""
    from .inplace_abn import InPlaceABN
    if not getattr(norm, "affine", True) or norm.weight is None:
        raise RuntimeError("training: ABN without affine parameters is not supported")
    if isinstance(norm, InPlaceABN):
        weight, abs_eps = norm.weight, float(norm.eps)       # |weight| + eps inside the epilogue kernel (and its gradient)
    else:                                                    # any other module: its own gamma expression stays an autograd graph
        weight, abs_eps = (norm._gamma() if hasattr(norm, "_gamma") else norm.weight), -1.0
    slope = norm.leaky_slope() if hasattr(norm, "leaky_slope") else float(getattr(norm, "activation_param", 0.01))
    return _ABNTrain.apply(x, weight, norm.bias, norm.running_mean, norm.running_var, float(norm.momentum), float(norm.eps), slope,
                           abs_eps)


class _UpsampleAdd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, lat, up):
        lat, up = lat.contiguous().float(), up.contiguous().float()
        N, C, H, W = lat.shape
        if tuple(up.shape) != (N, C, H // 2, W // 2):
            raise ValueError(f"upsample_add: shapes {tuple(lat.shape)} {tuple(up.shape)}")
        out = torch.empty_like(lat)
        with torch.cuda.device(lat.device):
            rc = _lib.load().casmvs_upsample2x_add_f32(_ptr(lat), _ptr(up), _ptr(out), N, C, H, W, _stream(lat))
        _lib.check(rc, "casmvs_upsample2x_add_f32")
        ctx.shape = (N, C, H, W)
        return out

    @staticmethod
    def backward(ctx, g):
        N, C, H, W = ctx.shape
        g = g.contiguous().float()
        gup = None
        if ctx.needs_input_grad[1]:
            gup = torch.empty((N, C, H // 2, W // 2), dtype=torch.float32, device=g.device)
            with torch.cuda.device(g.device):
                rc = _lib.load().casmvs_upsample2x_backward_f32(_ptr(g), _ptr(gup), N, C, H, W, _stream(g))
            _lib.check(rc, "casmvs_upsample2x_backward_f32")
        return (g if ctx.needs_input_grad[0] else None), gup


def upsample_add(lat, up):
    """mvsnet.py:36-38: F.interpolate(up, scale_factor=2, mode="bilinear", align_corners=True) + lat."""
    return _UpsampleAdd.apply(lat, up)


def _volume_backward_workspace(feats, G, D):
    """Caller-owned scratch of casmvs_costvol_{var,gwc}_backward_f32 (the 64-bit fixed-point gradient map + the channels' largest magnitudes: the backward
    is order-independent, so a training step gives the same bits run to run); from torch's caching allocator, i.e. stream-ordered and capture-safe."""
    B, V, C, h, w = feats.shape
    n = _lib.load().casmvs_costvol_backward_workspace_bytes(B, V, C, int(G), int(D), h, w)
    return torch.empty(n // 8 + 1, dtype=torch.int64, device=feats.device)


class _VarianceVolume(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feats, proj_mats, depth_values):
        feats = feats.contiguous().float()
        proj_mats, depth_values = proj_mats.detach().contiguous().float(), depth_values.detach().contiguous().float()
        ctx.save_for_backward(feats, proj_mats, depth_values)
        B, V, C, h, w = feats.shape
        if C in (8, 16, 32):   # pixel-major copy: what the fast forward kernels read
            nhwc = ops.nchw_to_nhwc(feats.reshape(B * V, C, h, w)).view(B, V, h, w, C)
            return ops.costvol(nhwc, proj_mats, depth_values, 1, channels_last=True)
        return ops.costvol(feats, proj_mats, depth_values, 1)

    @staticmethod
    def backward(ctx, gvol):
        feats, proj_mats, depth_values = ctx.saved_tensors
        B, V, C, h, w = feats.shape
        D = depth_values.shape[1]
        gvol = gvol.contiguous().float()
        gfeats = torch.empty_like(feats)
        ws = _volume_backward_workspace(feats, 0, D)
        with torch.cuda.device(feats.device):
            rc = _lib.load().casmvs_costvol_var_backward_f32(_ptr(feats), _ptr(proj_mats), _ptr(depth_values), _ptr(gvol), _ptr(gfeats), _ptr(ws),
                                                             B, V, C, h, w, D, _stream(feats))
        _lib.check(rc, "casmvs_costvol_var_backward_f32")
        return gfeats, None, None


def variance_volume(feats, proj_mats, depth_values):
    """Differentiable mvsnet.py:137-167 (G = 1): feats (B,V,C,h,w), proj_mats (B,V-1,3,4), depth_values (B,D,h,w) [no grad]
    -> (B,C,D,h,w)."""
    return _VarianceVolume.apply(feats, proj_mats, depth_values)


class _GroupwiseVolume(torch.autograd.Function):
    """mvsnet.py:142-144,157-162,169-172.  Forward: the fused inference kernel (no warped volume is materialised or kept for the
    backward).  Backward, from the saved features, ONE launch (casmvs_costvol_gwc_backward_f32, the variance backward's kernel with another
    contribution): d / d warped_v[c] = g[c // (C/G)] * ref[c] / (C/G * (V-1)) scattered through the bilinear weights into a fixed-point LDS image per
    workgroup, d / d ref[c] = g[c // (C/G)] * sum_v warped_v[c] / (C/G * (V-1)) from re-gathered values - no channel-expanded gradient volume, no warped
    volume (the previous form built both, per view, with torch operations between the warp's backward launches)."""

    @staticmethod
    def forward(ctx, feats, proj_mats, depth_values, G):
        feats = feats.contiguous().float()
        proj_mats, depth_values = proj_mats.detach().contiguous().float(), depth_values.detach().contiguous().float()
        ctx.save_for_backward(feats, proj_mats, depth_values)
        ctx.G = int(G)
        B, V, C, h, w = feats.shape
        if C in (8, 16, 32):   # pixel-major copy: what the fast forward kernels read
            nhwc = ops.nchw_to_nhwc(feats.reshape(B * V, C, h, w)).view(B, V, h, w, C)
            return ops.costvol(nhwc, proj_mats, depth_values, ctx.G, channels_last=True)
        return ops.costvol(feats, proj_mats, depth_values, ctx.G)

    @staticmethod
    def backward(ctx, gvol):
        feats, proj_mats, depth_values = ctx.saved_tensors
        B, V, C, h, w = feats.shape
        G, D = ctx.G, depth_values.shape[1]
        gvol = gvol.contiguous().float()
        gfeats = torch.empty_like(feats)
        ws = _volume_backward_workspace(feats, G, D)
        with torch.cuda.device(feats.device):
            rc = _lib.load().casmvs_costvol_gwc_backward_f32(_ptr(feats), _ptr(proj_mats), _ptr(depth_values), _ptr(gvol), _ptr(gfeats), _ptr(ws),
                                                             B, V, C, G, h, w, D, _stream(feats))
        _lib.check(rc, "casmvs_costvol_gwc_backward_f32")
        return gfeats, None, None, None


def groupwise_volume(feats, proj_mats, depth_values, G):
    """Differentiable mvsnet.py:142-144,157-162,169-172 (G > 1): feats (B,V,C,h,w), proj_mats (B,V-1,3,4), depth_values (B,D,h,w)
    [no grad] -> (B,G,D,h,w)."""
    return _GroupwiseVolume.apply(feats, proj_mats, depth_values, G)


def groupwise_volume_composed(feats, proj_mats, depth_values, G):
    """The same volume as the reference's training code composes it (one differentiable HIP warp per view + torch elementwise ops):
    kept as the test's second opinion on _GroupwiseVolume."""
    B, V, C, h, w = feats.shape
    D = depth_values.shape[1]
    ref = feats[:, 0].unsqueeze(2).expand(-1, -1, D, -1, -1).reshape(B, G, C // G, D, h, w)
    vsum = 0
    for v in range(1, V):
        warped = A.homo_warp(feats[:, v].contiguous(), proj_mats[:, v - 1].contiguous(), depth_values)
        vsum = vsum + warped.reshape(B, G, C // G, D, h, w)
    return (vsum * ref).mean(2).div(V - 1)


# ---- train-mode forwards of the three modules (the eval-mode forwards stay the fused inference engine) -----------------
_KIND2D = {(3, 1, 1): CONV2D_K3, (5, 2, 2): CONV2D_K5S2, (1, 1, 0): CONV2D_K1}
_KIND3D = {(3, 1, 1): CONV_S1, (3, 2, 1): CONV_S2}


def conv_bn_relu_2d(m, x):
    """modules.py:8-18 in train mode."""
    kind = _KIND2D.get(m._geometry)
    if kind is None:
        raise RuntimeError(f"training: ConvBnReLU geometry {m._geometry} is not a FeatureNet layer shape")
    return abn_train(m.bn, conv(x, m.conv.weight, None, kind))


def conv_bn_relu_3d(m, x):
    """modules.py:21-31 in train mode."""
    kind = _KIND3D.get(m._geometry)
    if kind is None:
        raise RuntimeError(f"training: ConvBnReLU3D geometry {m._geometry} is not a CostRegNet layer shape")
    return abn_train(m.bn, conv(x, m.conv.weight, None, kind))


def feature_net_train(net, x):
    """mvsnet.py:40-57 in train mode: x (N,3,H,W) -> {"level_0", "level_1", "level_2"} with an autograd graph."""
    c0 = x
    for m in net.conv0:
        c0 = conv_bn_relu_2d(m, c0)
    c1 = c0
    for m in net.conv1:
        c1 = conv_bn_relu_2d(m, c1)
    c2 = c1
    for m in net.conv2:
        c2 = conv_bn_relu_2d(m, c2)
    feat2 = conv(c2, net.toplayer.weight, net.toplayer.bias, CONV2D_K1)
    feat1 = upsample_add(conv(c1, net.lat1.weight, net.lat1.bias, CONV2D_K1), feat2)
    feat0 = upsample_add(conv(c0, net.lat0.weight, net.lat0.bias, CONV2D_K1), feat1)
    feat1 = conv(feat1, net.smooth1.weight, net.smooth1.bias, CONV2D_K3)
    feat0 = conv(feat0, net.smooth0.weight, net.smooth0.bias, CONV2D_K3)
    return {"level_0": feat0, "level_1": feat1, "level_2": feat2}


def cost_reg_net_train(net, x):
    """mvsnet.py:91-104 in train mode: x (B,Cin,D,h,w) -> (B,1,D,h,w)."""
    conv0 = conv_bn_relu_3d(net.conv0, x)
    conv2 = conv_bn_relu_3d(net.conv2, conv_bn_relu_3d(net.conv1, conv0))
    conv4 = conv_bn_relu_3d(net.conv4, conv_bn_relu_3d(net.conv3, conv2))
    y = conv_bn_relu_3d(net.conv6, conv_bn_relu_3d(net.conv5, conv4))
    y = conv4 + abn_train(net.conv7[1], conv(y, net.conv7[0].weight, None, CONV_T2))
    y = conv2 + abn_train(net.conv9[1], conv(y, net.conv9[0].weight, None, CONV_T2))
    y = conv0 + abn_train(net.conv11[1], conv(y, net.conv11[0].weight, None, CONV_T2))
    return conv(y, net.prob.weight, net.prob.bias, CONV_S1)


def cascade_forward_train(model, imgs, proj_mats, init_depth_min, depth_interval):
    """mvsnet.py:197-244 in train mode (what train.py:99-103 calls): the same loop as the inference forward, on the
    differentiable ops.  Depth hypotheses come from the DETACHED previous depth (mvsnet.py:231)."""
    from .modules import _per_sample
    B, V, _, H, W = imgs.shape
    dev = imgs.device
    plan = pack_plan_of(model)
    set_active_plan(plan)   # this step's forward AND backward take their layer images from the plan
    plan.begin_step()
    feats = feature_net_train(model.feature, imgs.reshape(B * V, 3, H, W).float())
    proj = proj_mats.float()
    results = {}
    depth_l = None
    for l in reversed(range(model.levels)):
        feats_l = feats[f"level_{l}"]
        C, h, w = feats_l.shape[1:]
        feats_l = feats_l.reshape(B, V, C, h, w)
        proj_l = proj[:, :, l].contiguous()
        D = model.n_depths[l]
        ratio = model.interval_ratios[l]
        if isinstance(depth_interval, torch.Tensor):
            interval_b = depth_interval.reshape(B).to(dev, torch.float32) * ratio
            half_b = (D / 2) * interval_b
        else:
            interval_b = _per_sample(depth_interval * ratio, B, dev)
            half_b = _per_sample(D / 2 * (depth_interval * ratio), B, dev)
        with torch.no_grad():
            if l == model.levels - 1:
                depth_values = ops.depth_hypotheses(None, _per_sample(init_depth_min, B, dev), interval_b, None, D, h, w)
            else:
                depth_values = ops.depth_hypotheses(depth_l.detach(), None, interval_b, half_b, D, h, w)
        if model.G == 1:
            volume = variance_volume(feats_l, proj_l, depth_values)
        else:
            volume = groupwise_volume(feats_l, proj_l, depth_values, model.G)
        cost = cost_reg_net_train(getattr(model, f"cost_reg_{l}"), volume).squeeze(1)
        depth_l, confidence_l = A.softmax_depth_regression(cost, depth_values)
        results[f"depth_{l}"] = depth_l
        results[f"confidence_{l}"] = confidence_l
    return results


def sl1_loss(results, depths, masks, levels=3):
    """losses.py:4-19 (SL1Loss): sum over the levels of SmoothL1(depth_l[mask_l], gt_l[mask_l]) * 2^(1 - l).  torch ops on
    (B,h,w) maps - the boolean indexing is a host sync per level, as in the reference."""
    loss = 0
    for l in range(levels):
        m = masks[f"level_{l}"]
        loss = loss + torch.nn.functional.smooth_l1_loss(results[f"depth_{l}"][m], depths[f"level_{l}"][m], reduction="mean") * 2 ** (1 - l)
    return loss


def sl1_loss_masked(results, depths, masks, levels=3):
    """The same loss without the boolean-mask indexing (no host sync, capturable into a hipGraph): the element-wise
    SmoothL1 times the float mask, summed and divided by the mask's count - the mean over the masked elements."""
    loss = 0
    for l in range(levels):
        m = masks[f"level_{l}"].to(results[f"depth_{l}"].dtype)
        e = torch.nn.functional.smooth_l1_loss(results[f"depth_{l}"], depths[f"level_{l}"], reduction="none")
        loss = loss + (e * m).sum() / m.sum() * 2 ** (1 - l)
    return loss


def train_steps(model, batches, optimizer, device="cuda"):
    """train.py:99-103 + Lightning's optimisation step, without Lightning: for every batch (dicts from pipeline.collate
    over pipeline.DTUReader samples in training layout: imgs_u8 / imgs, proj_mats, depths, masks, init_depth_min,
    depth_interval) forward in train mode, SL1 loss, backward, optimizer step.  -> list of loss values."""
    from .pipeline import DevicePrefetcher
    model.train()
    losses = []
    for b in DevicePrefetcher(batches, device, depth=2):
        depths = {k: torch.stack([d[k] for d in b["depths"]]).to(device) for k in b["depths"][0]}
        masks = {k: torch.stack([m[k] for m in b["masks"]]).to(device) for k in b["masks"][0]}
        optimizer.zero_grad(set_to_none=True)
        results = model(b["imgs"], b["proj_mats"], b["init_depth_min"], b["depth_interval"])
        loss = sl1_loss(results, depths, masks)
        loss.backward()
        optimizer.step()
        release_plan()   # the step's graph is consumed
        losses.append(float(loss.detach()))
    return losses
