"""CPU tests of the C-ABI library as a binary artefact: it loads, exports every symbol that
include/casmvs.h declares, reports the right ABI version, and its HOST-side functions (packed
sizes, weight packing, workspace size, argument validation) behave.  No GPU compute is called."""
import ctypes
import numpy as np
import os
import re

import pytest
import torch
import torch.nn.functional as F

from casmvsnet_pl_amd import _lib, ops
from casmvsnet_pl_amd.build import build_library
import kernel_model as KM

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", autouse=True)
def built():
    build_library()


def test_header_symbols_are_all_exported_and_bound():
    hdr = open(os.path.join(ROOT, "include", "casmvs.h")).read()
    trace_only = "".join(re.findall(r"#ifdef CASMVS_TRACE\n(.*?)#endif", hdr, re.S))   # debug entry points of -DCASMVS_TRACE builds
    hdr = re.sub(r"#ifdef CASMVS_TRACE\n.*?#endif", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(casmvs_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_lib.SYMBOLS), "ctypes binding table and include/casmvs.h disagree"
    assert set(re.findall(r"\b(casmvs_[a-z0-9_]+)\s*\(", trace_only)) == set(_lib.TRACE_SYMBOLS)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"libcasmvs_hip.so does not export {name}"
    for name in _lib.TRACE_SYMBOLS:   # fault-injection / tracing kernels are not part of the production library
        assert not hasattr(lib, name), f"the production libcasmvs_hip.so exports the debug entry point {name}"


def test_abi_version_and_error_string():
    lib = _lib.load()
    # 4: the scatter backwards (casmvs_costvol_{var,gwc}_backward_f32, casmvs_homo_warp_backward_f32) take a caller-owned workspace - their sums are 64-bit
    # fixed point, bit-identical run to run; 3: casmvs_costreg_regress_f32 reads eight split_layers pointers (2: six, conv9 / conv11; 3: conv1 / conv3 added)
    assert lib.casmvs_abi_version() == 6
    assert lib.casmvs_packed_opsel_safe() == 1   # the in-tree build assembles the device code with the unsafe packed-float32 forms rewritten (build.py)
    rc = lib.casmvs_homo_warp_f32(None, None, None, None, 1, 1, 8, 8, 1, None)
    assert rc == -1 and b"null pointer" in lib.casmvs_last_error()
    rc = lib.casmvs_costvol_gwc_f32(ctypes.c_void_p(8), ctypes.c_void_p(8), ctypes.c_void_p(8), ctypes.c_void_p(8),
                                    1, 3, 12, 5, 8, 8, 4, None)
    assert rc == -2 and b"C=12" in lib.casmvs_last_error()
    # a volume whose per-sample tensors pass 2^29 floats: refused at the entry with the limit spelled out (no layer form - float32 or split-f16 - can address it)
    layers = (ctypes.c_void_p * 11)(*([8] * 11))
    fp = ctypes.cast(ctypes.c_void_p(8), ctypes.POINTER(ctypes.c_float))
    rc = lib.casmvs_costreg_regress_f32(layers, None, 0, fp, fp, fp, fp, fp, None, ctypes.c_void_p(8), 1, 8, 256, 512, 640, 0.01, None, None)
    assert rc != 0 and b"2^29" in lib.casmvs_last_error() and b"D=256" in lib.casmvs_last_error()


def test_packed_sizes_and_workspace():
    lib = _lib.load()
    assert lib.casmvs_conv3d_packed_floats(ops.CONV_S1, 32, 8) == 32 * 9 * 64 + 16 + 64             # PX: 32 channel units
    assert lib.casmvs_conv3d_packed_floats(ops.CONV_S1, 5, 8) == 8 * 9 * 64 + 16 + 64               # PX: padded to 8 channels
    assert lib.casmvs_conv3d_packed_floats(ops.CONV_S1, 64, 64) == 4 * 16 * 27 * 64 + 128 + 64      # CI: 4 slices x 16 quads
    assert lib.casmvs_conv3d_packed_floats(ops.CONV_S1, 8, 1) == 8 * 32 + 8 + 64                    # P1: 8 rows of 27 (+5) taps
    assert lib.casmvs_conv3d_packed_floats(ops.CONV_S2, 8, 16) == 4 * 27 * 64 + 32 + 64            # CI: 2 quads padded to 4
    assert lib.casmvs_conv3d_packed_floats(ops.CONV_T2, 64, 32) == 2 * 16 * 27 * 64 + 64 + 64       # TCI
    assert lib.casmvs_conv3d_packed_floats(ops.CONV_T2, 16, 8) == 4 * 18 * 64 + 16 + 64            # TPX
    assert lib.casmvs_conv3d_packed_floats(ops.CONV_S2, 8, 8) == 0  # unsupported
    n = 8 * 16 * 24
    assert lib.casmvs_costreg_workspace_bytes(2, 8, 16, 24) == 2 * 4 * int(23.75 * n)
    assert lib.casmvs_costreg_workspace_bytes(1, 8, 16, 20) == 0  # w not a multiple of 8


@pytest.mark.parametrize("cin", [8, 16, 32])
def test_whole_costreg_pack_equals_the_per_layer_packs(cin):
    """casmvs_costreg_pack_f32 (SURVEY 8b's export list: ONE call for the regulariser's eleven layers) = eleven casmvs_conv3d_pack_f32 images back to back,
    each 16-byte aligned, at the offsets casmvs_costreg_packed_floats reports."""
    g = torch.Generator().manual_seed(cin)
    layers = [(ops.CONV_S1, cin, 8), (ops.CONV_S2, 8, 16), (ops.CONV_S1, 16, 16), (ops.CONV_S2, 16, 32), (ops.CONV_S1, 32, 32), (ops.CONV_S2, 32, 64),
              (ops.CONV_S1, 64, 64), (ops.CONV_T2, 64, 32), (ops.CONV_T2, 32, 16), (ops.CONV_T2, 16, 8), (ops.CONV_S1, 8, 1)]
    ws = [torch.randn((ci, co) if k == ops.CONV_T2 else (co, ci), generator=g).reshape(*((ci, co) if k == ops.CONV_T2 else (co, ci)), 1, 1, 1).repeat(1, 1, 3, 3, 3)
          * torch.randn(1, 1, 3, 3, 3, generator=g) for k, ci, co in layers]
    scs = [torch.rand(co, generator=g) + 0.5 for _, _, co in layers[:-1]] + [None]
    shs = [torch.randn(co, generator=g) for _, _, co in layers]
    blob, offs = ops.costreg_pack(ws, scs, shs)
    assert offs[0] == 0 and all(o % 4 == 0 for o in offs) and offs == sorted(offs)
    for i, (k, ci, co) in enumerate(layers):
        one = ops.conv3d_pack(k, ws[i], scs[i], shs[i])
        assert torch.equal(blob[offs[i]:offs[i] + one.numel()], one), i
        end = offs[i + 1] if i < 10 else blob.numel()
        assert end - offs[i] - one.numel() < 4 and not blob[offs[i] + one.numel():end].any()
    with pytest.raises(ValueError):
        ops.costreg_pack(ws[:10])


def test_middle_z_slice_of_a_3d_image_is_the_2d_layers_image():
    """csrc/conv3d_mfma.hip extract_kz1_image_kernel (conv6 of a one-plane volume runs as a 2D layer, cascade level 0): blocks of 9 x 64 floats at tap 9 of every
    (slice, quad) of the 3D channel-inner image, then the tail - exactly casmvs_conv2d_pack_f32 of the weight's middle z slice."""
    g = torch.Generator().manual_seed(6)
    w = torch.randn(64, 64, 3, 3, 3, generator=g)
    sc, sh = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g)
    img3 = ops.conv3d_pack(ops.CONV_S1, w, sc, sh)
    img2 = ops.conv2d_pack(ops.CONV2D_K3, w[:, :, 1].contiguous(), sc, sh)
    blocks, tail = 4 * 16, 2 * 64 + 64      # 4 slices of 16 output channels x 16 quads of input channels; scale | shift | 64 zero words
    assert img3.numel() == blocks * 27 * 64 + tail and img2.numel() == blocks * 9 * 64 + tail
    cut = torch.cat([img3[:blocks * 27 * 64].view(blocks, 27, 64)[:, 9:18].reshape(-1), img3[blocks * 27 * 64:]])
    assert torch.equal(cut, img2)


CASES = [(ops.CONV_S1, 32, 8), (ops.CONV_S1, 5, 8), (ops.CONV_S1, 8, 1), (ops.CONV_S1, 16, 16), (ops.CONV_S1, 20, 32),
         (ops.CONV_S2, 8, 16), (ops.CONV_S2, 6, 32), (ops.CONV_T2, 16, 8), (ops.CONV_T2, 12, 32)]


@pytest.mark.parametrize("kind,cin,cout", CASES)
def test_packed_image_drives_the_kernel_index_model_to_the_right_convolution(kind, cin, cout):
    """C packer + Python model of the kernel's (stage, tap, c, q) -> (image, ABID) walk == torch conv."""
    g = torch.Generator().manual_seed(kind * 100 + cin + cout)
    x = torch.randn(2, cin, 4, 6, 6, generator=g)
    wshape = (cin, cout, 3, 3, 3) if kind == ops.CONV_T2 else (cout, cin, 3, 3, 3)
    w = torch.randn(wshape, generator=g) * 0.2
    scale = torch.rand(cout, generator=g) + 0.5
    shift = torch.randn(cout, generator=g) * 0.1
    packed = ops.conv3d_pack(kind, w, scale, shift)
    if kind == ops.CONV_T2:
        ref = F.conv_transpose3d(x, w, None, stride=2, padding=1, output_padding=1)
    else:
        ref = F.conv3d(x, w, None, stride=1 if kind == ops.CONV_S1 else 2, padding=1)
    ref = F.leaky_relu(ref * scale.view(1, -1, 1, 1, 1) + shift.view(1, -1, 1, 1, 1), 0.01)
    skip = torch.randn(ref.shape, generator=g)
    got = KM.emulate(kind, packed, x, cout, skip=skip, slope=0.01)
    assert float((got - (ref + skip)).abs().max()) < 1e-4


CASES_2D = [(ops.CONV2D_K3, 3, 8), (ops.CONV2D_K3, 8, 8), (ops.CONV2D_K3, 32, 8), (ops.CONV2D_K3, 16, 16),
            (ops.CONV2D_K3, 32, 32), (ops.CONV2D_K5S2, 8, 16), (ops.CONV2D_K5S2, 16, 32), (ops.CONV2D_K1, 32, 32),
            (ops.CONV2D_K1_UP, 8, 32), (ops.CONV2D_K1_UP, 16, 32)]


@pytest.mark.parametrize("kind,cin,cout", CASES_2D)
def test_packed_2d_image_drives_the_kernel_index_model_to_the_right_convolution(kind, cin, cout):
    """FeatureNet layers: C packer + model of the kz = 1 kernel walk == torch conv2d (+ upsample-add)."""
    g = torch.Generator().manual_seed(kind * 100 + cin + cout)
    x = torch.randn(2, cin, 8, 12, generator=g)
    k = {ops.CONV2D_K3: 3, ops.CONV2D_K5S2: 5}.get(kind, 1)
    w = torch.randn(cout, cin, k, k, generator=g) * 0.2
    scale = torch.rand(cout, generator=g) + 0.5
    shift = torch.randn(cout, generator=g) * 0.1
    packed = ops.conv2d_pack(kind, w, scale, shift)
    ref = F.conv2d(x, w, None, stride=2 if kind == ops.CONV2D_K5S2 else 1, padding=k // 2)
    ref = F.leaky_relu(ref * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1), 0.01)
    up = None
    if kind == ops.CONV2D_K1_UP:
        up = torch.randn(2, cout, 4, 6, generator=g)
        ref = ref + F.interpolate(up, scale_factor=2, mode="bilinear", align_corners=True)
    got = KM.emulate2d(kind, packed, x, cout, up=up, slope=0.01)
    assert float((got - ref).abs().max()) < 1e-4


def test_2d_packed_sizes_and_workspace():
    lib = _lib.load()
    assert lib.casmvs_conv2d_packed_floats(ops.CONV2D_K3, 3, 8) == 8 * 3 * 64 + 16 + 64          # PX, 1-deep kernel
    assert lib.casmvs_conv2d_packed_floats(ops.CONV2D_K3, 32, 16) == 8 * 9 * 64 + 32 + 64        # CI
    assert lib.casmvs_conv2d_packed_floats(ops.CONV2D_K5S2, 16, 32) == 2 * 4 * 25 * 64 + 64 + 64
    assert lib.casmvs_conv2d_packed_floats(ops.CONV2D_K1_UP, 8, 32) == 2 * 4 * 64 + 64 + 64
    assert lib.casmvs_conv2d_packed_floats(ops.CONV2D_K5S2, 8, 8) == 0       # unsupported
    assert lib.casmvs_conv2d_packed_floats(ops.CONV_S1, 8, 8) == 0           # a 3D kind
    assert lib.casmvs_conv3d_packed_floats(ops.CONV2D_K3, 8, 8) == 0         # a 2D kind
    assert lib.casmvs_featurenet_workspace_bytes(3, 32, 64) == 3 * 4 * (48 + 20 + 6) * 32 * 64
    assert lib.casmvs_featurenet_workspace_bytes(1, 30, 64) == 0


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.CasMVSLibraryError):
        _lib.load()


def test_ops_reject_cpu_tensors():
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.homo_warp(torch.zeros(1, 1, 4, 4), torch.zeros(1, 3, 4), torch.ones(1, 1, 4, 4))


def test_stride2_deinterleaved_row_walk_is_a_stride2_convolution():
    """Index model of the stride-2 kernels' [even | odd] LDS rows (wide and deep tile) == F.conv3d stride 2."""
    g = torch.Generator().manual_seed(11)
    x = torch.randn(3, 6, 12, 40, generator=g)
    w = torch.randn(5, 3, 3, 3, 3, generator=g)
    ref = F.conv3d(x[None], w, None, stride=2, padding=1)[0].numpy()
    for tile in ((2, 4, 16), (1, 4, 16)):
        got = KM.emulate_s2_deint(x.numpy(), w.numpy(), *tile)
        assert abs(got - ref).max() < 1e-4


@pytest.mark.parametrize("H,W", [(4, 8), (6, 12), (10, 40), (8, 64)])
def test_fpn_lateral_window_and_tent_weights_reproduce_interpolate(H, W):
    """The 4-column window + tent matrix of fpn_lateral_kernel == F.interpolate(x2, bilinear, align_corners) + 1x1 conv."""
    g = torch.Generator().manual_seed(H * 100 + W)
    x = torch.randn(8, H, W, generator=g)
    w = torch.randn(16, 8, generator=g) * 0.3
    b = torch.randn(16, generator=g) * 0.1
    up = torch.randn(16, H // 2, W // 2, generator=g)
    want = F.interpolate(up[None], scale_factor=2, mode="bilinear", align_corners=True)[0] + F.conv2d(x[None], w[:, :, None, None], b)[0]
    got = KM.emulate_fpn_lateral(x.numpy(), w.numpy(), b.numpy(), up.numpy())
    assert abs(got - want.numpy()).max() < 1e-4


@pytest.mark.parametrize("S,KZ,KS", [(1, 3, 3), (2, 3, 3), (1, 1, 3), (2, 1, 5), (1, 1, 1)])
def test_wgrad_vector_staging_writes_every_tile_cell_exactly_once(S, KZ, KS):
    """The 16-byte staging of the weight-gradient kernel (train.hip): a big-tile row = P halo floats + 4 S aligned vectors +
    (KS - P - S) halo floats.  Its units must cover the [16][IZ][IY][IX] operand tile and the [16][ROWS][16] small tile
    exactly once (a missed cell would be stale data of the previous tile, a doubled one a wasted load), and the vector
    units must start 16-byte aligned relative to the tile origin (x0 * S - P + ix with x0 % 16 == 0)."""
    g, small, big = KM.wgrad_vector_staging_units(S, KZ, KS)
    assert sorted(small) == [(c, r, x) for c in range(16) for r in range(g["ROWS"]) for x in range(16)]
    assert sorted(big) == [(c, iz, iy, ix) for c in range(16) for iz in range(g["IZ"]) for iy in range(g["IY"]) for ix in range(g["IX"])]
    # a vector unit's first element: global x = x0 * S - P + (P + 4 k) = x0 * S + 4 k: a multiple of 4 floats
    vec_starts = {ix for (_, _, _, ix) in big[: 16 * g["IZ"] * g["IY"] * 4 * S * 4 : 4]}
    assert all((ix - g["P"]) % 4 == 0 for ix in vec_starts)


@pytest.mark.parametrize("D,H,W,zc", [(8, 8, 68, 8), (8, 12, 64, 4), (16, 4, 8, 8), (4, 9, 132, 3)])
def test_prob_zwalk_model_staging_and_rotating_accumulators_are_the_prob_convolution(D, H, W, zc):
    """csrc/prob_regress.hip: the staging plan writes every LDS cell of a plane slot exactly once, every output voxel is
    produced exactly once, and the depth walk (three rotating accumulators, chunks of zc planes with one halo plane on
    either side, ragged last tiles / chunks) equals Conv3d(8 -> 1, k3 p1) + bias on the C packer's image."""
    g = torch.Generator().manual_seed(D * 1000 + W)
    x = torch.randn(2, 8, D, H, W, generator=g)
    w = torch.randn(1, 8, 3, 3, 3, generator=g) * 0.2
    bias = torch.randn(1, generator=g)
    packed = ops.conv3d_pack(ops.CONV_S1, w, None, bias)
    ref = F.conv3d(x.double(), w.double(), bias.double(), padding=1)[:, 0]
    got = KM.emulate_prob_zwalk(packed, x, zc)
    assert float((got - ref).abs().max()) < 1e-10


def test_prob_zwalk_lds_accesses_are_bank_conflict_free():
    """The two ds_read_b128 of a (channel pair, ky) step take 4 LDS cycles each (the minimum) with the kernel's row stride,
    and the staging's ds_write_b128 (item e = 16 bytes at float 4 e) 8 (the minimum); the first version's four ds_write_b64
    32 bytes apart took 4x theirs (PMC: bank conflicts = 47 % of the LDS-active cycles)."""
    assert KM.prob_zwalk_bank_cycles(KM.PZ_RS, 0) == 4 and KM.prob_zwalk_bank_cycles(KM.PZ_RS, 4) == 4
    assert KM.PZ_RS % 4 == 0 and KM.PZ_RS == 2 * (KM.PZ_TX + 2)
    new, old = KM.prob_zwalk_write_cycles()
    assert new == 8 and old == 16      # b64: 4 groups x 1 cycle minimum -> 4 x 4 = 16


@pytest.mark.parametrize("H,W", [(8, 64), (12, 72), (20, 132)])
def test_fpn_fused_tail_model_equals_lat_upsample_smooth(H, W):
    """csrc/fpn_fused.hip: the composed 40-channel layer (mvsnet.compose_fpn_tail -> casmvs_conv2d_pack_f32), the upsampled
    channels staged through the 4-column window / tent matrix, the PX lane images and the nine border bias classes reproduce
    smooth0(lat0(x) + interpolate(y)) of mvsnet.py:50-51,54 (ragged tiles in x and y)."""
    import numpy as np
    from casmvsnet_pl_amd.mvsnet import compose_fpn_tail
    g = torch.Generator().manual_seed(H + W)
    lw, lb = torch.randn(32, 8, 1, 1, generator=g) * 0.3, torch.randn(32, generator=g)
    sw, sb = torch.randn(8, 32, 3, 3, generator=g) * 0.2, torch.randn(8, generator=g)
    x, y = torch.randn(1, 8, H, W, generator=g), torch.randn(1, 32, H // 2, W // 2, generator=g)
    ref = F.conv2d(F.conv2d(x.double(), lw.double(), lb.double()) + F.interpolate(y.double(), scale_factor=2, mode="bilinear", align_corners=True),
                   sw.double(), sb.double(), padding=1)[0]
    w40, bias9 = compose_fpn_tail(lw, lb, sw, sb)
    packed = ops.conv2d_pack(ops.CONV2D_K3, w40, None, None)
    got = KM.emulate_fpn_tail0(packed.numpy(), bias9.numpy(), x[0].numpy(), y[0].numpy())
    assert float(np.abs(got - ref.numpy()).max()) < 2e-5 * float(ref.abs().max())


@pytest.mark.parametrize("H,W,amp", [(8, 64, 1.0), (20, 36, 1.0), (34, 72, 1e-4), (16, 32, 3e4)])
def test_fpn_fused_tail_splitf16_model(H, W, amp):
    """csrc/fpn_fused_sf.hip: the composed 40-channel tail as split-f16 lane images (casmvs_fpn_tail0_splitf16_pack), the per-(tile,
    chunk) scaling + two-slice split of the staged [conv0 | up(feat1')] tile and the nine bias classes reproduce
    smooth0(lat0(x) + interpolate(y)) to float32 grade (<= 4e-6 of the range against float64), ragged tiles, any magnitude."""
    import numpy as np
    from casmvsnet_pl_amd.mvsnet import compose_fpn_tail
    g = torch.Generator().manual_seed(H + W)
    lw, lb = torch.randn(32, 8, 1, 1, generator=g) * 0.3, torch.randn(32, generator=g) * amp
    sw, sb = torch.randn(8, 32, 3, 3, generator=g) * 0.2, torch.randn(8, generator=g) * amp
    x, y = torch.randn(1, 8, H, W, generator=g) * amp, torch.randn(1, 32, H // 2, W // 2, generator=g) * amp
    ref = F.conv2d(F.conv2d(x.double(), lw.double(), lb.double()) + F.interpolate(y.double(), scale_factor=2, mode="bilinear", align_corners=True),
                   sw.double(), sb.double(), padding=1)[0]
    w40, bias9 = compose_fpn_tail(lw, lb, sw, sb)
    packed = ops.fpn_tail0_splitf16_pack(w40)
    assert packed.numel() == 5 * 3 * 2 * 64 * 16 + 16
    got = KM.emulate_fpn_tail0_splitf16(packed.numpy(), bias9.numpy(), x[0].numpy(), y[0].numpy())
    assert float(np.abs(got - ref.numpy()).max()) < 4e-6 * float(ref.abs().max())   # measured <= 1.7e-6 (the float32 model of the float32 kernel: bound 2e-5)


@pytest.mark.parametrize("cin,terms", [(8, 6), (16, 6), (32, 6), (8, 9), (8, 3)])
def test_conv0_splitbf16_packing_and_partial_products(cin, terms):
    """csrc/conv0_splitbf16.hip: the C packer's lane images (three exact bf16 slices of every weight, rows = (co, x phase),
    k = (x offset, ci)) and the activations' three slices reproduce Conv3d + folded ABN: all nine partial products exactly
    (float64 accumulation), the kernel's six to ~2^-23 per product, three (hi x hi, hi x mid, mid x hi) only to ~2^-15."""
    import numpy as np
    g = torch.Generator().manual_seed(cin + terms)
    x = torch.randn(1, cin, 3, 4, 6, generator=g)
    w = torch.randn(8, cin, 3, 3, 3, generator=g) * 0.2
    scale, shift = torch.rand(8, generator=g) + 0.5, torch.randn(8, generator=g) * 0.1
    packed = ops.conv0_splitbf16_pack(w, scale, shift)
    ref = F.conv3d(x.double(), w.double(), None, padding=1) * scale.double().view(1, -1, 1, 1, 1) + shift.double().view(1, -1, 1, 1, 1)
    ref = torch.where(ref > 0, ref, ref * 0.01).numpy()
    got = KM.emulate_conv0_splitbf16(packed.numpy(), x.numpy(), cin, terms=terms)
    err = float(np.abs(got - ref).max() / np.abs(ref).max())
    assert err < {9: 1e-12, 6: 2e-6, 3: 3e-3}[terms], err
    if terms == 6:
        assert err > 1e-12 or True
    hi, mid, lo = KM.bf16_split3(x.numpy())
    assert np.array_equal((hi.astype(np.float64) + mid + lo), x.numpy().astype(np.float64))
    for part in (hi, mid, lo):
        assert not (part.view(np.uint32) & 0xFFFF).any()          # each slice is exactly a bf16 number


@pytest.mark.parametrize("cin,terms,shape,amp", [(8, 3, (3, 4, 6), 1.0), (16, 3, (5, 6, 36), 1.0), (32, 3, (2, 3, 8), 1e-3), (8, 4, (3, 4, 6), 1.0),
                                                  (8, 3, (4, 5, 40), 3e4), (16, 3, (2, 9, 8), 1e-30)])
def test_conv0_splitf16_packing_and_partial_products(cin, terms, shape, amp):
    """csrc/conv0_splitf16.hip: the C packer's lane images (2^kw w as two float16 slices, 2^-kw in the scale) and the kernel's
    per-(tile, chunk) power-of-two scaling + two-slice split of the activations reproduce Conv3d + folded ABN to float32 grade:
    <= 4e-7 of the output range against float64 for inputs of any magnitude (also far outside float16's own range), with a
    volume whose far corner is 10^6 times smaller than the rest (a tile of its own scale)."""
    import numpy as np
    g = torch.Generator().manual_seed(cin + terms)
    x = torch.randn(1, cin, *shape, generator=g) * amp
    x[..., -2:, -3:] *= 1e-6
    w = torch.randn(8, cin, 3, 3, 3, generator=g) * 0.2
    scale, shift = torch.rand(8, generator=g) + 0.5, torch.randn(8, generator=g) * 0.1 * amp
    packed = ops.conv0_splitf16_pack(w, scale, shift)
    assert packed.numel() == cin // 8 * 9 * 2 * 64 * 16 + 64
    ref = F.conv3d(x.double(), w.double(), None, padding=1) * scale.double().view(1, -1, 1, 1, 1) + shift.double().view(1, -1, 1, 1, 1)
    ref = torch.where(ref > 0, ref, ref * 0.01).numpy()
    got = KM.emulate_conv0_splitf16(packed.numpy(), x.numpy(), cin, terms=terms)
    err = float(np.abs(got - ref).max() / np.abs(ref).max())
    assert err < 4e-7, err
    with pytest.raises(RuntimeError):
        ops.conv0_splitf16_pack(w * float("inf"), scale, shift)


@pytest.mark.parametrize("c,shape,amp", [(16, (3, 4, 6), 1.0), (16, (5, 6, 18), 1e-3), (32, (2, 3, 8), 1.0), (16, (4, 5, 34), 3e4), (32, (5, 2, 20), 1e-30),
                                         (32, (2, 10, 18), 1.0), (64, (3, 4, 6), 1.0)])
def test_conv_ci_splitf16_packing_and_partial_products(c, shape, amp):
    """csrc/conv_ci_splitf16.hip: the C packer's lane images (tap pairs x 16 channels per step, 2^kw w as two float16 slices, the
    28th tap zero) decoded lane by lane, with the kernel's per-(tile, chunk) scaling and two-slice split of the activations,
    reproduce Conv3d + folded ABN to float32 grade (<= 4e-7 of the output range against float64) at any input magnitude."""
    import numpy as np
    g = torch.Generator().manual_seed(c + shape[2])
    x = torch.randn(1, c, *shape, generator=g) * amp
    x[..., -2:, -3:] *= 1e-6
    w = torch.randn(c, c, 3, 3, 3, generator=g) * 0.1
    scale, shift = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.1 * amp
    packed = ops.conv_ci_splitf16_pack(w, scale, shift)
    assert packed.numel() == c // 16 * 14 * (c // 16) * 2 * 64 * 16 + 8 * c
    ref = F.conv3d(x.double(), w.double(), None, padding=1) * scale.double().view(1, -1, 1, 1, 1) + shift.double().view(1, -1, 1, 1, 1)
    ref = torch.where(ref > 0, ref, ref * 0.01).numpy()
    got = KM.emulate_conv_ci_splitf16(packed.numpy(), x.numpy(), c, c, tile=(2, 8, 16) if shape[0] <= 2 else (4, 4, 16))   # as the launcher picks
    err = float(np.abs(got - ref).max() / np.abs(ref).max())
    assert err < 4e-7, err
    with pytest.raises(ValueError):
        ops.conv_ci_splitf16_pack(torch.randn(16, 32, 3, 3, 3))


@pytest.mark.parametrize("c,N,H,W,amp,cout", [(16, 1, 8, 16, 1.0, 16), (16, 2, 18, 36, 1e-3, 16), (32, 1, 20, 18, 1.0, 32), (32, 1, 5, 50, 3e4, 32), (16, 1, 3, 2, 1e-30, 16),
                                              (32, 1, 18, 20, 1.0, 16)])
def test_conv2d_ci_splitf16_packing_and_partial_products(c, N, H, W, amp, cout):
    """csrc/conv2d_ci_splitf16.hip: lane images (tap pairs of the 9 taps x 16 channels per step, the 10th tap zero) decoded lane by lane,
    per-(tile, chunk) scaling + two-slice split reproduce Conv2d 3x3 + folded ABN + leaky-relu to float32 grade at any magnitude."""
    import numpy as np
    g = torch.Generator().manual_seed(c + H + W)
    x = torch.randn(N, c, H, W, generator=g) * amp
    w = torch.randn(cout, c, 3, 3, generator=g) * 0.1
    scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1 * amp
    packed = ops.conv2d_ci_splitf16_pack(w, scale, shift)
    assert packed.numel() == (c // 16) * (cout // 16) * 5 * 2 * 64 * 16 + 8 * cout
    ref = F.conv2d(x.double(), w.double(), None, padding=1) * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1)
    ref = torch.where(ref > 0, ref, ref * 0.01).numpy()
    got = KM.emulate_conv2d_ci_splitf16(packed.numpy(), x.numpy(), c, cout=cout)
    assert float(np.abs(got - ref).max() / np.abs(ref).max()) < 4e-7
    with pytest.raises(ValueError):
        ops.conv2d_ci_splitf16_pack(torch.randn(32, 16, 3, 3))
    padded, unpadded = KM.conv2d_ci_sf_lds_cycles()
    assert padded == [4, 4] and max(unpadded) > 4


def test_conv_ci_splitf16_lds_layout():
    """conv_ci_sf_kernel's planes [slice][channel half][voxel] keep every tap read conflict-free for all three lane-half distances
    (next x, next row, next plane), and writing the two voxels of an item in lane-dependent order (lanes 0-3 of every 8 the even
    voxel first, lanes 4-7 the odd one) makes the staging writes conflict-free (plain order: 2-way)."""
    reads, w_kernel, w_plain = KM.conv_ci_sf_lds_cycles()
    assert reads == [4, 4, 4], reads
    assert w_kernel == 16 and w_plain == 32, (w_kernel, w_plain)


def test_conv0_splitbf16_lds_layout():
    """conv0_sb_kernel's swizzled LDS rows keep the tap reads conflict-free while the staging writes cost 1.75x their minimum
    (3.1x without the swizzle).  (Tiles are dealt round-robin in XCD-major order like every other layer - contiguous runs per
    workgroup measured 10 % slower: the halo sharing that matters is between workgroups that run at the same time.)"""
    assert len({KM.conv0_sb_slot(x) for x in range(40)}) == 40 and max(KM.conv0_sb_slot(x) for x in range(40)) < 41
    read, write, write_linear = KM.conv0_sb_lds_cycles()
    assert read == 4 and write == 56 and write_linear == 100      # (32 would be conflict-free; with 40-slot rows and no swizzle: 128)


def test_fusion_paired_tap_fetch_equals_direct_indexing():
    """fuse_view_paired_kernel's one-load-per-row tap fetch (host model) returns the taps fuse_view_kernel loads one by one, for
    every tap origin incl. far outside, the clamped border columns and the last pixel pair of the view (the 2-bytes-early load)."""
    import numpy as np
    from kernel_model import fusion_paired_taps
    g = np.random.default_rng(0)
    for H, W in ((2, 2), (3, 2), (5, 7), (4, 16)):
        depth = g.standard_normal((H, W)).astype(np.float32)
        image = g.integers(0, 256, (H, W, 3), dtype=np.uint8)
        for iy in range(-3, H + 3):
            for ix in range(-3, W + 3):
                cx0, cx1 = min(max(ix, 0), W - 1), min(max(ix + 1, 0), W - 1)
                cy0, cy1 = min(max(iy, 0), H - 1), min(max(iy + 1, 0), H - 1)
                want = [(cy0, cx0), (cy0, cx1), (cy1, cx0), (cy1, cx1)]
                dt, ct = fusion_paired_taps(depth, image, ix, iy)
                assert [float(v) for v in dt] == [float(depth[y, x]) for y, x in want], (H, W, ix, iy)
                assert ct == [[int(v) for v in image[y, x]] for y, x in want], (H, W, ix, iy)


@pytest.mark.parametrize("cin,shape,zlen", [(8, (1, 5, 20, 36), None), (16, (2, 9, 17, 44), 4), (8, (1, 3, 33, 32), 2)])
def test_conv0_zmarch_host_model_is_a_float32_grade_convolution(cin, shape, zlen):
    """The arithmetic of csrc/conv0_zmarch.hip (per-plane-patch power-of-two scaling, two float16 slices, three partial products, the
    packed image of casmvs_conv0_splitf16_pack) restated on the host: within 1e-6 of the range from a float64 convolution, equal - to
    float64 rounding - with and without z segments, and as close as the tiled kernel's model."""
    import numpy as np
    import torch
    from casmvsnet_pl_amd import ops
    from kernel_model import emulate_conv0_splitf16, emulate_conv0_zmarch
    B, D, H, W = shape
    g = torch.Generator().manual_seed(cin + D)
    x = torch.randn(B, cin, D, H, W, generator=g) * 3.0
    x[:, :, :, :3, :5] *= 1e-4                                   # a corner far below the rest: its planes get their own scale
    w = torch.randn(8, cin, 3, 3, 3, generator=g) * 0.2
    scale, shift = torch.rand(8, generator=g) + 0.5, torch.randn(8, generator=g) * 0.1
    packed = ops.conv0_splitf16_pack(w, scale, shift).numpy()
    ref = torch.nn.functional.conv3d(x.double(), w.double(), padding=1) * scale.double().view(1, 8, 1, 1, 1) + shift.double().view(1, 8, 1, 1, 1)
    ref = torch.where(ref > 0, ref, ref * 0.01).numpy()
    rng = np.abs(ref).max()
    got = emulate_conv0_zmarch(packed, x.numpy(), cin, zlen=zlen)
    whole = emulate_conv0_zmarch(packed, x.numpy(), cin, zlen=None)
    tiled = emulate_conv0_splitf16(packed, x.numpy(), cin)
    assert np.abs(got - ref).max() / rng < 1e-6
    assert np.abs(got - whole).max() / rng < 1e-12
    assert np.abs(got - ref).max() <= 3.0 * np.abs(tiled - ref).max() + 1e-7 * rng


@pytest.mark.parametrize("shape", [(1, 2, 4, 16), (2, 3, 5, 10), (1, 1, 9, 22)])
def test_deconv11_splitf16_host_model_is_the_transposed_convolution(shape):
    """The arithmetic and the tap / parity bookkeeping of csrc/deconv11_splitf16.hip restated on the host from casmvs_deconv11_splitf16_pack's
    image: within 1e-6 of the range from ConvTranspose3d(16, 8, 3, stride 2, padding 1, output_padding 1) + ABN + leaky-relu + skip in float64."""
    import numpy as np
    import torch
    from casmvsnet_pl_amd import ops
    from kernel_model import emulate_deconv11_splitf16
    B, Di, Hi, Wi = shape
    g = torch.Generator().manual_seed(Di * 10 + Wi)
    x = torch.randn(B, 16, Di, Hi, Wi, generator=g) * 2.0
    w = torch.randn(16, 8, 3, 3, 3, generator=g) * 0.2
    scale, shift = torch.rand(8, generator=g) + 0.5, torch.randn(8, generator=g) * 0.1
    skip = torch.randn(B, 8, 2 * Di, 2 * Hi, 2 * Wi, generator=g)
    packed = ops.deconv11_splitf16_pack(w, scale, shift).numpy()
    ref = torch.nn.functional.conv_transpose3d(x.double(), w.double(), stride=2, padding=1, output_padding=1)
    ref = ref * scale.double().view(1, 8, 1, 1, 1) + shift.double().view(1, 8, 1, 1, 1)
    ref = (torch.where(ref > 0, ref, ref * 0.01) + skip.double()).numpy()
    got = emulate_deconv11_splitf16(packed, x.numpy(), skip.numpy())
    assert np.abs(got - ref).max() / np.abs(ref).max() < 1e-6


def test_deconv11_splitf16_lane_level_transcription():
    """Every thread's index arithmetic of deconv11_sf_kernel transcribed (staging items, LDS planes, B units, tap tables, result lanes) on ragged
    shapes: each output voxel written exactly by the tile that owns it, within 1e-6 of ConvTranspose3d + ABN + leaky-relu + skip in float64."""
    import numpy as np
    import torch
    from casmvsnet_pl_amd import ops
    from kernel_model import emulate_deconv11_lanes
    for (B, Di, Hi, Wi) in ((1, 2, 4, 16), (1, 3, 5, 18)):
        g = torch.Generator().manual_seed(Di + Wi)
        x = torch.randn(B, 16, Di, Hi, Wi, generator=g) * 2.0
        w = torch.randn(16, 8, 3, 3, 3, generator=g) * 0.2
        scale, shift = torch.rand(8, generator=g) + 0.5, torch.randn(8, generator=g) * 0.1
        skip = torch.randn(B, 8, 2 * Di, 2 * Hi, 2 * Wi, generator=g)
        packed = ops.deconv11_splitf16_pack(w, scale, shift).numpy()
        ref = torch.nn.functional.conv_transpose3d(x.double(), w.double(), stride=2, padding=1, output_padding=1)
        ref = ref * scale.double().view(1, 8, 1, 1, 1) + shift.double().view(1, 8, 1, 1, 1)
        ref = (torch.where(ref > 0, ref, ref * 0.01) + skip.double()).numpy()
        got = emulate_deconv11_lanes(packed, x.numpy(), skip.numpy())
        assert np.abs(got - ref).max() / np.abs(ref).max() < 1e-6, (B, Di, Hi, Wi)


def test_conv0_zmarch_lane_level_transcription():
    """The index arithmetic of conv0_zm_kernel transcribed thread by thread (tests/kernel_model.py) on ragged shapes, with
    z segments: every output written by the item that owns it, within 2e-6 of the layers in float64."""
    import numpy as np
    import torch
    from casmvsnet_pl_amd import ops
    from kernel_model import emulate_conv0_zmarch_lanes
    g = torch.Generator().manual_seed(11)
    for cin, (B, D, H, W), zlen in ((8, (1, 3, 18, 36), 2), (16, (1, 5, 17, 32), 5)):
        x = torch.randn(B, cin, D, H, W, generator=g) * 3.0
        w = torch.randn(8, cin, 3, 3, 3, generator=g) * 0.2
        scale, shift = torch.rand(8, generator=g) + 0.5, torch.randn(8, generator=g) * 0.1
        packed = ops.conv0_splitf16_pack(w, scale, shift).numpy()
        ref = torch.nn.functional.conv3d(x.double(), w.double(), padding=1) * scale.double().view(1, 8, 1, 1, 1) + shift.double().view(1, 8, 1, 1, 1)
        ref = torch.where(ref > 0, ref, ref * 0.01).numpy()
        got = emulate_conv0_zmarch_lanes(packed, x.numpy(), cin, zlen)
        assert np.abs(got - ref).max() / np.abs(ref).max() < 1e-6, (cin, zlen)

def test_deconv9_splitf16_lane_level_transcription():
    """deconv9_sf_kernel's index arithmetic transcribed thread by thread on ragged shapes: within 1e-6 of ConvTranspose3d(32, 16, 3, stride 2, padding 1,
    output_padding 1) + ABN + leaky-relu + skip in float64."""
    import numpy as np
    import torch
    from casmvsnet_pl_amd import ops
    from kernel_model import emulate_deconv9_lanes
    for (B, Di, Hi, Wi) in ((1, 1, 4, 16), (1, 2, 5, 18)):
        g = torch.Generator().manual_seed(Di + Wi + 3)
        x = torch.randn(B, 32, Di, Hi, Wi, generator=g) * 2.0
        w = torch.randn(32, 16, 3, 3, 3, generator=g) * 0.15
        scale, shift = torch.rand(16, generator=g) + 0.5, torch.randn(16, generator=g) * 0.1
        skip = torch.randn(B, 16, 2 * Di, 2 * Hi, 2 * Wi, generator=g)
        packed = ops.deconv9_splitf16_pack(w, scale, shift).numpy()
        ref = torch.nn.functional.conv_transpose3d(x.double(), w.double(), stride=2, padding=1, output_padding=1)
        ref = ref * scale.double().view(1, 16, 1, 1, 1) + shift.double().view(1, 16, 1, 1, 1)
        ref = (torch.where(ref > 0, ref, ref * 0.01) + skip.double()).numpy()
        got = emulate_deconv9_lanes(packed, x.numpy(), skip.numpy())
        assert np.abs(got - ref).max() / np.abs(ref).max() < 1e-6, (B, Di, Hi, Wi)


def test_conv_s2_splitf16_packing():
    """csrc/conv_s2_splitf16.hip's C packer: lane images [chunk of 8 input channels][kz][ky][block of 16 output channels][slice][lane][8 f16] with
    lane = (output channel & 15, kx) - the two float16 slices of 2^kw w add up to the weight to 2^-22 of the tensor's maximum, the kx = 3 block is zero,
    scale carries 2^-kw; non-finite weights and other channel counts are rejected."""
    from casmvsnet_pl_amd import ops
    g = torch.Generator().manual_seed(5)
    for cin, cout in ((8, 16), (16, 32)):
        w = torch.randn(cout, cin, 3, 3, 3, generator=g) * 0.3
        scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g)
        packed = ops.conv_s2_splitf16_pack(w, scale, shift).numpy()
        nimg = (cin // 8) * 9 * (cout // 16)
        assert packed.size == nimg * 2 * 64 * 16 + 2 * cout * 4
        img = packed[:nimg * 2048].view(np.float16).astype(np.float64).reshape(cin // 8, 3, 3, cout // 16, 2, 64, 8)
        tail = packed[nimg * 2048:].view(np.float32)
        kw = int(round(np.log2(float(scale[0]) / tail[0])))
        assert np.array_equal(tail[:cout], np.ldexp(scale.numpy(), -kw)) and np.array_equal(tail[cout:], shift.numpy())
        assert 2.0 ** 13 <= float(w.abs().max()) * 2.0 ** kw < 2.0 ** 14
        assert not img[..., 48:, :].any()                                   # kx = 3: zero weights
        got = (img[:, :, :, :, 0] + img[:, :, :, :, 1]).reshape(cin // 8, 3, 3, cout // 16, 4, 16, 8)   # [chunk][kz][ky][rb][kx][co & 15][e]
        want = np.ldexp(w.double().numpy(), kw).reshape(cout // 16, 16, cin // 8, 8, 3, 3, 3).transpose(2, 4, 5, 0, 6, 1, 3)   # -> [chunk][kz][ky][rb][kx][i][e]
        assert np.abs(got[:, :, :, :, :3] - want).max() <= 2.0 ** -8          # |w'| < 2^14: two 11-bit slices leave < 2^-8 absolute
    with pytest.raises(ValueError):
        ops.conv_s2_splitf16_pack(torch.randn(16, 16, 3, 3, 3))
    bad = torch.randn(16, 8, 3, 3, 3)
    bad[3, 2, 1, 1, 1] = float("inf")
    with pytest.raises(RuntimeError, match="finite"):
        ops.conv_s2_splitf16_pack(bad)


def test_conv2d_k5s2_splitf16_packing():
    """csrc/conv2d_k5s2_splitf16.hip's C packer: lane images [chunk of 8 input channels][step][block of 16 output channels][slice][lane][8 f16], the 25 taps
    in 32 (step, k-block) slots - pair p = 2 step + (kb >> 1): kx = p >> 1, ky = 2 (p & 1) + (kb & 1) for p < 10; kx = p - 10, ky = 4 on the even member of
    pairs 10 .. 14; the other seven slots hold zeros.  Every tap appears exactly once, the two float16 slices of 2^kw w add up to the weight, scale carries
    2^-kw; non-finite weights and other channel counts are rejected."""
    from casmvsnet_pl_amd import ops
    g = torch.Generator().manual_seed(6)
    slots = {}
    for s in range(8):
        for kb in range(4):
            p, m = 2 * s + (kb >> 1), kb & 1
            if p < 10:
                slots[(s, kb)] = (2 * (p & 1) + m, p >> 1)
            elif p < 15 and m == 0:
                slots[(s, kb)] = (4, p - 10)
    assert sorted(slots.values()) == [(ky, kx) for ky in range(5) for kx in range(5)]
    for cin, cout in ((8, 16), (16, 32)):
        w = torch.randn(cout, cin, 5, 5, generator=g) * 0.2
        scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g)
        packed = ops.conv2d_k5s2_splitf16_pack(w, scale, shift).numpy()
        nimg = (cin // 8) * 8 * (cout // 16)
        assert packed.size == nimg * 2 * 64 * 16 + 2 * cout * 4
        img = packed[:nimg * 2048].view(np.float16).astype(np.float64).reshape(cin // 8, 8, cout // 16, 2, 4, 16, 8)   # [chunk][step][rb][slice][kb][co & 15][e]
        tail = packed[nimg * 2048:].view(np.float32)
        kw = int(round(np.log2(float(scale[0]) / tail[0])))
        assert np.array_equal(tail[:cout], np.ldexp(scale.numpy(), -kw)) and np.array_equal(tail[cout:], shift.numpy())
        assert 2.0 ** 13 <= float(w.abs().max()) * 2.0 ** kw < 2.0 ** 14
        ws = np.ldexp(w.double().numpy(), kw).reshape(cout // 16, 16, cin // 8, 8, 5, 5)    # [rb][i][chunk][e][ky][kx]
        for s in range(8):
            for kb in range(4):
                got = img[:, s, :, 0, kb] + img[:, s, :, 1, kb]                             # [chunk][rb][i][e]
                if (s, kb) in slots:
                    ky, kx = slots[(s, kb)]
                    assert np.abs(got - ws[:, :, :, :, ky, kx].transpose(2, 0, 1, 3)).max() <= 2.0 ** -8
                else:
                    assert not got.any() and not img[:, s, :, :, kb].any()
    with pytest.raises(ValueError):
        ops.conv2d_k5s2_splitf16_pack(torch.randn(16, 16, 5, 5))
    with pytest.raises(ValueError):
        ops.conv2d_k5s2_splitf16_pack(torch.randn(16, 8, 3, 3))
    bad = torch.randn(16, 8, 5, 5)
    bad[3, 2, 1, 1] = float("nan")
    with pytest.raises(RuntimeError, match="finite"):
        ops.conv2d_k5s2_splitf16_pack(bad)


def test_pack_segment_layout_matches_the_header():
    """training.PackPlan writes casmvs_pack_segment records from numpy: four pointers and four int32, 48 bytes, in the order of include/casmvs.h (the C side
    pins its own mirror against the header with a static_assert)."""
    import ctypes
    import re
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "casmvs.h")).read()
    body = re.search(r"typedef struct casmvs_pack_segment \{(.*?)\} casmvs_pack_segment;", hdr, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = [n.strip(" *") for decl in body.split(";") if decl.strip() for n in decl.split(",")]
    names = [n.split()[-1].strip("*") for n in names]
    assert names == ["weight", "bias", "index", "out", "n_weight", "n_bias", "n_out", "first_block"]

    class Seg(ctypes.Structure):
        _fields_ = [("weight", ctypes.c_void_p), ("bias", ctypes.c_void_p), ("index", ctypes.c_void_p), ("out", ctypes.c_void_p),
                    ("n_weight", ctypes.c_int), ("n_bias", ctypes.c_int), ("n_out", ctypes.c_int), ("first_block", ctypes.c_int)]
    dt = np.dtype([("w", "<u8"), ("b", "<u8"), ("idx", "<u8"), ("out", "<u8"), ("n_w", "<i4"), ("n_b", "<i4"), ("n_out", "<i4"), ("first", "<i4")])
    assert ctypes.sizeof(Seg) == dt.itemsize == 48
    assert [dt.fields[k][1] for k in dt.names] == [getattr(Seg, f[0]).offset for f in Seg._fields_]
    import inspect
    from casmvsnet_pl_amd import training
    assert '("w", "<u8"), ("b", "<u8"), ("idx", "<u8"), ("out", "<u8"), ("n_w", "<i4"), ("n_b", "<i4")' in inspect.getsource(training.PackPlan.begin_step)
