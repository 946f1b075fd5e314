"""Executable model (torch CPU) of the index logic of csrc/conv3d_mfma.hip: it consumes the PACKED
parameter image produced by the C packer (casmvs_conv3d_pack_f32, pure host code) and walks the
same (slice, chunk, image) -> (row, k) -> (output feature, contraction index) mapping and the same
polyphase / transposed-conv parity decompositions as the kernels, assuming the documented MFMA
semantics
    v_mfma_f32_16x16x4_f32:  lane l supplies A[i = l & 15][k = l >> 4], D[row][col] += sum_k A[row][k] B[k][col]
    v_mfma_f32_4x4x1_16b_f32, CBSZ = 4:  D[r][lane] += A[4*ABID + r] * B[lane].
Used by the CPU tests to validate packing + indexing without a GPU (the GPU tests then validate
the MFMA semantics themselves via casmvs_selftest_mfma and full parity).
"""
import torch
import torch.nn.functional as F

S1, S2, T2 = 0, 1, 2
P1, CI, PX, TCI, TPX = 0, 1, 2, 3, 4


def _ru(x, m):
    return (x + m - 1) // m * m


def layer_cfg(kind, cin, cout):
    """-> fmt, coutb, slices, units, unit_floats (mirrors layer_cfg in conv3d_mfma.hip)."""
    q = _ru((cin + 3) // 4, 4)
    if kind == S1:
        fmt, coutb, units, uf = (P1, 4, _ru(cin, 8), 32) if cout == 1 else (PX, 8, _ru(cin, 8), 9 * 64) if cout == 8 else (CI, 16, q, 27 * 64)
    elif kind == S2:
        fmt, coutb, units, uf = CI, 16, q, 27 * 64
    else:
        fmt, coutb, units, uf = (TPX, 8, q, 18 * 64) if cout == 8 else (TCI, 16, q, 27 * 64)
    return fmt, coutb, (cout + coutb - 1) // coutb, units, uf


def emulate(kind, packed, x, cout, skip=None, slope=0.01):
    B, cin, D, H, W = x.shape
    fmt, coutb, slices, units, uf = layer_cfg(kind, cin, cout)
    nimg = uf // 64
    body = slices * units * uf
    assert packed.numel() == body + 2 * slices * coutb + 64
    if fmt != P1:
        img = packed[:body].reshape(slices, units, nimg, 64).double()   # [slice][unit][image][lane]
    scale = packed[body: body + slices * coutb].double()
    shift = packed[body + slices * coutb: body + 2 * slices * coutb].double()
    assert float(packed[body + 2 * slices * coutb:].abs().sum()) == 0.0
    if kind == T2:
        Do, Ho, Wo = 2 * D, 2 * H, 2 * W
    elif kind == S2:
        Do, Ho, Wo = D // 2, H // 2, W // 2
    else:
        Do, Ho, Wo = D, H, W
    acc = torch.zeros(B, slices * coutb, Do, Ho, Wo, dtype=torch.float64)
    xd = x.double()

    def chan(xp, ci):  # staged tile: channels >= cin are zero-filled
        return xp[:, ci] if ci < cin else torch.zeros_like(xp[:, 0])

    if fmt == P1:  # VALU kernel: row ci of the image = the channel's 27 taps (+ 5 zeros)
        rows = packed[:body].reshape(units, 32).double()
        assert float(rows[:, 27:].abs().sum()) == 0.0
        xp = F.pad(xd, (1, 1, 1, 1, 1, 1))
        for ci in range(units):
            for tap in range(27):
                kz, ky, kx = tap // 9, (tap // 3) % 3, tap % 3
                acc[:, 0] += rows[ci, tap] * chan(xp, ci)[:, kz:kz + D, ky:ky + H, kx:kx + W]
    elif fmt == CI:
        st = 1 if kind == S1 else 2
        xp = F.pad(xd, (1, 1, 1, 1, 1, 1))
        for sl in range(slices):
            for u in range(units):
                for tap in range(27):
                    kz, ky, kx = tap // 9, (tap // 3) % 3, tap % 3
                    A = img[sl, u, tap].reshape(4, 16)  # [k][i]
                    for k in range(4):
                        bval = chan(xp, u * 4 + k)[:, kz:kz + st * (Do - 1) + 1:st, ky:ky + st * (Ho - 1) + 1:st,
                                                   kx:kx + st * (Wo - 1) + 1:st]
                        for i in range(16):
                            acc[:, sl * 16 + i] += A[k, i] * bval
    elif fmt == PX:
        assert W % 2 == 0, "model handles even W only"
        xp = F.pad(xd, (1, 3, 1, 1, 1, 1))
        for u in range(units):
            for r9 in range(9):
                kz, ky = r9 // 3, r9 % 3
                A = img[0, u, r9].reshape(4, 16)  # [x-offset][i = 2*co + sx]
                plane = chan(xp, u)[:, kz:kz + D, ky:ky + H]
                for xo in range(4):
                    bval = plane[..., xo:xo + W:2]  # in[x0 + 2j + xo - 1], j = 0..W/2-1
                    for i in range(16):
                        acc[:, i >> 1, :, :, (i & 1)::2] += A[xo, i] * bval
    elif fmt in (TCI, TPX):
        xp = F.pad(xd, (0, 1, 0, 1, 0, 1))  # cell m + 1 beyond the edge reads zero
        for sl in range(slices):
            for pz in (0, 1):
                for py in (0, 1):
                    for u in range(units):
                        for zt in range(2 if pz else 1):
                            kz, dz = ((2, 0) if zt == 0 else (0, 1)) if pz else (1, 0)
                            for yt in range(2 if py else 1):
                                ky, dy = ((2, 0) if yt == 0 else (0, 1)) if py else (1, 0)
                                r9 = kz * 3 + ky
                                for k in range(4):
                                    pl = chan(xp, u * 4 + k)
                                    b0 = pl[:, dz:dz + D, dy:dy + H, 0:W]
                                    b1 = pl[:, dz:dz + D, dy:dy + H, 1:W + 1]
                                    if fmt == TCI:
                                        a0, a1, a2 = (img[sl, u, r9 * 3 + kx].reshape(4, 16)[k] for kx in range(3))
                                        for i in range(16):
                                            co = sl * 16 + i
                                            acc[:, co, pz::2, py::2, 0::2] += a1[i] * b0
                                            acc[:, co, pz::2, py::2, 1::2] += a2[i] * b0 + a0[i] * b1
                                    else:
                                        ad0 = img[sl, u, r9 * 2 + 0].reshape(4, 16)[k]
                                        ad1 = img[sl, u, r9 * 2 + 1].reshape(4, 16)[k]
                                        for i in range(16):
                                            acc[:, i >> 1, pz::2, py::2, (i & 1)::2] += ad0[i] * b0 + ad1[i] * b1
    y = acc * scale.reshape(1, -1, 1, 1, 1) + shift.reshape(1, -1, 1, 1, 1)
    y = torch.where(y > 0, y, y * slope)[:, :cout]
    if skip is not None:
        y = y + skip.double()
    return y.float()


# ---- FeatureNet 2D layers: the same CI / PX formats with a 1-deep kernel (kz = 1) ----------------
K3, K5S2, K1, K1_UP = 3, 4, 5, 6


def layer_cfg2d(kind, cin, cout):
    """-> fmt, coutb, slices, units, unit_floats, ks, stride (mirrors layer_cfg in conv3d_mfma.hip)."""
    q = _ru((cin + 3) // 4, 4)
    if kind == K3:
        fmt, coutb, units, uf, ks, st = (PX, 8, _ru(cin, 8), 3 * 64, 3, 1) if cout == 8 else (CI, 16, q, 9 * 64, 3, 1)
    elif kind == K5S2:
        fmt, coutb, units, uf, ks, st = CI, 16, q, 25 * 64, 5, 2
    else:
        fmt, coutb, units, uf, ks, st = CI, 16, q, 64, 1, 1
    return fmt, coutb, (cout + coutb - 1) // coutb, units, uf, ks, st


def emulate2d(kind, packed, x, cout, up=None, slope=0.01):
    N, cin, H, W = x.shape
    fmt, coutb, slices, units, uf, ks, st = layer_cfg2d(kind, cin, cout)
    nimg = uf // 64
    body = slices * units * uf
    assert packed.numel() == body + 2 * slices * coutb + 64
    img = packed[:body].reshape(slices, units, nimg, 64).double()
    scale = packed[body: body + slices * coutb].double()
    shift = packed[body + slices * coutb: body + 2 * slices * coutb].double()
    Ho, Wo = H // st, W // st
    acc = torch.zeros(N, slices * coutb, Ho, Wo, dtype=torch.float64)
    xd = x.double()
    pad = ks // 2

    def chan(xp, ci):
        return xp[:, ci] if ci < cin else torch.zeros_like(xp[:, 0])

    if fmt == CI:
        xp = F.pad(xd, (pad, pad, pad, pad))
        for sl in range(slices):
            for u in range(units):
                for tap in range(ks * ks):
                    ky, kx = tap // ks, tap % ks
                    A = img[sl, u, tap].reshape(4, 16)
                    for k in range(4):
                        bval = chan(xp, u * 4 + k)[:, ky:ky + st * (Ho - 1) + 1:st, kx:kx + st * (Wo - 1) + 1:st]
                        for i in range(16):
                            acc[:, sl * 16 + i] += A[k, i] * bval
    else:
        assert W % 2 == 0
        xp = F.pad(xd, (1, 3, 1, 1))
        for u in range(units):
            for ky in range(3):
                A = img[0, u, ky].reshape(4, 16)
                plane = chan(xp, u)[:, ky:ky + H]
                for xo in range(4):
                    bval = plane[..., xo:xo + W:2]
                    for i in range(16):
                        acc[:, i >> 1, :, (i & 1)::2] += A[xo, i] * bval
    y = acc * scale.reshape(1, -1, 1, 1) + shift.reshape(1, -1, 1, 1)
    y = torch.where(y > 0, y, y * slope)[:, :cout]
    if up is not None:
        y = y + F.interpolate(up.double(), scale_factor=2, mode="bilinear", align_corners=True)
    return y.float()
