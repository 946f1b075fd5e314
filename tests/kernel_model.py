"""Executable model (torch CPU) of the index logic of csrc/conv3d_mfma.hip: it consumes the PACKED
parameter image produced by the C packer (casmvs_conv3d_pack_f32, pure host code) and walks the
same (slice, stage, tap, c, q) -> (A image j, ABID) -> output-channel mapping and the same
transposed-conv parity decomposition as the kernels, assuming the documented MFMA semantics
    v_mfma_f32_4x4x1_16b_f32, CBSZ=4:  D[r][lane] += A[4*ABID + r] * B[lane].
Used by the CPU tests to validate packing + indexing without a GPU (the GPU tests then validate
the MFMA semantics themselves via casmvs_selftest_mfma and full parity).
"""
import torch
import torch.nn.functional as F

S1, S2, T2 = 0, 1, 2


def layer_cfg(kind, cin, cout):
    if kind == S1:
        coutb = 4 if cout == 1 else 8 if cout == 8 else 16
        ck = 8
    elif kind == S2:
        coutb, ck = 16, 4
    else:
        coutb, ck = (8 if cout == 8 else 16), 8
    slices = (cout + coutb - 1) // coutb
    nv = (ck * (coutb // 4) + 15) // 16
    nstages = (cin + ck - 1) // ck
    return coutb, ck, slices, nv, nstages


def emulate(kind, packed, x, cout, skip=None, slope=0.01):
    B, cin, D, H, W = x.shape
    coutb, ck, slices, nv, nstages = layer_cfg(kind, cin, cout)
    Q = coutb // 4
    T = nstages * 27
    img = packed[: slices * T * nv * 64].reshape(slices, T, nv, 64)
    scale = packed[slices * T * nv * 64: slices * T * nv * 64 + slices * coutb]
    shift = packed[slices * T * nv * 64 + slices * coutb: slices * T * nv * 64 + 2 * slices * coutb]
    assert float(packed[slices * T * nv * 64 + 2 * slices * coutb:].abs().sum()) == 0.0 and packed.numel() == slices * T * nv * 64 + 2 * slices * coutb + 64
    if kind == T2:
        Do, Ho, Wo = 2 * D, 2 * H, 2 * W
    elif kind == S2:
        Do, Ho, Wo = D // 2, H // 2, W // 2
    else:
        Do, Ho, Wo = D, H, W
    acc = torch.zeros(B, slices * coutb, Do, Ho, Wo, dtype=torch.float64)
    xd = x.double()
    if kind != T2:
        st = 1 if kind == S1 else 2
        xp = F.pad(xd, (1, 1, 1, 1, 1, 1))
        for sl in range(slices):
            for s in range(nstages):
                for tap in range(27):
                    kz, ky, kx = tap // 9, (tap // 3) % 3, tap % 3
                    for c in range(ck):
                        ci = s * ck + c
                        if ci >= cin:
                            continue  # the kernel multiplies staged zeros here
                        bval = xp[:, ci, kz:kz + st * (Do - 1) + 1:st, ky:ky + st * (Ho - 1) + 1:st, kx:kx + st * (Wo - 1) + 1:st]
                        for q in range(Q):
                            n = c * Q + q
                            a = img[sl, s * 27 + tap, n // 16].double()
                            for r in range(4):
                                acc[:, sl * coutb + 4 * q + r] += a[4 * (n % 16) + r] * bval
    else:
        xp = F.pad(xd, (0, 1, 0, 1, 0, 1))  # cell m + 1 beyond the edge reads zero
        for sl in range(slices):
            for pz in (0, 1):
                for py in (0, 1):
                    for s in range(nstages):
                        for zt in range(2 if pz else 1):
                            kz, dz = ((2, 0) if zt == 0 else (0, 1)) if pz else (1, 0)
                            for yt in range(2 if py else 1):
                                ky, dy = ((2, 0) if yt == 0 else (0, 1)) if py else (1, 0)
                                tap0 = s * 27 + (kz * 3 + ky) * 3
                                for c in range(ck):
                                    ci = s * ck + c
                                    if ci >= cin:
                                        continue
                                    b0 = xp[:, ci, dz:dz + D, dy:dy + H, 0:W]
                                    b1 = xp[:, ci, dz:dz + D, dy:dy + H, 1:W + 1]
                                    for q in range(Q):
                                        n = c * Q + q
                                        a0, a1, a2 = (img[sl, tap0 + k, n // 16].double() for k in range(3))
                                        for r in range(4):
                                            co = sl * coutb + 4 * q + r
                                            w0, w1, w2 = (a[4 * (n % 16) + r] for a in (a0, a1, a2))
                                            acc[:, co, pz::2, py::2, 0::2] += w1 * b0
                                            acc[:, co, pz::2, py::2, 1::2] += w2 * b0 + w0 * b1
    y = acc * scale.double().reshape(1, -1, 1, 1, 1) + shift.double().reshape(1, -1, 1, 1, 1)
    y = torch.where(y > 0, y, y * slope)[:, :cout]
    if skip is not None:
        y = y + skip.double()
    return y.float()
