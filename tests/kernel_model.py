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
        fmt, coutb, units, uf = (P1, 4, _ru(cin, 8) // 2, 64) if cout == 1 else (PX, 8, _ru(cin, 8), 9 * 64) if cout == 8 else (CI, 16, q, 27 * 64)
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

    if fmt == P1:  # VALU kernel: row p of the image = the 27 taps (+ 5 zeros) of the channel PAIR p, [tap][channel & 1]
        rows = packed[:body].reshape(units, 32, 2).double()
        assert float(rows[:, 27:].abs().sum()) == 0.0
        xp = F.pad(xd, (1, 1, 1, 1, 1, 1))
        for ci in range(2 * units):
            for tap in range(27):
                kz, ky, kx = tap // 9, (tap // 3) % 3, tap % 3
                acc[:, 0] += rows[ci // 2, tap, ci % 2] * chan(xp, ci)[:, kz:kz + D, ky:ky + H, kx:kx + W]
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


# ---- stride-2 layers on the x-de-interleaved LDS rows (Stager<5> / DbStager<DEINT>) ---------------
def emulate_s2_deint(x, w, TZ=2, TY=4, TX=16):
    """Index walk of the stride-2 CI kernels with 16-byte staging: the row of a tile starts at input column
    2 x0 - 4 and is stored [even columns | odd columns]; tap kx of output column j reads column 2 j + kx + 3 of
    the row = index j + 2 of the even half (kx = 1) or j + 1 / j + 2 of the odd half (kx = 0 / 2).
    x (cin, D, H, W), w (cout, cin, 3, 3, 3) -> (cout, D/2, H/2, W/2)."""
    import numpy as np
    cin, D, H, W = x.shape
    cout = w.shape[0]
    IZ, IY = 2 * (TZ - 1) + 3, 2 * (TY - 1) + 3
    IX = (4 + 2 * (TX - 1) + 2 + 3) // 4 * 4
    Do, Ho, Wo = D // 2, H // 2, W // 2
    out = np.zeros((cout, Do, Ho, Wo), dtype=np.float64)
    for tz0 in range(0, Do, TZ):
        for ty0 in range(0, Ho, TY):
            for tx0 in range(0, Wo, TX):
                tile = np.zeros((cin, IZ, IY, IX))
                iz0, iy0, ix0 = tz0 * 2 - 1, ty0 * 2 - 1, tx0 * 2 - 4
                for iz in range(IZ):
                    for iy in range(IY):
                        for xv in range(IX // 4):  # one staged 16-byte group
                            gz, gy, gx = iz0 + iz, iy0 + iy, ix0 + 4 * xv
                            v = np.zeros((cin, 4))
                            if 0 <= gz < D and 0 <= gy < H and gx >= 0 and gx + 3 < W:
                                v = x[:, gz, gy, gx:gx + 4]
                            tile[:, iz, iy, 2 * xv:2 * xv + 2] = v[:, [0, 2]]
                            tile[:, iz, iy, IX // 2 + 2 * xv:IX // 2 + 2 * xv + 2] = v[:, [1, 3]]
                for cz in range(TZ):
                    for cy in range(TY):
                        for j in range(TX):
                            oz, oy, ox = tz0 + cz, ty0 + cy, tx0 + j
                            if oz >= Do or oy >= Ho or ox >= Wo:
                                continue
                            acc = np.zeros(cout)
                            for it in range(27):
                                kz, ky, kx = it // 9, (it // 3) % 3, it % 3
                                xo = 2 if kx == 1 else IX // 2 + (1 if kx == 0 else 2)
                                acc += w[:, :, kz, ky, kx] @ tile[:, cz * 2 + kz, cy * 2 + ky, j + xo]
                            out[:, oz, oy, ox] = acc
    return out


# ---- FPN top-down step as fpn_lateral_kernel computes it (thread = 4 consecutive x) ---------------
def emulate_fpn_lateral(x, w, b, up):
    """x (cin, H, W), w (cout, cin), b (cout), up (cout, H/2, W/2) -> (cout, H, W): 4-column source window
    [xb, xb + 4) per thread and a 4 x 4 tent matrix for the horizontal interpolation (fp32 like the kernel)."""
    import numpy as np
    f32 = np.float32
    cin, H, W = x.shape
    cout, hc, wc = up.shape
    assert W % 4 == 0 and wc >= 4
    sy = f32(hc - 1) / f32(H - 1) if H > 1 else f32(0)
    sx = f32(wc - 1) / f32(W - 1) if W > 1 else f32(0)
    out = np.zeros((cout, H, W), dtype=np.float32)
    lat = np.einsum("oc,chw->ohw", w.astype(np.float64), x.astype(np.float64)) + b[:, None, None]
    for y in range(H):
        fy = f32(sy * f32(y))
        y0 = int(fy)
        y1 = y0 + (1 if y0 < hc - 1 else 0)
        ly1 = f32(fy - f32(y0))
        ly0 = f32(1) - ly1
        for ox in range(0, W, 4):
            T = np.zeros((4, 4), dtype=np.float32)
            xb = None
            for k in range(4):
                fx = f32(sx * f32(ox + k))
                x0 = int(fx)
                x1 = x0 + (1 if x0 < wc - 1 else 0)
                lx1 = f32(fx - f32(x0))
                lx0 = f32(1) - lx1
                if k == 0:
                    xb = x0 if x0 < wc - 4 else wc - 4
                assert 0 <= x0 - xb < 4 and 0 <= x1 - xb < 4, "a pixel's columns must lie inside the 4-column window"
                T[k, x0 - xb] += lx0
                T[k, x1 - xb] += lx1
            r0, r1 = up[:, y0, xb:xb + 4], up[:, y1, xb:xb + 4]      # (cout, 4)
            h0, h1 = r0 @ T.T, r1 @ T.T                               # (cout, 4 pixels)
            out[:, y, ox:ox + 4] = lat[:, y, ox:ox + 4] + (ly0 * h0 + ly1 * h1)
    return out


# ---- weight-gradient kernel: 16-byte staging of a tile (csrc/train.hip, conv_wgrad_kernel<S, KZ, KS, VEC = true>) ----------
def wgrad_cfg(S, KZ, KS):
    """WgradCfg<S, KZ, KS> of train.hip."""
    TZ = 4 if (KZ == 3 and S == 1) else 1
    TY, TX = 4, 16
    IZ, IY, IX = (TZ - 1) * S + KZ, (TY - 1) * S + KS, (TX - 1) * S + KS
    return dict(TZ=TZ, TY=TY, TX=TX, ROWS=TZ * TY, P=KS // 2, PZ=KZ // 2, IZ=IZ, IY=IY, IX=IX)


def wgrad_vector_staging_units(S, KZ, KS, threads=256):
    """The (kind, LDS cell(s), element offsets relative to the tile origin) every thread writes in the vector staging, in
    the kernel's own index arithmetic: 'small' units (c, row, 4 x), 'big' vector units (c, iz, iy, ix .. ix + 3), halo
    scalars (c, iz, iy, ix)."""
    g = wgrad_cfg(S, KZ, KS)
    ROWS, P, IZ, IY, IX = g["ROWS"], g["P"], g["IZ"], g["IY"], g["IX"]
    small, big = [], []
    SU = 16 * ROWS * 4
    for u in range(_ru(SU, threads)):
        if u < SU:
            c, r, k = u // (ROWS * 4), (u >> 2) % ROWS, u & 3
            small += [(c, r, 4 * k + j) for j in range(4)]
    RB, VPR, NH = 16 * IZ * IY, 4 * S, KS - S
    for u in range(_ru(RB * VPR, threads)):
        if u < RB * VPR:
            row, k = divmod(u, VPR)
            c, rem = divmod(row, IZ * IY)
            iz, iy = divmod(rem, IY)
            big += [(c, iz, iy, P + 4 * k + j) for j in range(4)]
    for u in range(_ru(RB * NH, threads) if NH > 0 else 0):
        if u < RB * NH:
            row, h = divmod(u, NH)
            c, rem = divmod(row, IZ * IY)
            iz, iy = divmod(rem, IY)
            big.append((c, iz, iy, h if h < P else 16 * S + h))
    return g, small, big


# ---- csrc/prob_regress.hip: the `prob` head walking the depth axis ---------------------------------------------------
PZ_TX, PZ_TY, PZ_THREADS = 64, 8, 256
PZ_IY, PZ_NPP = PZ_TY + 2, PZ_TX // 2 + 1
PZ_RS = 4 * PZ_NPP
PZ_SP = PZ_IY * PZ_RS
PZ_SLOT = 4 * PZ_SP
PZ_ITEMS = 4 * PZ_IY * PZ_NPP

# lane groups of one ds_read_b128 wave-instruction (MI355X_MICROARCH.md, LDS table): 4 x 16 lanes, one LDS cycle each
_B128_GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
_B128_GROUPS += [[l + 32 for l in g] for g in _B128_GROUPS]


def _b128_cycles(addr_of_lane, groups):
    total = 0
    for g in groups:
        banks = {}
        for l in g:
            a = addr_of_lane(l)
            for w in range(4):
                banks.setdefault((a + w) % 64, set()).add(a + w)
        total += max(len(v) for v in banks.values())
    return total


def prob_zwalk_bank_cycles(row_stride=PZ_RS, first_float=0):
    """LDS-array cycles of one ds_read_b128 of the z-walk kernel's tap reads (lane = (xi = l & 31, yi = l >> 5) reads the
    4 floats at yi * RS + 4 xi + first_float): 4 = conflict-free (64 banks of 4 bytes; same address broadcasts)."""
    return _b128_cycles(lambda l: (l >> 5) * row_stride + 4 * (l & 31) + first_float, _B128_GROUPS)


def prob_zwalk_write_cycles():
    """LDS-array cycles of one ds_write_b128 of the staging (8 x 8 contiguous lanes, 32 banks): lane l writes floats
    [4 l, 4 l + 4) - 8 = conflict-free.  The first version's ds_write_b64 (lane l writes 2 floats at 8 l) took 4 x its
    minimum."""
    def cyc(addr, width, groups):
        total = 0
        for g in groups:
            banks = {}
            for l in g:
                for w in range(width):
                    banks.setdefault((addr(l) + w) % 32, set()).add(addr(l) + w)
            total += max(len(v) for v in banks.values())
        return total
    g8 = [list(range(8 * i, 8 * i + 8)) for i in range(8)]
    g16 = [list(range(16 * i, 16 * i + 16)) for i in range(4)]
    return cyc(lambda l: 4 * l, 4, g8), cyc(lambda l: 8 * l, 2, g16)


def prob_zwalk_staging_plan(tx0, W):
    """-> list over staging items e < ITEMS of (pair, staged row, x of the pair's first position, x loaded, shift) with
    shift = 0 (as loaded), +1 (left edge: keep (0, loaded[0])), -1 (right edge: keep (loaded[1], 0))."""
    plan = []
    for e in range(PZ_ITEMS):
        p, r = divmod(e, PZ_IY * PZ_NPP)
        iy, m = divmod(r, PZ_NPP)
        x = tx0 - 1 + 2 * m
        edge_l, edge_r = x < 0, x + 1 == W
        plan.append((p, iy, x, 0 if edge_l else (x - 1 if edge_r else x), 1 if edge_l else (-1 if edge_r else 0)))
    return plan


def emulate_prob_zwalk(packed, x, zc, slope=1.0):
    """The kernel's data flow in float64: per (tile, chunk) the input planes z_lo - 1 .. z_hi are staged one at a time into a
    pair-interleaved halo tile (item e = 16 bytes at float 4 e), every thread (xi, yi) reads rows yi + ky at floats
    [4 xi, 4 xi + 8) and accumulates the plane into the three rotating accumulators (output planes z_in + 1, z_in, z_in - 1
    for kz = 0, 1, 2).  packed: the P1 image of casmvs_conv3d_pack_f32 (cin = 8, cout = 1).
    x (B, 8, D, H, W), W % 4 == 0 -> (B, D, H, W)."""
    import numpy as np
    B, cin, D, H, W = x.shape
    assert cin == 8 and W % 4 == 0
    xn = x.double().numpy()
    pk = packed.double().numpy()
    wq = pk[:256].reshape(4, 32, 2)           # [pair][tap (27 + 5 zeros)][channel of the pair]
    sc0, sh0 = pk[256], pk[260]
    out = np.full((B, D, H, W), np.nan)
    written = np.zeros((B, D, H, W), dtype=np.int32)
    tiles_x, tiles_y = -(-W // PZ_TX), -(-H // PZ_TY)
    nchunk = -(-D // zc)
    tid = np.arange(PZ_THREADS)
    xi, yi = tid & 31, tid >> 5
    for b in range(B):
        for ty in range(tiles_y):
            for tx in range(tiles_x):
                tx0, ty0 = tx * PZ_TX, ty * PZ_TY
                plan = prob_zwalk_staging_plan(tx0, W)
                for ch in range(nchunk):
                    z_lo, z_hi = ch * zc, min(ch * zc + zc, D)
                    A = np.zeros((3, 2, 2, PZ_THREADS))   # [acc][pixel][channel parity][thread]
                    nplanes = z_hi - z_lo + 2
                    for it in range(nplanes):
                        zin = z_lo - 1 + it
                        if 0 <= zin < D:
                            slot = np.full(PZ_SLOT, np.nan)
                            for e, (p, iy, xp, xl, shift) in enumerate(plan):
                                gy = ty0 - 1 + iy
                                if 0 <= gy < H and xp < W:
                                    assert 0 <= xl and xl + 1 < W, "the loaded pair lies inside the row"
                                    ld = xn[b, 2 * p:2 * p + 2, zin, gy, xl:xl + 2]       # [channel][2 positions]
                                    pos = ld if shift == 0 else (np.stack([np.zeros(2), ld[:, 0]], 1) if shift == 1 else np.stack([ld[:, 1], np.zeros(2)], 1))
                                else:
                                    pos = np.zeros((2, 2))                                # out-of-range offset: the hardware returns 0
                                assert np.isnan(slot[4 * e:4 * e + 4]).all()
                                slot[4 * e:4 * e + 4] = [pos[0, 0], pos[1, 0], pos[0, 1], pos[1, 1]]
                            assert not np.isnan(slot).any(), "every LDS cell of the slot is written exactly once"
                            kzs = [0] if it == 0 else [2] if it == nplanes - 1 else [0, 1, 2]
                            for p in range(4):
                                for ky in range(3):
                                    base = p * PZ_SP + (yi + ky) * PZ_RS + 4 * xi
                                    P = slot[base[None, :] + np.arange(8)[:, None]].reshape(4, 2, PZ_THREADS)  # [position][parity][thread]
                                    for kx in range(3):
                                        for kz in kzs:
                                            w = wq[p, kz * 9 + ky * 3 + kx]     # (even, odd) channel weights
                                            A[2 - kz, 0] += P[kx] * w[:, None]
                                            A[2 - kz, 1] += P[kx + 1] * w[:, None]
                        if it >= 2:
                            z = zin - 1
                            assert z_lo <= z < z_hi
                            for j in range(2):
                                v = (A[0, j, 0] + A[0, j, 1]) * sc0 + sh0
                                v = np.where(v > 0, v, v * slope)
                                oy, ox = ty0 + yi, tx0 + 2 * xi + j
                                m = (oy < H) & (ox < W)
                                out[b, z, oy[m], ox[m]] = v[m]
                                written[b, z, oy[m], ox[m]] += 1
                        A[0], A[1] = A[1].copy(), A[2].copy()
                        A[2] = 0.0
    assert (written == 1).all(), "every output voxel is produced exactly once"
    return torch.from_numpy(out)


# ---- csrc/fpn_fused.hip: lat0 + upsample-add + smooth0 as one 40-channel PX-form 3x3 layer -----------------------------
def emulate_fpn_tail0(packed40, bias9, c0, f1):
    """The kernel's data flow (float32 staging like the kernel, float64 accumulation): per 8 x 64 output tile and chunk of 4
    input channels the halo tile [4][10][72] is staged - channels 0..7 straight from c0, channels 8..39 interpolated from the
    half-resolution f1 through the 4-column window / 4 x 4 tent matrix and ATen's vertical weights - and multiplied in the PX
    form (row i = (co = i >> 1, x phase i & 1), k = input x offset) with the lane images of the C packer.
    c0 (8, H, W), f1 (32, H/2, W/2) numpy float32 -> (8, H, W)."""
    import numpy as np
    f32 = np.float32
    _, H, W = c0.shape
    hc, wc = H // 2, W // 2
    img = np.asarray(packed40[:40 * 192], dtype=np.float64).reshape(40, 3, 4, 16)     # [ci][ky][k][i]
    sy = f32(hc - 1) / f32(H - 1) if H > 1 else f32(0)
    sx = f32(wc - 1) / f32(W - 1) if W > 1 else f32(0)
    out = np.full((8, H, W), np.nan)
    for ty0 in range(0, H, 8):
        for tx0 in range(0, W, 64):
            acc = np.zeros((8, 8, 64))                                                  # [co][cy][x - tx0]
            for s in range(10):
                tile = np.zeros((4, 10, 72), dtype=np.float64)
                for c in range(4):
                    for iy in range(10):
                        for g in range(18):
                            gy, gx = ty0 - 1 + iy, tx0 - 4 + 4 * g
                            if not (0 <= gy < H and 0 <= gx < W):
                                continue                                                 # out-of-range offset: zeros
                            if s < 2:
                                tile[c, iy, 4 * g:4 * g + 4] = c0[4 * s + c, gy, gx:gx + 4]
                                continue
                            fy = f32(sy * f32(gy))
                            y0 = int(fy)
                            y1 = y0 + (1 if y0 < hc - 1 else 0)
                            ly1 = f32(fy - f32(y0))
                            ly0 = f32(1) - ly1
                            T = np.zeros((4, 4), dtype=np.float32)
                            xb = None
                            for j in range(4):
                                fx = f32(sx * f32(gx + j))
                                x0 = int(fx)
                                x1 = x0 + (1 if x0 < wc - 1 else 0)
                                lx1 = f32(fx - f32(x0))
                                if j == 0:
                                    xb = x0 if x0 < wc - 4 else wc - 4
                                assert 0 <= x0 - xb < 4 and 0 <= x1 - xb < 4, "a pixel's columns must lie inside the 4-column window"
                                T[j, x0 - xb] += f32(1) - lx1
                                T[j, x1 - xb] += lx1
                            ch = 4 * (s - 2) + c
                            top, bot = T @ f1[ch, y0, xb:xb + 4], T @ f1[ch, y1, xb:xb + 4]
                            tile[c, iy, 4 * g:4 * g + 4] = ly0 * top + ly1 * bot
                for c in range(4):
                    for ky in range(3):
                        A = img[4 * s + c, ky]                                           # [k][i]
                        for cy in range(8):
                            for cx in range(2):
                                for k in range(4):
                                    bvals = tile[c, cy + ky, cx * 32 + 3 + k + 2 * np.arange(16)]    # B[k][j]
                                    for i in range(16):
                                        acc[i >> 1, cy, cx * 32 + (i & 1) + 2 * np.arange(16)] += A[k, i] * bvals
            for cy in range(8):
                oy = ty0 + cy
                if oy >= H:
                    continue
                r = 0 if oy == 0 else (2 if oy == H - 1 else 1)
                for x in range(64):
                    ox = tx0 + x
                    if ox >= W:
                        continue
                    cc = 0 if ox == 0 else (2 if ox == W - 1 else 1)
                    out[:, oy, ox] = acc[:, cy, x] + np.asarray(bias9[r, cc], dtype=np.float64)
    assert not np.isnan(out).any()
    return out


def emulate_fpn_tail0_splitf16(packed, bias9, c0, f1, tile=(20, 32)):
    """Data flow of fpn_tail0_sf_kernel in float64: the 40-channel input [c0 | up(f1)] (float32: the upsample by torch's ATen kernel,
    whose rule the kernel restates) is staged per 20 x 32 output tile and chunk of 8 channels as the halo tile (y0-1..y0+20,
    x0-4..x0+35; zero outside), scaled to [2^14, 2^15), split into two float16 slices and multiplied (aa, ab, ba) with the packer's
    lane images [chunk][ky][slice][lane][8 f16]; chunks unscaled and summed; then 2^-kw and the nine border bias classes.
    c0 (8, H, W), f1 (32, H/2, W/2) numpy float32 -> (8, H, W)."""
    import numpy as np
    import torch
    raw = np.asarray(packed, dtype=np.uint8)
    body = 5 * 3 * 2 * 64 * 8 * 2
    img = raw[:body].view(np.float16).reshape(5, 3, 2, 64, 8).astype(np.float64)
    unscale = float(raw[body:body + 4].view(np.float32)[0])
    _, H, W = c0.shape
    up = torch.nn.functional.interpolate(torch.from_numpy(np.ascontiguousarray(f1))[None], scale_factor=2, mode="bilinear", align_corners=True)[0].numpy()
    x40 = np.concatenate([c0.astype(np.float32), up.astype(np.float32)], 0)
    TY, TX = tile
    py, px = ((H + TY - 1) // TY) * TY - H, ((W + TX - 1) // TX) * TX - W
    xp = np.pad(x40, ((0, 0), (1, 1 + py), (4, 4 + px)))
    out = np.zeros((8, H, W))
    for y0 in range(0, H, TY):
        for x0 in range(0, W, TX):
            acc = np.zeros((8, TY, TX))
            for ch in range(5):
                halo = xp[ch * 8:ch * 8 + 8, y0:y0 + TY + 2, x0:x0 + TX + 8]
                e = max(int(np.abs(halo).max().view(np.uint32)) >> 23, 15)
                mult, inv = np.float32(2.0) ** (141 - e), 2.0 ** (e - 141)
                xs = halo * mult
                xa = xs.astype(np.float16)
                xb = (xs - xa.astype(np.float32)).astype(np.float16)
                sl = [xa.astype(np.float64), xb.astype(np.float64)]
                part = np.zeros((8, TY, TX))
                for ky in range(3):
                    for (sa, sb) in ((0, 0), (0, 1), (1, 0)):
                        A = img[ch, ky, sa].reshape(4, 16, 8)                               # [u][i][ci]
                        for u in range(4):
                            for s in range(2):
                                wrow = A[u, s::2, :]                                        # (co, ci); zero where u - s is not a tap
                                if not wrow.any():
                                    continue
                                src = sl[sb][:, ky:ky + TY, u + 3:u + 3 + TX:2]             # outputs x0 + 2 j + s read halo column 2 j + u + 3
                                part[:, :, s::2] += np.einsum("oc,chw->ohw", wrow, src)
                acc += part * inv
            dy, dx = min(TY, H - y0), min(TX, W - x0)
            out[:, y0:y0 + dy, x0:x0 + dx] = acc[:, :dy, :dx] * unscale
    b9 = np.asarray(bias9, dtype=np.float64)
    rows = np.array([0 if y == 0 else (2 if y == H - 1 else 1) for y in range(H)])
    cols = np.array([0 if x == 0 else (2 if x == W - 1 else 1) for x in range(W)])
    return out + b9[rows][:, cols].transpose(2, 0, 1)


# ---- csrc/conv0_splitbf16.hip: conv0 on the bf16 matrix cores, float32 operands as three exact bf16 slices ---------------
def bf16_split3(x):
    """x float32 ndarray -> three float32 arrays (each exactly a bf16 value) with hi + mid + lo == x exactly (truncation:
    the three 8-bit slices of the 24-bit significand, as the kernel's mask-and-subtract)."""
    import numpy as np
    x = np.asarray(x, dtype=np.float32)
    hi = (x.view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)
    r = (x - hi).astype(np.float32)
    mid = (r.view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)
    lo = (r - mid).astype(np.float32)
    return hi, mid, lo


SB_TERMS = [(0, 0), (0, 1), (1, 0), (0, 2), (2, 0), (1, 1), (1, 2), (2, 1), (2, 2)]   # (weight slice, activation slice), by magnitude class


def emulate_conv0_splitbf16(packed, x, cin, terms=6, slope=0.01):
    """Data flow of conv0_sb_kernel in float64: the packed lane images [chunk][kz*3+ky][slice][lane][8 bf16] are decoded, the
    activations split like the kernel does, and every output = sum over (chunk, kz, ky, x offset u, ci) of the first `terms`
    partial products A_slice[i][k] * B_slice[k][j] with rows i = (co, x phase), k = (u, ci), then scale / shift / leaky-relu.
    x (B, cin, D, H, W) float32 numpy, W even -> (B, 8, D, H, W)."""
    import numpy as np
    raw = np.asarray(packed, dtype=np.uint8)
    nch = cin // 8
    body = nch * 9 * 3 * 64 * 8 * 2
    img = (raw[:body].view(np.uint16).astype(np.uint32) << 16).view(np.float32).reshape(nch, 9, 3, 64, 8).astype(np.float64)
    tail = raw[body:body + 64].view(np.float32).astype(np.float64)
    scale, shift = tail[:8], tail[8:16]
    B, _, D, H, W = x.shape
    xs = [np.pad(s.astype(np.float64), ((0, 0), (0, 0), (1, 1), (1, 1), (1, 3))) for s in bf16_split3(x)]   # zero halo (x: -1 .. W + 2)
    assert float(np.abs(xs[0] + xs[1] + xs[2] - np.pad(x.astype(np.float64), ((0, 0), (0, 0), (1, 1), (1, 1), (1, 3)))).max()) == 0.0
    acc = np.zeros((B, 8, D, H, W))
    for ch in range(nch):
        for r9 in range(9):
            kz, ky = divmod(r9, 3)
            for (sa, sb) in SB_TERMS[:terms]:
                A = img[ch, r9, sa].reshape(4, 16, 8)                                   # [u][i][ci]
                for u in range(4):
                    for s in range(2):                                                   # row i = 2 co + s
                        kx = u - s                                                       # A is zero where kx is not a tap
                        wrow = A[u, s::2, :]                                             # (co, ci)
                        if not wrow.any():
                            continue
                        assert 0 <= kx <= 2
                        # outputs x = 2 j + s read input x + kx - 1 = 2 j + u - 1 -> padded index 2 j + u
                        src = xs[sb][:, ch * 8:ch * 8 + 8, kz:kz + D, ky:ky + H, u:u + W:2][..., :(W - s + 1) // 2]
                        acc[:, :, :, :, s::2] += np.einsum("oc,bcdhw->bodhw", wrow, src)
    y = acc * scale[None, :, None, None, None] + shift[None, :, None, None, None]
    return np.where(y > 0, y, y * slope)


SF_TERMS = [(0, 0), (0, 1), (1, 0), (1, 1)]


def emulate_conv0_splitf16(packed, x, cin, terms=3, slope=0.01, tile=(4, 4, 32), halo_x=(4, 4)):
    """Data flow of conv0_sf_kernel in float64: per output tile and chunk of 8 input channels the staged halo tile
    (z0-1..z0+TZ, y0-1..y0+TY, x0-4..x0+TX+3; zero outside the volume) is scaled by the power of two that puts its largest
    magnitude into [2^14, 2^15), split into f16(x') and f16(x' - f16(x')) (round to nearest even), multiplied with the packed
    lane images' two float16 weight slices (first `terms` of aa, ab, ba, bb), unscaled by 2^-kx and summed over the chunks;
    then scale (which carries 2^-kw) / shift / leaky-relu.  x (B, cin, D, H, W) float32 numpy -> (B, 8, D, H, W)."""
    import numpy as np
    raw = np.asarray(packed, dtype=np.uint8)
    nch = cin // 8
    body = nch * 9 * 2 * 64 * 8 * 2
    img = raw[:body].view(np.float16).reshape(nch, 9, 2, 64, 8).astype(np.float64)
    tail = raw[body:body + 64].view(np.float32).astype(np.float64)
    scale, shift = tail[:8], tail[8:16]
    B, _, D, H, W = x.shape
    TZ, TY, TX = tile
    hl, hr = halo_x
    px = ((W + TX - 1) // TX) * TX - W
    xp = np.pad(x.astype(np.float32), ((0, 0), (0, 0), (1, TZ + 1), (1, TY + 1), (hl, hr + px)))
    acc = np.zeros((B, 8, D, H, W))
    for b in range(B):
        for z0 in range(0, D, TZ):
            for y0 in range(0, H, TY):
                for x0 in range(0, W, TX):
                    out = np.zeros((8, TZ, TY, TX))
                    for ch in range(nch):
                        halo = xp[b, ch * 8:ch * 8 + 8, z0:z0 + TZ + 2, y0:y0 + TY + 2, x0:x0 + TX + hl + hr]
                        e = max(int(np.abs(halo).max().view(np.uint32)) >> 23, 15)
                        mult, inv = np.float32(2.0) ** (141 - e), 2.0 ** (e - 141)
                        xs = halo * mult
                        xa = xs.astype(np.float16)
                        xb = (xs - xa.astype(np.float32)).astype(np.float16)
                        sl = [xa.astype(np.float64), xb.astype(np.float64)]
                        part = np.zeros((8, TZ, TY, TX))
                        for r9 in range(9):
                            kz, ky = divmod(r9, 3)
                            for (sa, sb) in SF_TERMS[:terms]:
                                A = img[ch, r9, sa].reshape(4, 16, 8)                       # [u][i][ci]
                                for u in range(4):
                                    for s in range(2):
                                        wrow = A[u, s::2, :]                                # (co, ci); zero where u - s is not a tap
                                        if not wrow.any():
                                            continue
                                        # outputs x = x0 + 2 j + s read input x0 + 2 j + u - 1 = halo column 2 j + u - 1 + hl
                                        src = sl[sb][:, kz:kz + TZ, ky:ky + TY, u - 1 + hl:u - 1 + hl + TX:2]
                                        part[:, :, :, s::2] += np.einsum("oc,cdhw->odhw", wrow, src)
                        out += part * inv
                    dz, dy, dx = min(TZ, D - z0), min(TY, H - y0), min(TX, W - x0)
                    acc[b, :, z0:z0 + dz, y0:y0 + dy, x0:x0 + dx] = out[:, :dz, :dy, :dx]
    y = acc * scale[None, :, None, None, None] + shift[None, :, None, None, None]
    return np.where(y > 0, y, y * slope)


def emulate_conv_ci_splitf16(packed, x, cin, cout, slope=0.01, tile=(4, 4, 16), halo_x=(2, 2)):
    """Data flow of conv_ci_sf_kernel in float64.  The lane images [chunk][step][row block][slice][lane][8 f16] are decoded into the
    two weight slices W_s[co][ci][tap] exactly as the kernel's lanes meet them (lane (i, kb) of step m: co = 16 rb + i,
    ci = 16 chunk + 8 (kb & 1) + e, tap 2 m + (kb >> 1); tap 27 must be zero); per output tile and chunk of 16 input channels the
    staged halo tile (z0-1..z0+TZ, y0-1..y0+TY, x0-2..x0+TX+1; zero outside) is scaled, split and multiplied (aa, ab, ba), unscaled
    and summed over the chunks; then scale / shift / leaky-relu.  x (B, cin, D, H, W) float32 numpy -> (B, cout, D, H, W)."""
    import numpy as np
    raw = np.asarray(packed, dtype=np.uint8)
    nch, nrb = cin // 16, cout // 16
    body = nch * 14 * nrb * 2 * 64 * 8 * 2
    img = raw[:body].view(np.float16).reshape(nch, 14, nrb, 2, 64, 8).astype(np.float64)
    tail = raw[body:body + 8 * cout].view(np.float32).astype(np.float64)
    scale, shift = tail[:cout], tail[cout:]
    Ws = np.zeros((2, cout, cin, 28))
    for ch in range(nch):
        for st in range(14):
            for rb in range(nrb):
                for lane in range(64):
                    i, kb = lane & 15, lane >> 4
                    Ws[:, 16 * rb + i, 16 * ch + 8 * (kb & 1):16 * ch + 8 * (kb & 1) + 8, 2 * st + (kb >> 1)] = img[ch, st, rb, :, lane, :]
    assert not Ws[:, :, :, 27].any()
    Ws = Ws[:, :, :, :27].reshape(2, cout, cin, 3, 3, 3)
    B, _, D, H, W = x.shape
    TZ, TY, TX = tile
    hl, hr = halo_x
    px = ((W + TX - 1) // TX) * TX - W
    xp = np.pad(x.astype(np.float32), ((0, 0), (0, 0), (1, TZ + 1), (1, TY + 1), (hl, hr + px)))
    acc = np.zeros((B, cout, D, H, W))
    for b in range(B):
        for z0 in range(0, D, TZ):
            for y0 in range(0, H, TY):
                for x0 in range(0, W, TX):
                    out = np.zeros((cout, TZ, TY, TX))
                    for ch in range(nch):
                        halo = xp[b, ch * 16:ch * 16 + 16, z0:z0 + TZ + 2, y0:y0 + TY + 2, x0:x0 + TX + hl + hr]
                        e = max(int(np.abs(halo).max().view(np.uint32)) >> 23, 15)
                        mult, inv = np.float32(2.0) ** (141 - e), 2.0 ** (e - 141)
                        xs = halo * mult
                        xa = xs.astype(np.float16)
                        xb = (xs - xa.astype(np.float32)).astype(np.float16)
                        sl = [xa.astype(np.float64), xb.astype(np.float64)]
                        part = np.zeros((cout, TZ, TY, TX))
                        for (sa, sb) in ((0, 0), (0, 1), (1, 0)):
                            wv = Ws[sa][:, ch * 16:ch * 16 + 16]
                            for kz in range(3):
                                for ky in range(3):
                                    for kx in range(3):
                                        src = sl[sb][:, kz:kz + TZ, ky:ky + TY, kx + hl - 1:kx + hl - 1 + TX]
                                        part += np.einsum("oc,cdhw->odhw", wv[:, :, kz, ky, kx], src)
                        out += part * inv
                    dz, dy, dx = min(TZ, D - z0), min(TY, H - y0), min(TX, W - x0)
                    acc[b, :, z0:z0 + dz, y0:y0 + dy, x0:x0 + dx] = out[:, :dz, :dy, :dx]
    y = acc * scale[None, :, None, None, None] + shift[None, :, None, None, None]
    return np.where(y > 0, y, y * slope)


def emulate_conv2d_ci_splitf16(packed, x, c, slope=0.01, tile=(16, 16), halo_x=(2, 2), cout=None):
    """Data flow of conv2d_ci_sf_kernel in float64 (the 2D sibling of emulate_conv_ci_splitf16): lane images
    [chunk][step][row block][slice][lane][8 f16] decoded lane by lane (tap 2 m + (kb >> 1) of the 9, tap 9 zero), per 16 x 16 output tile
    and chunk of 16 input channels the halo tile (y0-1..y0+16, x0-2..x0+17) scaled, split, multiplied (aa, ab, ba), unscaled, summed.
    x (N, c, H, W) float32 numpy -> (N, c, H, W)."""
    import numpy as np
    raw = np.asarray(packed, dtype=np.uint8)
    cout = c if cout is None else cout
    nch, nrb = c // 16, cout // 16
    body = nch * 5 * nrb * 2 * 64 * 8 * 2
    img = raw[:body].view(np.float16).reshape(nch, 5, nrb, 2, 64, 8).astype(np.float64)
    tail = raw[body:body + 8 * cout].view(np.float32).astype(np.float64)
    scale, shift = tail[:cout], tail[cout:]
    Ws = np.zeros((2, cout, c, 10))
    for ch in range(nch):
        for st in range(5):
            for rb in range(nrb):
                for lane in range(64):
                    i, kb = lane & 15, lane >> 4
                    Ws[:, 16 * rb + i, 16 * ch + 8 * (kb & 1):16 * ch + 8 * (kb & 1) + 8, 2 * st + (kb >> 1)] = img[ch, st, rb, :, lane, :]
    assert not Ws[:, :, :, 9].any()
    Ws = Ws[:, :, :, :9].reshape(2, cout, c, 3, 3)
    N, _, H, W = x.shape
    TY, TX = tile
    hl, hr = halo_x
    py, px = ((H + TY - 1) // TY) * TY - H, ((W + TX - 1) // TX) * TX - W
    xp = np.pad(x.astype(np.float32), ((0, 0), (0, 0), (1, 1 + py), (hl, hr + px)))
    acc = np.zeros((N, cout, H, W))
    for n in range(N):
        for y0 in range(0, H, TY):
            for x0 in range(0, W, TX):
                out = np.zeros((cout, TY, TX))
                for ch in range(nch):
                    halo = xp[n, ch * 16:ch * 16 + 16, y0:y0 + TY + 2, x0:x0 + TX + hl + hr]
                    e = max(int(np.abs(halo).max().view(np.uint32)) >> 23, 15)
                    mult, inv = np.float32(2.0) ** (141 - e), 2.0 ** (e - 141)
                    xs = halo * mult
                    xa = xs.astype(np.float16)
                    xb = (xs - xa.astype(np.float32)).astype(np.float16)
                    sl = [xa.astype(np.float64), xb.astype(np.float64)]
                    part = np.zeros((cout, TY, TX))
                    for (sa, sb) in ((0, 0), (0, 1), (1, 0)):
                        wv = Ws[sa][:, ch * 16:ch * 16 + 16]
                        for ky in range(3):
                            for kx in range(3):
                                part += np.einsum("oc,chw->ohw", wv[:, :, ky, kx], sl[sb][:, ky:ky + TY, kx + hl - 1:kx + hl - 1 + TX])
                    out += part * inv
                dy, dx = min(TY, H - y0), min(TX, W - x0)
                acc[n, :, y0:y0 + dy, x0:x0 + dx] = out[:, :dy, :dx]
    y = acc * scale[None, :, None, None] + shift[None, :, None, None]
    return np.where(y > 0, y, y * slope)


def conv2d_ci_sf_lds_cycles():
    """Tap-read cycles of one ds_read_b128 of conv2d_ci_sf_kernel for its two lane-half distances [4 = conflict-free] with the planes
    padded to 368 units (a multiple of 256 B), and with the unpadded 360 units (the two channel halves then collide)."""
    RS = 20

    def reads(nvox):
        return [_b128_cycles(lambda l, d=dist: 4 * (((l >> 4) & 1) * nvox + (l & 15) + 1 + (l >> 5) * d), _B128_GROUPS) for dist in (1, RS - 2)]
    return reads(368), reads(360)


def conv_ci_sf_lds_cycles():
    """(tap-read cycles of one ds_read_b128 of conv_ci_sf_kernel for every step's lane-half distance [4 = conflict-free],
    staging-write cycles of one ds_write_b128 pair in the kernel's order and in the plain order [8 = conflict-free per write])."""
    RS, IY, NVOX = 20, 6, 720
    reads = []
    for dist in (1, RS - 2, (IY - 2) * RS - 2):
        reads.append(_b128_cycles(lambda l, d=dist: 4 * (((l >> 4) & 1) * NVOX + (l & 15) + 1 + (l >> 5) * d), _B128_GROUPS))

    def write_cycles(order):
        tot = 0
        for which in range(2):
            for g0 in range(0, 64, 8):
                banks = {}
                for i in range(g0, g0 + 8):
                    swp = ((i >> 2) & 1) if order else 0
                    row, g = divmod(i, 10)
                    unit = row * RS + 2 * g + (swp if which == 0 else 1 - swp)
                    banks.setdefault(unit % 8, set()).add(unit)
                tot += max(len(v) for v in banks.values())
        return tot
    return reads, write_cycles(True), write_cycles(False)


def conv0_sb_slot(x):
    """16-byte slot of column x inside a staged row (SbCfg::slot)."""
    return x ^ (((x >> 3) & 1) << 1)


def conv0_sb_lds_cycles():
    """(tap-read cycles of one ds_read_b128 [4 = minimum], staging-write cycles summed over the four ds_write_b128 of an item
    [32 = minimum; 100 without the swizzle at this row stride]) for the kernel's layout: slot(x) within a row, 41 slots per row."""
    read = _b128_cycles(lambda l: 4 * conv0_sb_slot(2 * (l & 15) + (l >> 4) + 3), _B128_GROUPS)

    def write(slot_fn):
        tot = 0
        for j in range(4):
            for g0 in range(0, 64, 8):
                banks = {}
                for i in range(g0, g0 + 8):
                    row, gi = divmod(i, 10)
                    s = row * 41 + slot_fn(4 * gi + j)
                    banks.setdefault(s % 8, set()).add(s)
                tot += max(len(v) for v in banks.values())
        return tot
    return read, write(conv0_sb_slot), write(lambda x: x)


# ---- fuse_view_paired_kernel (csrc/fusion.hip): the two taps of a row from one 8-byte load -------------------------------------
def fusion_paired_taps(depth, image, ix, iy):
    """Host model of the tap fetch of fuse_view_paired_kernel for one source view: depth (H, W) float32, image (H, W, 3) uint8,
    integer tap origins ix, iy (any int32: the kernel clamps).  Returns (4 depth taps, 4 x 3 colour taps) in the kernel's tap order
    (y0x0, y0x1, y1x0, y1x1) fetched the way the kernel does - pair base bx = min(cx0, W - 2), one little-endian 8-byte word holding
    six colour bytes, read 2 bytes early where it would run past the view - from the CLAMPED addresses (zeroing of outside taps comes
    after, as in fuse_view_kernel)."""
    H, W = depth.shape
    hw = H * W
    dflat = depth.reshape(-1)
    bflat = image.reshape(-1)
    cx0, cx1 = min(max(ix, 0), W - 1), min(max(ix + 1, 0), W - 1)
    cy0, cy1 = min(max(iy, 0), H - 1), min(max(iy + 1, 0), H - 1)
    bx = min(cx0, W - 2)
    hi0, hi1 = cx0 != bx, cx1 != bx
    dtaps, ctaps = [], []
    for cy in (cy0, cy1):
        r = cy * W + bx
        pair = (dflat[r], dflat[r + 1])                      # 8 bytes at a 4-byte aligned address: never past the view (r + 1 <= hw - 1)
        back = 2 if r >= hw - 2 else 0
        start = 3 * r - back
        assert 0 <= start and start + 8 <= 3 * hw, "the 8-byte colour load leaves the view"
        word = int.from_bytes(bflat[start:start + 8].tobytes(), "little") >> (8 * back)
        for hi in (hi0, hi1):
            dtaps.append(pair[1] if hi else pair[0])
            w32 = (word >> 24 if hi else word) & 0xFFFFFFFF
            ctaps.append([(w32 >> (8 * c)) & 255 for c in range(3)])
    return dtaps, ctaps


def emulate_conv0_zmarch(packed, x, cin, slope=0.01, patch=(16, 32), halo_x=(4, 4), zlen=None):
    """Data flow of conv0_zm_kernel (csrc/conv0_zmarch.hip) in float64: a staged UNIT is one input plane of a 16 x 32 (y, x) patch with its
    halo (y0-1..y0+16, x0-4..x0+35; zero outside the volume) and one chunk of 8 channels; it is scaled by the power of two that puts its
    largest magnitude into [2^14, 2^15), split into f16(x') and f16(x' - f16(x')), multiplied (aa, ab, ba) with the packed lane images'
    weight slices of ALL three kz - input plane zi is tap kz of output plane zi + 1 - kz - unscaled by 2^-kx and added to that output
    plane.  zlen: z segment length (None: the whole depth); a segment's first / last halo plane contributes only inside the segment, so
    the segmentation does not change a single number.  x (B, cin, D, H, W) float32 numpy -> (B, 8, D, H, W)."""
    import numpy as np
    raw = np.asarray(packed, dtype=np.uint8)
    nch = cin // 8
    body = nch * 9 * 2 * 64 * 8 * 2
    img = raw[:body].view(np.float16).reshape(nch, 9, 2, 64, 8).astype(np.float64)
    tail = raw[body:body + 64].view(np.float32).astype(np.float64)
    scale, shift = tail[:8], tail[8:16]
    B, _, D, H, W = x.shape
    TY, TX = patch
    hl, hr = halo_x
    px = ((W + TX - 1) // TX) * TX - W
    xp = np.pad(x.astype(np.float32), ((0, 0), (0, 0), (0, 0), (1, TY + 1), (hl, hr + px)))
    zlen = D if zlen is None else zlen
    acc = np.zeros((B, 8, D, H, W))
    for b in range(B):
        for zs in range(0, D, zlen):
            ze = min(zs + zlen, D)
            for y0 in range(0, H, TY):
                for x0 in range(0, W, TX):
                    out = np.zeros((8, D + 2, TY, TX))                     # output planes -1 .. D (the out-of-range ones are dropped)
                    for zi in range(max(zs - 1, 0), min(ze, D - 1) + 1):
                        for ch in range(nch):
                            halo = xp[b, ch * 8:ch * 8 + 8, zi, y0:y0 + TY + 2, x0:x0 + TX + hl + hr]
                            e = max(int(np.abs(halo).max().view(np.uint32)) >> 23, 15)
                            mult, inv = np.float32(2.0) ** (141 - e), 2.0 ** (e - 141)
                            xs = halo * mult
                            xa = xs.astype(np.float16)
                            xb = (xs - xa.astype(np.float32)).astype(np.float16)
                            sl = [xa.astype(np.float64), xb.astype(np.float64)]
                            for r9 in range(9):
                                kz, ky = divmod(r9, 3)
                                zo = zi + 1 - kz
                                if not (zs <= zo < ze):
                                    continue
                                part = np.zeros((8, TY, TX))
                                for (sa, sb) in SF_TERMS[:3]:
                                    A = img[ch, r9, sa].reshape(4, 16, 8)                       # [u][i][ci]
                                    for u in range(4):
                                        for s in range(2):
                                            wrow = A[u, s::2, :]                                # (co, ci); zero where u - s is not a tap
                                            if not wrow.any():
                                                continue
                                            src = sl[sb][:, ky:ky + TY, u - 1 + hl:u - 1 + hl + TX:2]
                                            part[:, :, s::2] += np.einsum("oc,chw->ohw", wrow, src)
                                out[:, zo + 1] += part * inv
                    dy, dx = min(TY, H - y0), min(TX, W - x0)
                    acc[b, :, zs:ze, y0:y0 + dy, x0:x0 + dx] = out[:, zs + 1:ze + 1, :dy, :dx]
    y = acc * scale[None, :, None, None, None] + shift[None, :, None, None, None]
    return np.where(y > 0, y, y * slope)


def emulate_deconv11_splitf16(packed, x, skip=None, slope=0.01, tile=(4, 8, 32)):
    """Data flow of deconv11_sf_kernel (csrc/deconv11_splitf16.hip) in float64.  The lane images [kz * 3 + ky][slice][lane][8 f16] are decoded into the two
    weight slices W_s[ci][co][kz][ky][kx] exactly as the kernel's lanes meet them (lane (i, kb): co = i >> 1, px = i & 1, dx = kb >> 1,
    ci = 8 (kb & 1) + e; kx = 1 / 2 / 0 for (px, dx) = (0, 0) / (1, 0) / (1, 1), and (0, 1) must be zero); per output tile the input box
    (tz0/2 .. +2, ty0/2 .. +4, tx0/2 .. +17; zero beyond the volume) is scaled by the power of two that puts its largest magnitude into
    [2^14, 2^15), split into two float16 slices and multiplied (aa, ab, ba) tap by tap - out[o] += in[i] w[k] with o = 2 i - 1 + k per axis -,
    unscaled; scale (which carries 2^-kw) / shift / leaky-relu, then + skip.  x (B, 16, Di, Hi, Wi) float32 numpy -> (B, 8, 2 Di, 2 Hi, 2 Wi)."""
    import numpy as np
    raw = np.asarray(packed, dtype=np.uint8)
    body = 9 * 2 * 64 * 8 * 2
    img = raw[:body].view(np.float16).reshape(9, 2, 64, 8).astype(np.float64)
    tail = raw[body:body + 64].view(np.float32).astype(np.float64)
    scale, shift = tail[:8], tail[8:16]
    Ws = np.zeros((2, 16, 8, 3, 3, 3))
    for r9 in range(9):
        for lane in range(64):
            i, kb = lane & 15, lane >> 4
            co, px, dx = i >> 1, i & 1, kb >> 1
            kx = {(0, 0): 1, (1, 0): 2, (1, 1): 0}.get((px, dx))
            vals = img[r9, :, lane, :]
            if kx is None:
                assert not vals.any(), "the (even x, next input) quarter of a lane image must be zero"
                continue
            Ws[:, 8 * (kb & 1):8 * (kb & 1) + 8, co, r9 // 3, r9 % 3, kx] = vals
    B, _, Di, Hi, Wi = x.shape
    TZ, TY, TX = tile
    Do, Ho, Wo = 2 * Di, 2 * Hi, 2 * Wi
    bz, by, bx = TZ // 2 + 1, TY // 2 + 1, TX // 2 + 1
    xp = np.pad(x.astype(np.float32), ((0, 0), (0, 0), (0, bz + TZ), (0, by + TY), (0, bx + TX)))
    acc = np.zeros((B, 8, Do, Ho, Wo))
    for b in range(B):
        for z0 in range(0, Do, TZ):
            for y0 in range(0, Ho, TY):
                for x0 in range(0, Wo, TX):
                    box = xp[b, :, z0 // 2:z0 // 2 + bz, y0 // 2:y0 // 2 + by, x0 // 2:x0 // 2 + bx + 1]   # 3 x 5 x 18: what the kernel stages
                    e = max(int(np.abs(box).max().view(np.uint32)) >> 23, 15)
                    mult, inv = np.float32(2.0) ** (141 - e), 2.0 ** (e - 141)
                    xs = box * mult
                    xa = xs.astype(np.float16)
                    xb = (xs - xa.astype(np.float32)).astype(np.float16)
                    sl = [xa.astype(np.float64), xb.astype(np.float64)]
                    out = np.zeros((8, TZ, TY, TX))
                    for (sa, sb) in SF_TERMS[:3]:
                        for kz in range(3):
                            for ky in range(3):
                                for kx in range(3):
                                    w = Ws[sa, :, :, kz, ky, kx]                          # (ci, co)
                                    # outputs o = 2 i - 1 + k inside the tile, i relative to the box origin (= tile origin / 2)
                                    for iz in range(bz):
                                        oz = 2 * iz - 1 + kz
                                        if not (0 <= oz < TZ):
                                            continue
                                        for iy in range(by):
                                            oy = 2 * iy - 1 + ky
                                            if not (0 <= oy < TY):
                                                continue
                                            ix = np.arange(bx + 1)
                                            ox = 2 * ix - 1 + kx
                                            ok = (ox >= 0) & (ox < TX)
                                            out[:, oz, oy, ox[ok]] += np.einsum("io,iw->ow", w, sl[sb][:, iz, iy, ix[ok]])
                    dz, dy, dx_ = min(TZ, Do - z0), min(TY, Ho - y0), min(TX, Wo - x0)
                    acc[b, :, z0:z0 + dz, y0:y0 + dy, x0:x0 + dx_] = (out * inv)[:, :dz, :dy, :dx_]
    y = acc * scale[None, :, None, None, None] + shift[None, :, None, None, None]
    y = np.where(y > 0, y, y * slope)
    return y + (0 if skip is None else skip.astype(np.float64))


# ---- lane-level transcriptions of the kernels written without a GPU run (round 3): the index arithmetic of every thread, the LDS
# ---- images and the operand / result lanes of v_mfma_f32_16x16x32_f16, so that a wrong offset shows on the CPU
def mfma_16x16x32(a_lanes, b_lanes):
    """D (16 x 16) = A B for one wave: a_lanes / b_lanes (64 lanes, 8 values).  Lane l of A holds row i = l & 15, k = 8 (l >> 4) .. + 8; lane l of
    B holds column j = l & 15, the same k.  (The result lanes: lane l holds column j = l & 15, rows 4 (l >> 4) + r; casmvs_selftest_mfma_f16.)"""
    import numpy as np
    A = np.zeros((16, 32))
    Bm = np.zeros((32, 16))
    for l in range(64):
        A[l & 15, 8 * (l >> 4):8 * (l >> 4) + 8] = a_lanes[l]
        Bm[8 * (l >> 4):8 * (l >> 4) + 8, l & 15] = b_lanes[l]
    return A @ Bm


def split_f16_np(x32, mult):
    import numpy as np
    xs = (x32.astype(np.float32) * np.float32(mult)).astype(np.float32)
    a = xs.astype(np.float16)
    b = (xs - a.astype(np.float32)).astype(np.float16)
    return a.astype(np.float64), b.astype(np.float64)


def tile_scale_np(values):
    """casmvs::tile_scale: mult = 2^kx that puts the largest magnitude into [2^14, 2^15), inv = 2^-kx (exponent field >= 15)."""
    import numpy as np
    e = max(int(np.abs(np.asarray(values, np.float32)).max().view(np.uint32)) >> 23, 15)
    return np.float32(2.0) ** (141 - e), 2.0 ** (e - 141)


def emulate_deconv11_lanes(packed, x, skip, slope=0.01):
    """deconv11_sf_kernel thread by thread (csrc/deconv11_splitf16.hip): staging items -> LDS planes [slice][half][272], the lanes' B units vb[izl][iyr],
    the (kz, ky) -> (output plane, input plane, row) tables, the result lanes' (channel, x parity, column) -> output addresses."""
    import numpy as np
    raw = np.asarray(packed, dtype=np.uint8)
    wl = raw[:9 * 2 * 64 * 16].view(np.float16).reshape(9 * 2 * 64, 8).astype(np.float64)   # [(r9 * 2 + s) * 64 + lane]
    tail = raw[9 * 2 * 64 * 16:][:64].view(np.float32).astype(np.float64)
    B, _, Di, Hi, Wi = x.shape
    Do, Ho, Wo = 2 * Di, 2 * Hi, 2 * Wi
    out = np.full((B, 8, Do, Ho, Wo), np.nan)
    JY, JX, NVOX = 5, 18, 272
    PA, PB = (0, 0, 1), (0, 1, 0)
    for b in range(B):
        for tz0 in range(0, Do, 4):
            for ty0 in range(0, Ho, 8):
                for tx0 in range(0, Wo, 32):
                    # staging: thread e -> (box plane, row, pair of x), 16 channels, 2 voxels
                    R = np.zeros((256, 16, 2), np.float32)
                    vox = np.full(256, -1)
                    for tid in range(135):
                        e_iz, rem = divmod(tid, JY * (JX // 2))
                        e_iy, e_g = divmod(rem, JX // 2)
                        vox[tid] = (e_iz * JY + e_iy) * JX + 2 * e_g
                        gz, gy, gx = tz0 // 2 + e_iz, ty0 // 2 + e_iy, tx0 // 2 + 2 * e_g
                        if gz < Di and gy < Hi and gx < Wi:
                            R[tid] = x[b, :, gz, gy, gx:gx + 2]
                    mult, inv = tile_scale_np(R)
                    act = np.zeros((4 * NVOX, 8))
                    for tid in range(135):
                        for hf in range(2):
                            for p in range(2):
                                sa, sb = split_f16_np(R[tid, hf * 8:hf * 8 + 8, p], mult)
                                act[(0 * 2 + hf) * NVOX + vox[tid] + p] = sa
                                act[(1 * 2 + hf) * NVOX + vox[tid] + p] = sb
                    for wave in range(4):
                        lanes = np.arange(64)
                        jcol, kb = lanes & 15, lanes >> 4
                        half, dx = kb & 1, kb >> 1
                        vb = lambda izl, iyr: half * NVOX + (izl * JY + wave + iyr) * JX + jcol + dx
                        rowv = {(izl, iyr, s): act[s * 2 * NVOX + vb(izl, iyr)] for izl in range(3) for iyr in range(2) for s in range(2)}
                        acc = np.zeros((4, 2, 16, 16))
                        for kz in range(3):
                            for ky in range(3):
                                a = [wl[((kz * 3 + ky) * 2 + s) * 64 + lanes] for s in range(2)]
                                yo, iyr = (0 if ky == 1 else 1), (1 if ky == 0 else 0)
                                for q in range(2):
                                    zl = 2 * q if kz == 1 else 2 * q + 1
                                    izl = q if kz == 1 else (q + 1 if kz == 0 else q)
                                    for p in range(3):
                                        acc[zl, yo] += mfma_16x16x32(a[PA[p]], rowv[(izl, iyr, PB[p])])
                        for zl in range(4):
                            for yo in range(2):
                                oz, oy = tz0 + zl, ty0 + 2 * wave + yo
                                for l in range(64):
                                    u, j = l >> 4, l & 15
                                    ox = tx0 + 2 * j
                                    if not (oz < Do and oy < Ho and ox < Wo):
                                        continue
                                    for h in range(2):
                                        co = 2 * u + h
                                        for ph in range(2):
                                            v = acc[zl, yo, 4 * u + 2 * h + ph, j] * inv * tail[co] + tail[8 + co]
                                            v = v if v > 0 else v * slope
                                            out[b, co, oz, oy, ox + ph] = v + (0.0 if skip is None else skip[b, co, oz, oy, ox + ph])
    assert not np.isnan(out).any(), "an output voxel was never written"
    return out


def _px_slot(x):
    return x ^ (((x >> 3) & 1) << 1)


def emulate_conv0_zmarch_lanes(packed, x, cin, zlen, slope=0.01):
    """conv0_zm_kernel thread by thread (csrc/conv0_zmarch.hip): staging items -> LDS slots (row stride 41, slot swizzle), the lanes' B units, the three
    rotating accumulator sets, the segment bookkeeping, the result lanes -> output addresses."""
    import numpy as np
    raw = np.asarray(packed, dtype=np.uint8)
    nch = cin // 8
    wl = raw[:nch * 9 * 2 * 64 * 16].view(np.float16).reshape(nch * 9 * 2 * 64, 8).astype(np.float64)
    tail = raw[nch * 9 * 2 * 64 * 16:][:64].view(np.float32).astype(np.float64)
    B, _, D, H, W = x.shape
    ROW, NV = 41, 18 * 41
    PA, PB = (0, 0, 1), (0, 1, 0)
    out = np.full((B, 8, D, H, W), np.nan)
    lanes = np.arange(64)
    jcol, u = lanes & 15, lanes >> 4
    for b in range(B):
        for zs in range(0, D, zlen):
            ze = min(zs + zlen, D)
            for ty0 in range(0, H, 16):
                for tx0 in range(0, W, 32):
                    acc = np.zeros((4, 3, 4, 16, 16))                       # [wave][slot][t]
                    for zi in range(zs - 1, ze + 1):
                        if 0 <= zi < D:
                            for ch in range(nch):
                                R = np.zeros((256, 8, 4), np.float32)
                                vox, vxor = np.full(256, -1), np.zeros(256, int)
                                for tid in range(180):
                                    iy, g = divmod(tid, 10)
                                    gy, gx = ty0 - 1 + iy, tx0 - 4 + 4 * g
                                    vox[tid], vxor[tid] = iy * ROW + 4 * g, ((g >> 1) & 1) << 1
                                    if 0 <= gy < H and 0 <= gx < W:
                                        R[tid] = x[b, ch * 8:ch * 8 + 8, zi, gy, gx:gx + 4]
                                mult, inv = tile_scale_np(R)
                                act = np.zeros((2 * NV, 8))
                                for tid in range(180):
                                    for j in range(4):
                                        sa, sb = split_f16_np(R[tid, :, j], mult)
                                        act[0 * NV + vox[tid] + (j ^ vxor[tid])] = sa
                                        act[1 * NV + vox[tid] + (j ^ vxor[tid])] = sb
                                for wave in range(4):
                                    vbase = (4 * wave) * ROW + _px_slot(2 * jcol + u + 3)
                                    row = {(yr, s): act[s * NV + vbase + yr * ROW] for yr in range(6) for s in range(2)}
                                    for kz in range(3):
                                        for ky in range(3):
                                            a = [wl[((ch * 9 + kz * 3 + ky) * 2 + s) * 64 + lanes] for s in range(2)]
                                            for p in range(3):
                                                for t in range(4):
                                                    acc[wave, 2 - kz, t] += mfma_16x16x32(a[PA[p]], row[(t + ky, PB[p])]) * inv
                        zo = zi - 1
                        for wave in range(4):
                            if zs <= zo < ze:
                                for t in range(4):
                                    oy = ty0 + 4 * wave + t
                                    for l in range(64):
                                        uu, j = l >> 4, l & 15
                                        ox = tx0 + 2 * j
                                        if not (oy < H and ox < W):
                                            continue
                                        for h in range(2):
                                            co = 2 * uu + h
                                            for ph in range(2):
                                                v = acc[wave, 0, t, 4 * uu + 2 * h + ph, j] * tail[co] + tail[8 + co]
                                                out[b, co, zo, oy, ox + ph] = v if v > 0 else v * slope
                            acc[wave, 0], acc[wave, 1] = acc[wave, 1].copy(), acc[wave, 2].copy()
                            acc[wave, 2] = 0.0
    assert not np.isnan(out).any(), "an output voxel was never written"
    return out


def emulate_deconv9_lanes(packed, x, skip, slope=0.01):
    """deconv9_sf_kernel thread by thread (csrc/deconv9_splitf16.hip): staging items (16-channel half, box voxel pair) -> LDS planes [slice][quarter][192],
    the lanes' B units (and the next voxel for the kx = 0 tap), the (kz, ky, kx) -> (output plane, row, x parity, input plane, row) tables, the result lanes."""
    import numpy as np
    raw = np.asarray(packed, dtype=np.uint8)
    nw = 9 * 3 * 2 * 64
    wl = raw[:nw * 16].view(np.float16).reshape(nw, 8).astype(np.float64)   # [((r9 * 3 + kx) * 2 + s) * 64 + lane]
    tail = raw[nw * 16:][:128].view(np.float32).astype(np.float64)
    B, _, Di, Hi, Wi = x.shape
    Do, Ho, Wo = 2 * Di, 2 * Hi, 2 * Wi
    out = np.full((B, 16, Do, Ho, Wo), np.nan)
    JY, JX, NVOX = 5, 18, 192
    PA, PB = (0, 0, 1), (0, 1, 0)
    lanes = np.arange(64)
    jcol, kb = lanes & 15, lanes >> 4
    for b in range(B):
        for tz0 in range(0, Do, 2):
            for ty0 in range(0, Ho, 8):
                for tx0 in range(0, Wo, 32):
                    R = np.zeros((256, 16, 2), np.float32)
                    vox, ehf = np.full(256, -1), np.zeros(256, int)
                    for tid in range(180):
                        e_hf, e_pr = divmod(tid, 90)
                        e_iz, rem = divmod(e_pr, JY * (JX // 2))
                        e_iy, e_g = divmod(rem, JX // 2)
                        vox[tid], ehf[tid] = (e_iz * JY + e_iy) * JX + 2 * e_g, e_hf
                        gz, gy, gx = tz0 // 2 + e_iz, ty0 // 2 + e_iy, tx0 // 2 + 2 * e_g
                        if gz < Di and gy < Hi and gx < Wi:
                            R[tid] = x[b, e_hf * 16:e_hf * 16 + 16, gz, gy, gx:gx + 2]
                    mult, inv = tile_scale_np(R)
                    act = np.zeros((8 * NVOX, 8))
                    for tid in range(180):
                        for hh in range(2):
                            for p in range(2):
                                sa, sb = split_f16_np(R[tid, hh * 8:hh * 8 + 8, p], mult)
                                act[(0 * 4 + ehf[tid] * 2 + hh) * NVOX + vox[tid] + p] = sa
                                act[(1 * 4 + ehf[tid] * 2 + hh) * NVOX + vox[tid] + p] = sb
                    for wave in range(4):
                        vb = lambda izl, iyr: kb * NVOX + (izl * JY + wave + iyr) * JX + jcol
                        acc = np.zeros((2, 2, 2, 16, 16))
                        for kz in range(3):
                            for ky in range(3):
                                zl, izl = (0 if kz == 1 else 1), (1 if kz == 0 else 0)
                                yo, iyr = (0 if ky == 1 else 1), (1 if ky == 0 else 0)
                                for kx in range(3):
                                    a = [wl[(((kz * 3 + ky) * 3 + kx) * 2 + s) * 64 + lanes] for s in range(2)]
                                    px, nx = (0 if kx == 1 else 1), (1 if kx == 0 else 0)
                                    for p in range(3):
                                        acc[zl, yo, px] += mfma_16x16x32(a[PA[p]], act[PB[p] * 4 * NVOX + vb(izl, iyr) + nx])
                        for zl in range(2):
                            for yo in range(2):
                                oz, oy = tz0 + zl, ty0 + 2 * wave + yo
                                for l in range(64):
                                    u, j = l >> 4, l & 15
                                    ox = tx0 + 2 * j
                                    if not (oz < Do and oy < Ho and ox < Wo):
                                        continue
                                    for r in range(4):
                                        co = 4 * u + r
                                        for px in range(2):
                                            v = acc[zl, yo, px, 4 * u + r, j] * inv * tail[co] + tail[16 + co]
                                            v = v if v > 0 else v * slope
                                            out[b, co, oz, oy, ox + px] = v + (0.0 if skip is None else skip[b, co, oz, oy, ox + px])
    assert not np.isnan(out).any(), "an output voxel was never written"
    return out
