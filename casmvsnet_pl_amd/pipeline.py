"""Input side of the hot path: what the reference's datasets and eval loop do between the files and
`CascadeMVSNet.forward` - camera files -> per-level projection matrices (datasets/dtu.py:51-96), the relative matrices a
sample carries (dtu.py:181-186), image normalisation (dtu.py:134-137), batching of reference views with per-sample
depth ranges (train.py's DataLoader collate), PFM depth / confidence files (datasets/utils.py, eval.py:226-227) - plus
what the MI355X adds: the images travel as uint8 (4x fewer PCIe bytes) and are normalised by a HIP kernel, and a
double-buffered prefetcher overlaps the host->device copies of batch i+1 with the forward of batch i on a second stream.

`DTUReader` reads a DTU-format tree the way datasets/dtu.py does (pair file, camera files, PIL-decoded images, PFM depth
maps and visibility masks with cv2's nearest-neighbour down-sampling restated in numpy) and yields samples with uint8
images, ready for `collate` + `DevicePrefetcher`.
"""
import os
import ctypes
import re

import numpy as np
import torch

from . import _io, _lib, streams

IMAGENET_MEAN = (0.485, 0.456, 0.406)   # dtu.py:135-136
IMAGENET_STD = (0.229, 0.224, 0.225)


# ---- cameras --------------------------------------------------------------------------------------------------------

def read_cam_file(filename):
    """datasets/dtu.py:79-91: MVSNet camera text file -> intrinsics (3,3) f32, extrinsics (4,4) f32, depth_min."""
    with open(filename) as f:
        lines = [line.rstrip() for line in f.readlines()]
    extrinsics = np.array(" ".join(lines[1:5]).split(), dtype=np.float32).reshape(4, 4)
    intrinsics = np.array(" ".join(lines[7:10]).split(), dtype=np.float32).reshape(3, 3)
    return intrinsics, extrinsics, float(lines[11].split()[0])


def build_proj_mats(intrinsics_coarsest, extrinsics, levels=3):
    """datasets/dtu.py:66-74: 4x4 projection matrices of one view for every level, fine -> coarse.  `intrinsics_coarsest`
    are the intrinsics at the COARSEST level (1/4 resolution for 3 levels); each finer level doubles the first two rows."""
    K = np.array(intrinsics_coarsest, dtype=np.float32).copy()
    E = np.asarray(extrinsics, dtype=np.float32)
    mats = []
    for _ in range(levels):                       # coarse -> fine
        P = np.eye(4)
        P[:3, :4] = K @ E[:3, :4]
        mats.append(torch.tensor(P, dtype=torch.float32))
        K[:2] *= 2
    return torch.stack(mats[::-1])                # (levels, 4, 4) fine -> coarse


def relative_proj_mats(proj_ref, proj_srcs):
    """datasets/dtu.py:181-186: (P_src,l @ inverse(P_ref,l))[:3] per source view and level -> (V-1, levels, 3, 4)."""
    ref_inv = torch.inverse(proj_ref)
    return torch.stack([p @ ref_inv for p in proj_srcs])[:, :, :3]


# ---- files of a DTU-format scene (datasets/dtu.py) ------------------------------------------------------------------

def _decode_file(filename, channels):
    """(H, W, 3) / (H, W) uint8 of an image file: PNGs go through libcasmvs_io.so (include/casmvs_io.h: byte-identical to
    PIL's `Image.open(f).convert("RGB" / "L")`, 2-3x faster and outside the GIL); other formats - and the PNG variants that
    library declines (interlaced, 1/2/4/16-bit samples) - are decoded by PIL as in the reference."""
    if str(filename).lower().endswith(".png"):
        with open(filename, "rb") as f:
            data = f.read()
        arr = _io.decode_png(data, channels)
        if arr is not None:
            return arr
    from PIL import Image
    return np.array(Image.open(filename).convert("RGB" if channels == 3 else "L"), dtype=np.uint8)   # a writable copy (torch.from_numpy)


def read_image_u8(filename, img_wh=None):
    """dtu.py:168-170: decode (RGB), optionally PIL's bilinear resize to (W, H) -> (H, W, 3) uint8.  The normalisation
    (dtu.py:134-137) happens on the device (normalize_images_u8)."""
    arr = _decode_file(filename, 3)
    if img_wh is not None and (arr.shape[1], arr.shape[0]) != tuple(img_wh):   # PIL's resize to the same size is a copy
        from PIL import Image
        arr = np.array(Image.fromarray(arr).resize(tuple(img_wh), Image.BILINEAR), dtype=np.uint8)
    return arr


def read_images_u8(filenames, img_wh=None):
    """The views of one sample -> (V, H, W, 3) uint8 tensor.  PNG views are decoded straight into their slice of the sample's
    array (no per-view copy, no stack); anything else - other formats, PNG variants libcasmvs_io.so declines, views that
    need the resize - goes through read_image_u8."""
    out = None
    for i, filename in enumerate(filenames):
        if str(filename).lower().endswith(".png"):
            with open(filename, "rb") as f:
                data = f.read()
            info = _io.png_info(data)
            if info is not None and (img_wh is None or tuple(info[:2]) == tuple(img_wh)):
                if out is None:
                    out = np.empty((len(filenames), info[1], info[0], 3), np.uint8)
                if out.shape[1:3] == (info[1], info[0]) and _io.decode_png(data, 3, out=out[i]) is not None:
                    continue
        arr = read_image_u8(filename, img_wh)
        if out is None:
            out = np.empty((len(filenames),) + arr.shape, np.uint8)
        out[i] = arr      # a view of another size raises here, as torch.stack did
    return torch.from_numpy(out)


def resize_nearest(a, out_hw=None, fx=None, fy=None):
    """cv2.resize(..., interpolation=cv2.INTER_NEAREST) as dtu.py:94-129 uses it: source index = min(floor(dst * scale),
    n - 1) with scale = 1 / fx (or n_src / n_dst) in double, per axis."""
    a = np.asarray(a)
    H, W = a.shape[:2]
    if out_hw is None:
        out_hw = (int(round(H * fy)), int(round(W * fx)))
        sy, sx = 1.0 / fy, 1.0 / fx
    else:
        sy, sx = H / out_hw[0], W / out_hw[1]
    iy = np.minimum(np.floor(np.arange(out_hw[0]) * sy).astype(np.int64), H - 1)
    ix = np.minimum(np.floor(np.arange(out_hw[1]) * sx).astype(np.int64), W - 1)
    return a[iy][:, ix]


def _pyramid(level0):
    l1 = resize_nearest(level0, fx=0.5, fy=0.5)
    return level0, l1, resize_nearest(l1, fx=0.5, fy=0.5)


class DTUReader:
    """datasets/dtu.py as a plain reader: `len()` samples, `reader[i]` -> the dict DTUDataset.__getitem__ returns, with the
    images as uint8 (`imgs_u8` (V,H,W,3)) instead of normalised floats.  `scans`: the list dtu.py reads from
    datasets/lists/dtu/<split>.txt.  img_wh=None: training layout (640x512 crops, all 7 light conditions, ground-truth
    depths and masks); img_wh=(W,H): test layout (full images resized, light condition 3, intrinsics rescaled)."""

    def __init__(self, root_dir, scans, n_views=3, levels=3, depth_interval=2.65, img_wh=None, n_cameras=49):
        if img_wh is not None and (img_wh[0] % 32 or img_wh[1] % 32):
            raise ValueError("img_wh must both be multiples of 32!")          # dtu.py:21-23
        self.root_dir, self.scans, self.n_views, self.levels = root_dir, list(scans), n_views, levels
        self.depth_interval, self.img_wh = depth_interval, img_wh
        self.metas = []                                                       # dtu.py:30-49
        light_idxs = [3] if img_wh else range(7)
        with open(os.path.join(root_dir, "Cameras/pair.txt")) as f:
            lines = [l.rstrip() for l in f.readlines()]
        n = int(lines[0])
        pairs = [(int(lines[1 + 2 * i]), [int(x) for x in lines[2 + 2 * i].split()[1::2]]) for i in range(n)]
        for scan in self.scans:
            for ref_view, src_views in pairs:
                for light_idx in light_idxs:
                    self.metas.append((scan, light_idx, ref_view, src_views))
        self.proj_mats = []                                                   # dtu.py:51-77
        for vid in range(n_cameras):
            name = f"Cameras/train/{vid:08d}_cam.txt" if img_wh is None else f"Cameras/{vid:08d}_cam.txt"
            path = os.path.join(root_dir, name)
            if not os.path.isfile(path):
                self.proj_mats.append(None)
                continue
            K, E, depth_min = read_cam_file(path)
            if img_wh is not None:                                            # to the coarsest level of the resized image
                K[0] *= img_wh[0] / 1600 / 4
                K[1] *= img_wh[1] / 1200 / 4
            self.proj_mats.append((build_proj_mats(K, E, levels), depth_min))

    def __len__(self):
        return len(self.metas)

    def read_depth(self, filename):
        """dtu.py:93-111: (1200,1600) PFM -> level_0 (512,640) crop of the half-size map (or the resized map), level_1, level_2."""
        depth = np.array(read_pfm(filename)[0], dtype=np.float32)
        if self.img_wh is None:
            d0 = resize_nearest(depth, fx=0.5, fy=0.5)[44:556, 80:720]
        else:
            d0 = resize_nearest(depth, out_hw=(self.img_wh[1], self.img_wh[0]))
        return {f"level_{l}": torch.from_numpy(np.ascontiguousarray(d)) for l, d in enumerate(_pyramid(d0))}

    def read_mask(self, filename):
        """dtu.py:113-131: 8-bit visibility image -> boolean masks at the three levels."""
        mask = _decode_file(filename, 1)
        if self.img_wh is None:
            m0 = resize_nearest(mask, fx=0.5, fy=0.5)[44:556, 80:720]
        else:
            m0 = resize_nearest(mask, out_hw=(self.img_wh[1], self.img_wh[0]))
        return {f"level_{l}": torch.from_numpy(np.ascontiguousarray(m)).bool() for l, m in enumerate(_pyramid(m0))}

    def __getitem__(self, idx):                                               # dtu.py:147-192
        scan, light_idx, ref_view, src_views = self.metas[idx]
        view_ids = [ref_view] + src_views[:self.n_views - 1]
        sample, img_files, proj_mats = {}, [], []
        for i, vid in enumerate(view_ids):
            if self.img_wh is None:
                img_files.append(os.path.join(self.root_dir, f"Rectified/{scan}_train/rect_{vid + 1:03d}_{light_idx}_r5000.png"))
            else:
                img_files.append(os.path.join(self.root_dir, f"Rectified/{scan}/rect_{vid + 1:03d}_{light_idx}_r5000.png"))
            proj_mat_ls, depth_min = self.proj_mats[vid]
            if i == 0:
                sample["init_depth_min"] = torch.tensor([depth_min], dtype=torch.float32)
                if self.img_wh is None:
                    sample["masks"] = self.read_mask(os.path.join(self.root_dir, f"Depths/{scan}/depth_visual_{vid:04d}.png"))
                    sample["depths"] = self.read_depth(os.path.join(self.root_dir, f"Depths/{scan}/depth_map_{vid:04d}.pfm"))
                ref_proj = proj_mat_ls
            else:
                proj_mats.append(proj_mat_ls)
        sample["imgs_u8"] = read_images_u8(img_files, self.img_wh)              # (V, H, W, 3) uint8
        sample["proj_mats"] = relative_proj_mats(ref_proj, proj_mats)          # (V-1, levels, 3, 4) fine -> coarse
        sample["depth_interval"] = torch.tensor([self.depth_interval], dtype=torch.float32)
        sample["scan_vid"] = (scan, ref_view)
        return sample


class BlendedMVSReader:
    """datasets/blendedmvs.py as a plain reader: per scan `cams/pair.txt` (reference views with fewer than `n_views` valid
    views are skipped, :46-57), per-scan depth scale 100 / depth_min of the scan's first camera applied to the depth range,
    the extrinsic translations and the rendered depth maps (:98-104,108), intrinsics rescaled to the coarsest level of
    `img_wh` (:73-74), and the PER-SAMPLE depth interval (depth_max - depth_min) / n_depths of BASELINE config 5 (:171).
    Samples carry uint8 images (`imgs_u8`) like DTUReader.  `n_coarse_intervals` is the reference's `depth_interval`
    constructor argument (192: it is a COUNT there)."""

    def __init__(self, root_dir, scans, n_views=3, levels=3, n_coarse_intervals=192.0, img_wh=(768, 576)):
        if img_wh[0] % 32 or img_wh[1] % 32:
            raise ValueError("img_wh must both be multiples of 32!")
        self.root_dir, self.scans, self.n_views, self.levels = root_dir, list(scans), n_views, levels
        self.n_depths, self.img_wh = n_coarse_intervals, img_wh
        self.metas, ref_views = [], {}
        for scan in self.scans:                                                # blendedmvs.py:32-57
            with open(os.path.join(root_dir, scan, "cams/pair.txt")) as f:
                lines = [l.rstrip() for l in f.readlines()]
            ref_views[scan] = []
            for i in range(int(lines[0])):
                ref_view, line = int(lines[1 + 2 * i]), lines[2 + 2 * i].split()
                ref_views[scan].append(ref_view)
                if int(line[0]) < n_views:
                    continue
                self.metas.append((scan, -1, ref_view, [int(x) for x in line[1::2]]))
        low = root_dir.rstrip("/").endswith("dataset_low_res")                 # :62-66
        img_w, img_h = (768, 576) if low else (2048, 1536)
        self.proj_mats, self.scale_factors = {}, {}
        for scan in self.scans:                                                # :67-85
            self.proj_mats[scan] = {}
            for vid in ref_views[scan]:
                K, E, depth_min = read_cam_file(os.path.join(root_dir, scan, f"cams/{vid:08d}_cam.txt"))
                if scan not in self.scale_factors:
                    self.scale_factors[scan] = 100 / depth_min                 # the scan's first camera fixes the scale
                depth_min *= self.scale_factors[scan]
                E[:3, 3] *= self.scale_factors[scan]
                K[0] *= img_wh[0] / img_w / 4
                K[1] *= img_wh[1] / img_h / 4
                self.proj_mats[scan][vid] = (build_proj_mats(K, E, levels), depth_min)

    def __len__(self):
        return len(self.metas)

    def read_depth_and_mask(self, scan, filename, depth_min):                  # :106-128
        depth = np.array(read_pfm(filename)[0], dtype=np.float32)
        depth *= self.scale_factors[scan]                                      # in place on the float32 array, like the reference
        levels_ = _pyramid(resize_nearest(depth, out_hw=(self.img_wh[1], self.img_wh[0])))
        depths = {f"level_{l}": torch.from_numpy(np.ascontiguousarray(d)) for l, d in enumerate(levels_)}
        masks = {f"level_{l}": torch.from_numpy(np.ascontiguousarray(d > depth_min)) for l, d in enumerate(levels_)}
        return depths, masks, float(levels_[0].max())

    def __getitem__(self, idx):                                                # :149-187
        scan, _, ref_view, src_views = self.metas[idx]
        sample, imgs, proj_mats = {}, [], []
        for i, vid in enumerate([ref_view] + src_views[:self.n_views - 1]):
            imgs.append(torch.from_numpy(read_image_u8(os.path.join(self.root_dir, f"{scan}/blended_images/{vid:08d}.jpg"), self.img_wh)))
            proj_mat_ls, depth_min = self.proj_mats[scan][vid]
            if i == 0:
                depths, masks, depth_max = self.read_depth_and_mask(
                    scan, os.path.join(self.root_dir, f"{scan}/rendered_depth_maps/{vid:08d}.pfm"), depth_min)
                sample["init_depth_min"] = torch.tensor([depth_min], dtype=torch.float32)
                sample["depth_interval"] = torch.tensor([(depth_max - depth_min) / self.n_depths], dtype=torch.float32)
                ref_proj = proj_mat_ls
            else:
                proj_mats.append(proj_mat_ls)
        sample["imgs_u8"] = torch.stack(imgs)
        sample["proj_mats"] = relative_proj_mats(ref_proj, proj_mats)
        sample["depths"], sample["masks"] = depths, masks
        sample["scan_vid"] = (scan, ref_view)
        return sample


class TanksReader:
    """datasets/tanks.py as a plain reader (test only, like the reference): `<root>/<split>/<scan>/{pair.txt, cams/, images/}`,
    intrinsics rescaled from the scan's native image size to the coarsest level of `img_wh` (:77-79), and the hand-tuned
    depth interval of every scan (:42-49, :59-64) as the sample's `depth_interval`."""

    IMAGE_SIZES = {"intermediate": {"Family": (1920, 1080), "Francis": (1920, 1080), "Horse": (1920, 1080), "Lighthouse": (2048, 1080),
                                    "M60": (2048, 1080), "Panther": (2048, 1080), "Playground": (1920, 1080), "Train": (1920, 1080)},
                   "advanced": {s: (1920, 1080) for s in ("Auditorium", "Ballroom", "Courtroom", "Museum", "Palace", "Temple")}}
    DEPTH_INTERVALS = {"intermediate": {"Family": 2.5e-3, "Francis": 1e-2, "Horse": 1.5e-3, "Lighthouse": 1.5e-2, "M60": 5e-3,
                                        "Panther": 5e-3, "Playground": 7e-3, "Train": 5e-3},
                       "advanced": {"Auditorium": 3e-2, "Ballroom": 2e-2, "Courtroom": 2e-2, "Museum": 2e-2, "Palace": 1e-2, "Temple": 1e-2}}

    def __init__(self, root_dir, split="intermediate", scans=None, n_views=3, levels=3, img_wh=(1152, 864)):
        if img_wh[0] % 32 or img_wh[1] % 32:
            raise ValueError("img_wh must both be multiples of 32!")
        self.root_dir, self.split, self.n_views, self.levels, self.img_wh = root_dir, split, n_views, levels, img_wh
        self.image_sizes, self.depth_interval = self.IMAGE_SIZES[split], self.DEPTH_INTERVALS[split]
        self.scans = list(scans) if scans is not None else list(self.image_sizes)
        self.metas, self.proj_mats = [], {}
        for scan in self.scans:
            with open(os.path.join(root_dir, split, scan, "pair.txt")) as f:
                lines = [l.rstrip() for l in f.readlines()]
            img_w, img_h = self.image_sizes[scan]
            self.proj_mats[scan] = {}
            for i in range(int(lines[0])):
                ref_view = int(lines[1 + 2 * i])
                self.metas.append((scan, -1, ref_view, [int(x) for x in lines[2 + 2 * i].split()[1::2]]))
                K, E, depth_min = read_cam_file(os.path.join(root_dir, split, scan, f"cams/{ref_view:08d}_cam.txt"))
                K[0] *= img_wh[0] / img_w / 4
                K[1] *= img_wh[1] / img_h / 4
                self.proj_mats[scan][ref_view] = (build_proj_mats(K, E, levels), depth_min)

    def __len__(self):
        return len(self.metas)

    def __getitem__(self, idx):                                                # tanks.py:131-163
        scan, _, ref_view, src_views = self.metas[idx]
        sample, imgs, proj_mats = {}, [], []
        for i, vid in enumerate([ref_view] + src_views[:self.n_views - 1]):
            imgs.append(torch.from_numpy(read_image_u8(os.path.join(self.root_dir, self.split, scan, f"images/{vid:08d}.jpg"), self.img_wh)))
            proj_mat_ls, depth_min = self.proj_mats[scan][vid]
            if i == 0:
                ref_proj = proj_mat_ls
                sample["init_depth_min"] = torch.tensor([depth_min], dtype=torch.float32)
                sample["depth_interval"] = torch.tensor([self.depth_interval[scan]], dtype=torch.float32)
            else:
                proj_mats.append(proj_mat_ls)
        sample["imgs_u8"] = torch.stack(imgs)
        sample["proj_mats"] = relative_proj_mats(ref_proj, proj_mats)
        sample["scan_vid"] = (scan, ref_view)
        return sample


# ---- PFM -----------------------------------------------------------------------------------------------------------------

def read_pfm(filename):
    """datasets/utils.py:5-40: -> (array float32 (H,W) or (H,W,3), top row first; scale)."""
    with open(filename, "rb") as f:
        kind = f.readline().decode("utf-8").rstrip()
        if kind not in ("PF", "Pf"):
            raise ValueError("Not a PFM file.")
        m = re.match(r"^(\d+)\s(\d+)\s$", f.readline().decode("utf-8"))
        if not m:
            raise ValueError("Malformed PFM header.")
        width, height = int(m.group(1)), int(m.group(2))
        scale = float(f.readline().rstrip())
        data = np.fromfile(f, ("<" if scale < 0 else ">") + "f")
    shape = (height, width, 3) if kind == "PF" else (height, width)
    return np.flipud(data.reshape(shape)), abs(scale)


def save_pfm(filename, image, scale=1):
    """datasets/utils.py:43-69: float32 (H,W) / (H,W,1) / (H,W,3), stored bottom row first, little endian = negative scale."""
    image = np.asarray(image)
    if image.dtype != np.float32:
        raise ValueError("Image dtype must be float32.")
    color = image.ndim == 3 and image.shape[2] == 3
    if not color and not (image.ndim == 2 or (image.ndim == 3 and image.shape[2] == 1)):
        raise ValueError("Image must have H x W x 3, H x W x 1 or H x W dimensions.")
    little = image.dtype.byteorder == "<" or (image.dtype.byteorder == "=" and np.little_endian)
    with open(filename, "wb") as f:
        f.write(b"PF\n" if color else b"Pf\n")
        f.write(f"{image.shape[1]} {image.shape[0]}\n".encode("utf-8"))
        f.write(("%f\n" % (-scale if little else scale)).encode("utf-8"))
        np.flipud(image).tofile(f)


# ---- device side -----------------------------------------------------------------------------------------------------------

def normalize_images_u8(images_u8, mean=IMAGENET_MEAN, std=IMAGENET_STD):
    """(..., H, W, 3) uint8 device tensor -> (..., 3, H, W) float32 = T.Normalize(mean, std)(T.ToTensor()(img))
    (dtu.py:134-137), one casmvs_normalize_images_u8 launch."""
    if not images_u8.is_cuda or images_u8.dtype != torch.uint8 or images_u8.shape[-1] != 3:
        raise RuntimeError("normalize_images_u8: expected a uint8 (..., H, W, 3) tensor on the MI355X (no CPU fallback)")
    x = images_u8.contiguous()
    lead, (H, W) = x.shape[:-3], x.shape[-3:-1]
    N = int(np.prod(lead)) if lead else 1
    out = torch.empty(tuple(lead) + (3, H, W), dtype=torch.float32, device=x.device)
    m = (ctypes.c_float * 3)(*mean)
    s = (ctypes.c_float * 3)(*std)
    with torch.cuda.device(x.device):
        rc = _lib.load().casmvs_normalize_images_u8(ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(out.data_ptr()), N, H, W, m, s,
                                                    streams.launch_stream(x))
    _lib.check(rc, "casmvs_normalize_images_u8")
    return out


def collate(samples):
    """Batch of reference views: stacks `imgs` / `imgs_u8` and `proj_mats`, and turns the per-sample depth ranges into the
    (B,1) tensors `CascadeMVSNet.forward` takes (what torch's default collate makes of dtu.py:177,190's FloatTensor([x]))."""
    out = {}
    for key in ("imgs", "imgs_u8", "proj_mats"):
        if key in samples[0]:
            out[key] = torch.stack([torch.as_tensor(s[key]) for s in samples])
    for key in ("init_depth_min", "depth_interval"):
        out[key] = torch.tensor([[float(torch.as_tensor(s[key]).reshape(-1)[0])] for s in samples], dtype=torch.float32)
    for key in samples[0]:
        if key not in out:
            out[key] = [s[key] for s in samples]
    return out


def configure_host_threads(intra_op_threads=1):
    """Call once in a process that feeds the engine from files.  torch's CPU operators (the `torch.stack` of `collate`, the pinned
    staging copy of DevicePrefetcher) run on an OpenMP pool with one thread per visible hardware thread, and after every parallel
    region those threads SPIN (libiomp's 200 ms block time) on the cores the decode threads need: measured on 8 cores
    (tools/cpu_loader_rate.py), ParallelLoader with 8 threads delivers 130 depth maps/s with torch's default pool and 435 with one
    intra-op thread - the sample function alone, without collate, reaches 460.  A GPU pod that sees 256 hardware threads
    but is granted 16 cores suffers most.  The host-side torch work of this pipeline is a few memcpy-sized copies; one thread
    is the right number.  Returns the previous setting."""
    prev = torch.get_num_threads()
    torch.set_num_threads(int(intra_op_threads))
    return prev


class ParallelLoader:
    """train.py:85-97 / eval.py:213 (`DataLoader(dataset, num_workers=4, pin_memory=True)`): an iterable of collated
    batches of `reader` whose samples are read by `num_workers` THREADS - a sample's cost is PNG / JPEG decoding and the
    resize in PIL and the PFM parsing in numpy, all of which release the GIL - with `prefetch_batches` batches in flight
    ahead of the consumer and delivery in index order (same batches as a sequential loop).  Feed it to DevicePrefetcher:
    decode (threads) -> pinned staging + H2D copy (side stream) -> engine, three stages overlapped.

    indices: the sample order (default: all samples in order; pass a permutation for training); a last partial batch is
    kept (drop_last=False like the reference's validation / test loaders).  processes=True: worker processes (see
    _iter_processes).  Call configure_host_threads() first: torch's spinning intra-op pool otherwise takes the decode threads'
    cores (3x fewer samples per second, measured)."""

    def __init__(self, reader, batch_size=1, num_workers=4, indices=None, prefetch_batches=4, drop_last=False, processes=False):
        self.reader, self.batch_size, self.num_workers = reader, int(batch_size), max(1, int(num_workers))
        self.indices = list(range(len(reader))) if indices is None else list(indices)
        self.prefetch = max(1, int(prefetch_batches))
        self.processes = bool(processes)
        if drop_last:
            self.indices = self.indices[:len(self.indices) // self.batch_size * self.batch_size]

    def __len__(self):
        return (len(self.indices) + self.batch_size - 1) // self.batch_size

    def _iter_processes(self):
        """Worker PROCESSES instead of threads (measured on the GPU box: the threaded form stops scaling at ~16 workers =
        1 100 images/s - the numpy / torch glue around PIL's decoder holds the GIL): torch's own DataLoader machinery, as the
        reference uses it (train.py:85-97), over the reader as a map-style dataset with this module's `collate`; whole
        batches are built in the workers and travel through shared memory, in order."""
        from torch.utils.data import DataLoader, Sampler

        class _Fixed(Sampler):
            def __init__(self, idx):
                self.idx = idx

            def __iter__(self):
                return iter(self.idx)

            def __len__(self):
                return len(self.idx)
        per_worker = max(2, (self.prefetch + self.num_workers - 1) // self.num_workers)
        return iter(DataLoader(self.reader, batch_size=self.batch_size, sampler=_Fixed(self.indices), num_workers=self.num_workers,
                               collate_fn=collate, drop_last=False, prefetch_factor=per_worker, persistent_workers=False))

    def __iter__(self):
        if self.processes:
            yield from self._iter_processes()
            return
        from concurrent.futures import ThreadPoolExecutor
        batches = [self.indices[i:i + self.batch_size] for i in range(0, len(self.indices), self.batch_size)]
        pool = ThreadPoolExecutor(max_workers=self.num_workers, thread_name_prefix="casmvs-loader")
        try:
            pending = []   # per batch: the futures of its samples, submitted in order
            nxt = 0
            while nxt < len(batches) or pending:
                while nxt < len(batches) and len(pending) < self.prefetch:
                    pending.append([pool.submit(self.reader.__getitem__, i) for i in batches[nxt]])
                    nxt += 1
                yield collate([f.result() for f in pending.pop(0)])
        finally:
            pool.shutdown(wait=False, cancel_futures=True)


class DevicePrefetcher:
    """Iterates batches (dicts from `collate`) and hands them over device-resident, with the host->device copies of the
    NEXT batch in flight on a side stream while the caller computes on the current one (`depth` batches ahead).
    Tensors are staged through pinned buffers (a pageable source would make the copy synchronous); `imgs_u8` batches are
    normalised on the device, still on the side stream, and delivered as `imgs`.  Non-tensor entries pass through."""

    def __init__(self, batches, device="cuda", depth=2, threaded=False):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("DevicePrefetcher feeds the MI355X engine; there is no CPU path")
        self.it = iter(batches)
        self.depth = max(1, depth)
        self.stream = torch.cuda.Stream(device=self.device)
        self.queue = []
        # threaded=True: a background thread pulls batches and stages them (pinned copy, H2D enqueue, normalisation launch) so
        # that the consumer's thread only launches the engine - at ~450 batches/s the 0.6 ms staging copy of a 6 MB batch is
        # a quarter of the consumer's time per batch.  (Added at the end of round 3 without a GPU run: opt-in.)
        self.threaded = bool(threaded)
        self._thread = None
        if self.threaded:
            import queue
            import threading
            self._q = queue.Queue(maxsize=self.depth)
            self._thread = threading.Thread(target=self._worker, name="casmvs-stager", daemon=True)
            self._thread.start()

    def _worker(self):
        try:
            torch.cuda.set_device(self.device)
            for batch in self.it:
                self._q.put(self._stage(batch))
            self._q.put(None)
        except BaseException as e:   # delivered to the consumer by __next__
            self._q.put(e)

    def _stage(self, batch):
        out, keep = {}, []
        with torch.cuda.stream(self.stream):
            for k, v in batch.items():
                if isinstance(v, torch.Tensor):
                    host = v if v.is_pinned() else v.contiguous().pin_memory()
                    keep.append(host)   # the pinned source must outlive the asynchronous copy
                    out[k] = host.to(self.device, non_blocking=True)
                else:
                    out[k] = v
            if "imgs_u8" in out:
                out["imgs"] = normalize_images_u8(out.pop("imgs_u8"))
            ev = torch.cuda.Event()
            ev.record(self.stream)
        return out, ev, keep

    def __iter__(self):
        return self

    def __next__(self):
        if self.threaded:
            item = self._q.get()
            if item is None:
                self._q.put(None)   # stay exhausted
                raise StopIteration
            if isinstance(item, BaseException):
                self._q.put(item)
                raise item
            out, ev, _keep = item
            torch.cuda.current_stream(self.device).wait_event(ev)
            for v in out.values():
                if isinstance(v, torch.Tensor):
                    v.record_stream(torch.cuda.current_stream(self.device))
            return out
        while len(self.queue) < self.depth:
            try:
                self.queue.append(self._stage(next(self.it)))
            except StopIteration:
                break
        if not self.queue:
            raise StopIteration
        out, ev, _keep = self.queue.pop(0)
        torch.cuda.current_stream(self.device).wait_event(ev)   # the compute stream waits, the host does not
        for v in out.values():
            if isinstance(v, torch.Tensor):
                v.record_stream(torch.cuda.current_stream(self.device))
        return out
