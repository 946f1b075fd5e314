"""Input side of the hot path: what the reference's datasets and eval loop do between the files and
`CascadeMVSNet.forward` - camera files -> per-level projection matrices (datasets/dtu.py:51-96), the relative matrices a
sample carries (dtu.py:181-186), image normalisation (dtu.py:134-137), batching of reference views with per-sample
depth ranges (train.py's DataLoader collate), PFM depth / confidence files (datasets/utils.py, eval.py:226-227) - plus
what the MI355X adds: the images travel as uint8 (4x fewer PCIe bytes) and are normalised by a HIP kernel, and a
double-buffered prefetcher overlaps the host->device copies of batch i+1 with the forward of batch i on a second stream.

Image decoding (PIL / cv2 in the reference) is not part of this module: it takes decoded uint8 arrays.
"""
import ctypes
import re

import numpy as np
import torch

from . import _lib

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
                                                    ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream))
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


class DevicePrefetcher:
    """Iterates batches (dicts from `collate`) and hands them over device-resident, with the host->device copies of the
    NEXT batch in flight on a side stream while the caller computes on the current one (`depth` batches ahead).
    Tensors are staged through pinned buffers (a pageable source would make the copy synchronous); `imgs_u8` batches are
    normalised on the device, still on the side stream, and delivered as `imgs`.  Non-tensor entries pass through."""

    def __init__(self, batches, device="cuda", depth=2):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("DevicePrefetcher feeds the MI355X engine; there is no CPU path")
        self.it = iter(batches)
        self.depth = max(1, depth)
        self.stream = torch.cuda.Stream(device=self.device)
        self.queue = []

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
