"""The three transforms the reference's datasets compose (dtu.py:134-141): ToTensor = uint8 HWC -> float32 CHW / 255,
Normalize = (x - mean[c]) / std[c] in float32 (torchvision.transforms.functional.to_tensor / normalize as published).
ColorJitter (blendedmvs.py:132, training split only) is random and NOT restated: it raises."""
import numpy as np
import torch


class Compose:
    def __init__(self, transforms):
        self.transforms = transforms

    def __call__(self, x):
        for t in self.transforms:
            x = t(x)
        return x


class ToTensor:
    def __call__(self, pic):
        a = np.asarray(pic)
        if a.ndim == 2:
            a = a[:, :, None]
        t = torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1)))
        return t.to(torch.float32).div(255) if t.dtype == torch.uint8 else t.to(torch.float32)


class Normalize:
    def __init__(self, mean, std):
        self.mean, self.std = mean, std

    def __call__(self, t):
        mean = torch.as_tensor(self.mean, dtype=t.dtype).view(-1, 1, 1)
        std = torch.as_tensor(self.std, dtype=t.dtype).view(-1, 1, 1)
        return (t - mean) / std          # torchvision: tensor.sub_(mean).div_(std)


class ColorJitter:
    def __init__(self, *args, **kwargs):
        pass

    def __call__(self, x):
        raise NotImplementedError("torchvision shim: ColorJitter is random; pin the val / test splits")


class ToPILImage:
    def __call__(self, x):
        raise NotImplementedError("torchvision shim: visualisation is out of scope")
