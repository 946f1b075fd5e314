"""TEST INFRASTRUCTURE ONLY - import shim for the one kornia symbol the reference
uses (/root/reference/models/modules.py:6,66-67): kornia==0.2.0
`create_meshgrid(H, W, normalized_coordinates=False)` -> (1, H, W, 2) pixel grid with
[..., 0] = x in [0, W-1] and [..., 1] = y in [0, H-1].
"""
import torch


def create_meshgrid(height, width, normalized_coordinates=True, device=None):
    xs = torch.linspace(0, width - 1, width, device=device, dtype=torch.float32)
    ys = torch.linspace(0, height - 1, height, device=device, dtype=torch.float32)
    if normalized_coordinates:
        xs = (xs / (width - 1) - 0.5) * 2
        ys = (ys / (height - 1) - 0.5) * 2
    gy, gx = torch.meshgrid(ys, xs, indexing="ij")
    return torch.stack([gx, gy], dim=-1).unsqueeze(0)  # (1, H, W, 2)
