"""casmvsnet_pl_amd - MI355X-native cascade-MVS depth engine.

The hot path of kwea123/CasMVSNet_pl's `CascadeMVSNet.forward` as hand-written gfx950 HIP
kernels behind a C ABI (include/casmvs.h, libcasmvs_hip.so), with a Python host side that
mirrors the reference's `models/mvsnet.py` / `models/modules.py` API.
"""
from .inplace_abn import ABN, InPlaceABN
from .modules import ConvBnReLU, ConvBnReLU3D, depth_regression, get_depth_values, homo_warp
from .mvsnet import CascadeMVSNet, CostRegNet, FeatureNet

__all__ = ["ABN", "InPlaceABN", "ConvBnReLU", "ConvBnReLU3D", "depth_regression", "get_depth_values",
           "homo_warp", "CascadeMVSNet", "CostRegNet", "FeatureNet"]
