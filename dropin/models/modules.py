from casmvsnet_pl_amd.inplace_abn import InPlaceABN  # noqa: F401
from casmvsnet_pl_amd.modules import (ConvBnReLU, ConvBnReLU3D, depth_regression,  # noqa: F401
                                      get_depth_values, homo_warp)
