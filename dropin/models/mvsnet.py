from casmvsnet_pl_amd.modules import *  # noqa: F401,F403  (the reference does `from .modules import *`)
from casmvsnet_pl_amd.mvsnet import CascadeMVSNet, CostRegNet, FeatureNet  # noqa: F401
