"""TEST INFRASTRUCTURE ONLY - import shim for `from torchvision import transforms as T` in the reference's dataset classes
(/root/reference/datasets/dtu.py:8,134-141 etc.); torchvision is not installed offline.  Never imported by the product."""
from . import transforms  # noqa: F401
