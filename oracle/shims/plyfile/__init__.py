"""TEST INFRASTRUCTURE ONLY - import shim for /root/reference/eval.py:17 (`from plyfile import PlyData, PlyElement`), used
only by the script's __main__ block (never executed by the tests).  Never imported by the product package."""


class PlyElement:
    @staticmethod
    def describe(*args, **kwargs):
        raise NotImplementedError("plyfile shim: PLY writing is not part of the pinned path")


class PlyData:
    def __init__(self, *args, **kwargs):
        raise NotImplementedError("plyfile shim: PLY writing is not part of the pinned path")
