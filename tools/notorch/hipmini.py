"""The dozen HIP runtime calls a measurement script needs, through ctypes - no torch.  A python + numpy process starts in a
fraction of a second; `import torch` on a fresh GPU box pages in for 1-2 minutes, which is most of what a short gpurun call is
charged for.  tools/notorch/step_runner.py drives the whole forward of the engine through the C ABI on top of this."""
import ctypes

import numpy as np

import os

_hip = ctypes.CDLL("libamdhip64.so")
H2D, D2H = 1, 2
FAKE = os.environ.get("HIPMINI_FAKE") == "1"   # no GPU: "device" arrays live in host memory (checks a script up to its first kernel launch)


def _check(rc, what):
    if rc != 0:
        _hip.hipGetErrorString.restype = ctypes.c_char_p
        raise RuntimeError(f"{what}: {_hip.hipGetErrorString(rc).decode()}")


class DeviceArray:
    """A device allocation with a shape (float32 unless dtype is given)."""

    def __init__(self, shape, dtype=np.float32):
        self.shape = tuple(int(s) for s in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        if FAKE:
            self._host = np.zeros(max(self.nbytes, 16), np.uint8)
            self.ptr = self._host.ctypes.data
            return
        p = ctypes.c_void_p()
        _check(_hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(max(self.nbytes, 16))), "hipMalloc")
        self.ptr = p.value

    @classmethod
    def from_numpy(cls, a):
        a = np.ascontiguousarray(a)
        d = cls(a.shape, a.dtype)
        if FAKE:
            d._host[:a.nbytes] = a.reshape(-1).view(np.uint8)
            return d
        _check(_hip.hipMemcpy(ctypes.c_void_p(d.ptr), a.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(a.nbytes), H2D), "hipMemcpy H2D")
        return d

    def numpy(self):
        out = np.empty(self.shape, self.dtype)
        _check(_hip.hipMemcpy(out.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(self.ptr), ctypes.c_size_t(self.nbytes), D2H), "hipMemcpy D2H")
        return out

    def fill_bytes(self, value=0, stream=None):
        _check(_hip.hipMemsetAsync(ctypes.c_void_p(self.ptr), int(value), ctypes.c_size_t(self.nbytes), ctypes.c_void_p(stream or 0)), "hipMemsetAsync")

    def free(self):
        if self.ptr:
            _hip.hipFree(ctypes.c_void_p(self.ptr))
            self.ptr = None

    @property
    def p(self):
        return ctypes.c_void_p(self.ptr)


def synchronize():
    _check(_hip.hipDeviceSynchronize(), "hipDeviceSynchronize")


def stream_create():
    if FAKE:
        return 0
    s = ctypes.c_void_p()
    _check(_hip.hipStreamCreate(ctypes.byref(s)), "hipStreamCreate")
    return s.value


class Event:
    def __init__(self):
        e = ctypes.c_void_p()
        if not FAKE:
            _check(_hip.hipEventCreate(ctypes.byref(e)), "hipEventCreate")
        self.e = e.value

    def record(self, stream=None):
        _check(_hip.hipEventRecord(ctypes.c_void_p(self.e), ctypes.c_void_p(stream or 0)), "hipEventRecord")

    def synchronize(self):
        _check(_hip.hipEventSynchronize(ctypes.c_void_p(self.e)), "hipEventSynchronize")

    def ms_since(self, start):
        ms = ctypes.c_float()
        _check(_hip.hipEventElapsedTime(ctypes.byref(ms), ctypes.c_void_p(start.e), ctypes.c_void_p(self.e)), "hipEventElapsedTime")
        return ms.value


def device_name():
    class Props(ctypes.Structure):
        _fields_ = [("name", ctypes.c_char * 256), ("pad", ctypes.c_char * 4096)]
    p = Props()
    fn = getattr(_hip, "hipGetDevicePropertiesR0600", None) or _hip.hipGetDeviceProperties
    fn(ctypes.byref(p), 0)
    return p.name.decode(errors="replace")
