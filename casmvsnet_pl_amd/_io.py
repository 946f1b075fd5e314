"""ctypes binding of libcasmvs_io.so (include/casmvs_io.h): host-side PNG decoding for the input pipeline.

The library is plain C++ (its own inflate + PNG un-filtering, no zlib / libpng / torch / HIP) and thread-safe; ctypes releases the
GIL for the duration of a call, so `pipeline.ParallelLoader`'s threads decode in parallel.  `decode_png(data)` returns None for
PNG variants the library does not handle (interlaced, 1/2/4/16-bit samples): the caller keeps PIL for those, exactly what the
reference uses (datasets/dtu.py:168).  A missing library raises with the build command - it is not silently replaced."""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_int32, c_size_t, c_void_p

import numpy as np

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CASMVS_IO_LIB_PATH") or os.path.join(_PKG_DIR, "libcasmvs_io.so")
OK, UNSUPPORTED, CORRUPT, BAD_ARGUMENT = 0, 1, 2, 3

# every symbol include/casmvs_io.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "casmvs_io_last_error": (c_char_p, []),
    "casmvs_png_info": (c_int32, [c_char_p, c_size_t, POINTER(c_int32), POINTER(c_int32), POINTER(c_int32)]),
    "casmvs_png_decode": (c_int32, [c_char_p, c_size_t, c_void_p, c_size_t, c_int32]),
    "casmvs_png_decode_file": (c_int32, [c_char_p, c_void_p, c_size_t, c_int32, c_int32, c_int32]),
    "casmvs_png_decode_files": (c_int32, [POINTER(c_char_p), c_int32, c_void_p, c_size_t, c_size_t, c_int32, c_int32, c_int32, c_int32, POINTER(c_int32)]),
    "casmvs_zlib_inflate": (c_int32, [c_char_p, c_size_t, c_void_p, c_size_t, POINTER(c_size_t)]),
}
_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH) and "CASMVS_IO_LIB_PATH" not in os.environ:
            try:   # a two-second g++ build of one C++ file (no GPU, no dependencies); every host of this package has the compiler
                from .build import build_io_library
                build_io_library()
            except Exception as e:
                raise RuntimeError(f"{LIB_PATH} is missing and could not be built ({e}): run `python -m casmvsnet_pl_amd.build`") from e
        if not os.path.isfile(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: build it with `python -m casmvsnet_pl_amd.build` (g++, no GPU needed)")
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def last_error():
    return (load().casmvs_io_last_error() or b"").decode("utf-8", "replace")


def png_info(data):
    """(width, height, channels) of a PNG file image held in `data` (bytes); None if the library does not handle the variant."""
    w, h, c = c_int32(), c_int32(), c_int32()
    rc = load().casmvs_png_info(data, len(data), ctypes.byref(w), ctypes.byref(h), ctypes.byref(c))
    if rc == UNSUPPORTED:
        return None
    if rc != OK:
        raise ValueError(last_error())
    return w.value, h.value, c.value


def decode_png(data, channels=3, out=None):
    """PNG file image (bytes) -> (H, W, 3) uint8 [= np.asarray(Image.open(f).convert("RGB"))] or (H, W) [convert("L")].
    Returns None when the PNG variant is outside the library's set; raises ValueError for damaged files."""
    info = png_info(data)
    if info is None:
        return None
    w, h, _ = info
    shape = (h, w, 3) if channels == 3 else (h, w)
    if out is None:
        out = np.empty(shape, np.uint8)
    elif out.shape != shape or out.dtype != np.uint8 or not out.flags.c_contiguous:
        raise ValueError(f"decode_png: out must be a C-contiguous uint8 array of shape {shape}")
    rc = load().casmvs_png_decode(data, len(data), out.ctypes.data, out.strides[0], channels)
    if rc == UNSUPPORTED:
        return None
    if rc != OK:
        raise ValueError(last_error())
    return out


def decode_png_files(paths, width, height, channels=3, threads=0, out=None):
    """`len(paths)` PNG files of one size -> (N, H, W, 3) uint8 (or (N, H, W)) on native threads, one GIL-free call."""
    n = len(paths)
    shape = (n, height, width, 3) if channels == 3 else (n, height, width)
    if out is None:
        out = np.empty(shape, np.uint8)
    elif out.shape != shape or out.dtype != np.uint8 or not out.flags.c_contiguous:
        raise ValueError(f"decode_png_files: out must be a C-contiguous uint8 array of shape {shape}")
    arr = (c_char_p * n)(*[os.fsencode(p) for p in paths])
    status = (c_int32 * n)()
    rc = load().casmvs_png_decode_files(arr, n, out.ctypes.data, out.strides[0] if n else 0, out.strides[1] if n else 0, width, height, channels, threads, status)
    if rc != OK:
        err = ValueError(last_error())
        err.status = list(status)
        raise err
    return out


def zlib_inflate(data, capacity):
    """zlib stream -> bytes (tests: the library's inflate against python's zlib)."""
    buf = np.empty(max(capacity, 1), np.uint8)
    n = c_size_t()
    rc = load().casmvs_zlib_inflate(data, len(data), buf.ctypes.data, capacity, ctypes.byref(n))
    if rc != OK:
        raise ValueError(last_error())
    return buf[:n.value].tobytes()
