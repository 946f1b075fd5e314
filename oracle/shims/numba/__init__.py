"""TEST INFRASTRUCTURE ONLY - import shim for /root/reference/eval.py:16 (`from numba import jit`): `jit` is the identity, so
the decorated functions (eval.py:113-153) run as the plain numpy code they are written in.  numba's `fastmath` only frees
the ORDER of float32 sums; numpy / BLAS pick one.  Never imported by the product package."""


def jit(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return args[0]
    return lambda f: f


njit = jit
