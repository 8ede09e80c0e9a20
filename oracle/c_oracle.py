"""ctypes front-end of oracle/selscan_ref.c (TEST INFRASTRUCTURE ONLY; double-precision truth)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libmlagg_oracle.so")
_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _f(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def selscan_fwd(u, delta, A, B, C, D=None, delta_bias=None, softplus=True):
    u, delta, A, B, C, D, delta_bias = map(_f, (u, delta, A, B, C, D, delta_bias))
    b, d, L = u.shape
    out = np.empty_like(u)
    rc = lib().oracle_selscan_fwd(_p(u), _p(delta), _p(A), _p(B), _p(C), _p(D), _p(delta_bias), _p(out),
                                  b, d, L, A.shape[1], B.shape[1], int(softplus))
    assert rc == 0
    return out


def selscan_bwd(u, delta, A, B, C, D, delta_bias, dout, softplus=True):
    u, delta, A, B, C, D, delta_bias, dout = map(_f, (u, delta, A, B, C, D, delta_bias, dout))
    b, d, L = u.shape
    du, ddelta = np.empty_like(u), np.empty_like(u)
    dA, dB, dC = np.empty_like(A), np.empty_like(B), np.empty_like(C)
    dD, dbias = np.empty(d, np.float32), np.empty(d, np.float32)
    rc = lib().oracle_selscan_bwd(_p(u), _p(delta), _p(A), _p(B), _p(C), _p(D), _p(delta_bias), _p(dout),
                                  _p(du), _p(ddelta), _p(dA), _p(dB), _p(dC), _p(dD), _p(dbias),
                                  b, d, L, A.shape[1], B.shape[1], int(softplus))
    assert rc == 0
    return du, ddelta, dA, dB, dC, dD, dbias
