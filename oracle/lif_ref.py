"""TEST INFRASTRUCTURE -- ctypes view of oracle/liblif_ref.so (the plain-C LIF restatement)."""
import ctypes
import os

import numpy as np

_here = os.path.dirname(os.path.abspath(__file__))
_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(_here, "liblif_ref.so")
        if not os.path.exists(path):
            raise ImportError(f"{path} missing: run `make -C oracle` (or __graft_entry__.build())")
        _lib = ctypes.CDLL(path)
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def seq_fwd(x, v0=None, D=8, vth=1.0):
    x = np.ascontiguousarray(x, dtype=np.float32)
    T, n = x.shape[0], int(np.prod(x.shape[1:]))
    y = np.empty_like(x); vT = np.empty(x.shape[1:], np.float32)
    counts = np.empty(x.shape, np.uint8); inr = np.empty(x.shape, np.uint8)
    v0c = None if v0 is None else np.ascontiguousarray(v0, dtype=np.float32)
    lib().lif_ref_seq_fwd(_p(x), _p(v0c), _p(y), _p(vT), _p(counts), _p(inr), ctypes.c_int(T), ctypes.c_int64(n),
                          ctypes.c_float(vth), ctypes.c_int(D))
    return y, vT, counts, inr


def seq_bwd(gy, inrange, gvT=None, D=8, vth=1.0):
    gy = np.ascontiguousarray(gy, dtype=np.float32)
    T, n = gy.shape[0], int(np.prod(gy.shape[1:]))
    gx = np.empty_like(gy); gv0 = np.empty(gy.shape[1:], np.float32)
    g = None if gvT is None else np.ascontiguousarray(gvT, dtype=np.float32)
    lib().lif_ref_seq_bwd(_p(gy), _p(g), _p(np.ascontiguousarray(inrange)), _p(gx), _p(gv0), ctypes.c_int(T),
                          ctypes.c_int64(n), ctypes.c_float(vth), ctypes.c_int(D))
    return gx, gv0


def leaky_seq_fwd(x, v0=None, D=8, vth=1.0, tau=2.0, decay_input=True):
    x = np.ascontiguousarray(x, dtype=np.float32)
    T, n = x.shape[0], int(np.prod(x.shape[1:]))
    y = np.empty_like(x); vT = np.empty(x.shape[1:], np.float32)
    counts = np.empty(x.shape, np.uint8); inr = np.empty(x.shape, np.uint8)
    v0c = None if v0 is None else np.ascontiguousarray(v0, dtype=np.float32)
    lib().lif_ref_leaky_seq_fwd(_p(x), _p(v0c), _p(y), _p(vT), _p(counts), _p(inr), ctypes.c_int(T), ctypes.c_int64(n),
                                ctypes.c_float(vth), ctypes.c_int(D), ctypes.c_float(tau), ctypes.c_int(int(decay_input)))
    return y, vT, counts, inr


def leaky_seq_bwd(gy, inrange, gvT=None, D=8, vth=1.0, tau=2.0, decay_input=True):
    gy = np.ascontiguousarray(gy, dtype=np.float32)
    T, n = gy.shape[0], int(np.prod(gy.shape[1:]))
    gx = np.empty_like(gy); gv0 = np.empty(gy.shape[1:], np.float32)
    g = None if gvT is None else np.ascontiguousarray(gvT, dtype=np.float32)
    lib().lif_ref_leaky_seq_bwd(_p(gy), _p(g), _p(np.ascontiguousarray(inrange)), _p(gx), _p(gv0), ctypes.c_int(T),
                                ctypes.c_int64(n), ctypes.c_float(vth), ctypes.c_int(D), ctypes.c_float(tau),
                                ctypes.c_int(int(decay_input)))
    return gx, gv0
