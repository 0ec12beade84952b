"""ctypes access to oracle/vq_ref.c (CPU ORACLE -- test infrastructure only)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libvq_ref.so")
_lib = None


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            subprocess.check_call(["bash", os.path.join(_HERE, "build.sh")])
        _lib = C.CDLL(_SO)
    return _lib


def _fp(a):
    return a.ctypes.data_as(C.c_void_p)


def prepare(codebook):
    w = np.ascontiguousarray(codebook, dtype=np.float32)
    V, E = w.shape
    en = np.empty_like(w)
    sq = np.empty(V, dtype=np.float32)
    _load().vq_ref_prepare(_fp(w), _fp(en), _fp(sq), C.c_int(V), C.c_int(E))
    return en, sq


def quantize(z, en, sq):
    """-> idx int64 [M], zn [M,E], dmin [M], gap [M] (second best - best distance)"""
    z = np.ascontiguousarray(z, dtype=np.float32)
    M, E = z.shape
    V = en.shape[0]
    idx = np.empty(M, dtype=np.int64)
    zn = np.empty_like(z)
    dmin = np.empty(M, dtype=np.float32)
    gap = np.empty(M, dtype=np.float32)
    _load().vq_ref_quantize(_fp(z), _fp(en), _fp(sq), C.c_int(M), C.c_int(V), C.c_int(E), _fp(idx), _fp(zn), _fp(dmin),
                            _fp(gap))
    return idx, zn, dmin, gap
