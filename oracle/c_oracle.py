"""TEST INFRASTRUCTURE ONLY -- ctypes binding of oracle/vq_argmin.c."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle_vq.so")


def build():
    subprocess.run(["make", "-s", "-C", _HERE], check=True)
    return _SO


def _lib():
    if not os.path.exists(_SO):
        build()
    lib = ctypes.CDLL(_SO)
    lib.oracle_vq_argmin.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64,
                                     ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]
    lib.oracle_vq_argmin.restype = None
    lib.oracle_vq_argmax_cos.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64,
                                         ctypes.c_int64, ctypes.c_void_p]
    lib.oracle_vq_argmax_cos.restype = None
    lib.oracle_vq_argmin_cdist.argtypes = lib.oracle_vq_argmax_cos.argtypes
    lib.oracle_vq_argmin_cdist.restype = None
    return lib


def vq_argmin(x: np.ndarray, codebook: np.ndarray, return_dist=False):
    x = np.ascontiguousarray(x, dtype=np.float32)
    e = np.ascontiguousarray(codebook, dtype=np.float32)
    n, dim = x.shape
    assert e.shape[1] == dim
    ids = np.empty(n, np.int64)
    dist = np.empty(n, np.float32)
    _lib().oracle_vq_argmin(x.ctypes.data, e.ctypes.data, n, e.shape[0], dim, ids.ctypes.data,
                            dist.ctypes.data)
    return (ids, dist) if return_dist else ids


def vq_argmax_cos(x: np.ndarray, embed: np.ndarray):
    x = np.ascontiguousarray(x, dtype=np.float32)
    e = np.ascontiguousarray(embed, dtype=np.float32)
    n, dim = x.shape
    assert e.shape[1] == dim
    ids = np.empty(n, np.int64)
    _lib().oracle_vq_argmax_cos(x.ctypes.data, e.ctypes.data, n, e.shape[0], dim, ids.ctypes.data)
    return ids


def vq_argmin_cdist(x: np.ndarray, embed: np.ndarray):
    x = np.ascontiguousarray(x, dtype=np.float32)
    e = np.ascontiguousarray(embed, dtype=np.float32)
    n, dim = x.shape
    ids = np.empty(n, np.int64)
    _lib().oracle_vq_argmin_cdist(x.ctypes.data, e.ctypes.data, n, e.shape[0], dim, ids.ctypes.data)
    return ids
