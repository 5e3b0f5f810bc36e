"""ctypes wrapper of oracle/liboracle.so (plain-C CPU oracle; test infrastructure only)."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: run `make -C {_HERE}`")
        _lib = ctypes.CDLL(path)
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def log_qz(z, mu, logvar, log_norm, prod_marginals=True):
    z, mu, logvar = (np.ascontiguousarray(a, dtype=np.float32) for a in (z, mu, logvar))
    M, D = z.shape
    N = mu.shape[0]
    lq = np.empty(M, dtype=np.float64)
    pm = np.empty(M, dtype=np.float64) if prod_marginals else None
    f = lib().oracle_log_qz
    f.restype = None
    f.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_double,
                                          ctypes.c_void_p, ctypes.c_void_p]
    f(_p(z), _p(mu), _p(logvar), M, N, D, float(log_norm), _p(lq), _p(pm))
    return lq, pm


def log_qz_excl(z, mu, logvar, log_norm, diag_offset=0):
    """logsumexp over the posteriors j != i + diag_offset (mi_upper_bound's masked logsumexp)."""
    z, mu, logvar = (np.ascontiguousarray(a, dtype=np.float32) for a in (z, mu, logvar))
    M, D = z.shape
    N = mu.shape[0]
    out = np.empty(M, dtype=np.float64)
    f = lib().oracle_log_qz_excl
    f.restype = None
    f.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_double,
                                          ctypes.c_int64, ctypes.c_void_p]
    f(_p(z), _p(mu), _p(logvar), M, N, D, float(log_norm), int(diag_offset), _p(out))
    return out


def reparam_iw(mu, logvar, eps, log_p_x_z=None):
    mu, logvar, eps = (np.ascontiguousarray(a, dtype=np.float32) for a in (mu, logvar, eps))
    B, K, D = eps.shape
    lpxz = None if log_p_x_z is None else np.ascontiguousarray(log_p_x_z, dtype=np.float32)
    z = np.empty((B, K, D), dtype=np.float32)
    lq, lp, iw = np.empty((B, K)), np.empty((B, K)), np.empty(B)
    f = lib().oracle_reparam_iw
    f.restype = None
    f.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int64, ctypes.c_int64, ctypes.c_int] + [ctypes.c_void_p] * 4
    f(_p(mu), _p(logvar), _p(eps), _p(lpxz), B, K, D, _p(z), _p(lq), _p(lp), _p(iw))
    return z, lq, lp, iw
