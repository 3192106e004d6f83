"""ctypes loader for oracle/_build/liboracle.so (plain-C restatements; test infrastructure)."""

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            subprocess.check_call(["make", "-s", "-C", _HERE])
        _lib = ctypes.CDLL(_LIB)
        _lib.lasso_admm_run.restype = ctypes.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


class LassoState(object):
    def __init__(self, n):
        self.x0, self.x1, self.u, self.y0, self.y1 = (np.zeros(n) for _ in range(5))
        self.iter = 0
        self.resid = np.zeros(4)
        self.optimal = False


def lasso_run(A, Minv, b, lam, state, k, abs_tol=1e-4, rel_tol=1e-2, epoch=10):
    """k more sweeps of the compiled lasso (A col-major fp64); returns sweeps executed."""
    m, n = A.shape
    assert A.flags.f_contiguous and Minv.flags.f_contiguous
    opt = ctypes.c_int(0)
    done = lib().lasso_admm_run(
        ctypes.c_int(m), ctypes.c_int(n), _p(A), _p(Minv), _p(b), ctypes.c_double(lam),
        _p(state.x0), _p(state.x1), _p(state.u), _p(state.y0), _p(state.y1),
        ctypes.c_int(state.iter), ctypes.c_int(k), ctypes.c_double(abs_tol),
        ctypes.c_double(rel_tol), ctypes.c_int(epoch), _p(state.resid), ctypes.byref(opt))
    state.optimal = bool(opt.value)
    state.iter += done - (1 if state.optimal else 0)
    return done


def set_threads(t):
    """Threads of the mat-vecs in lasso_run (1 = the reference's configuration)."""
    lib().oracle_set_threads(ctypes.c_int(int(t)))


def max_threads():
    return int(lib().oracle_max_threads())


def gram(A):
    """A A^T by the plain-C blocked contraction (Init timing sample of bench.py)."""
    m, n = A.shape
    assert A.flags.f_contiguous
    G = np.empty((m, m), order="F")
    lib().gram_aat(ctypes.c_int(m), ctypes.c_int(n), _p(A), _p(G))
    return G


def tv1d(y, lam):
    y = np.ascontiguousarray(y, dtype=np.float64)
    out = np.empty_like(y)
    lib().tv1d_prox(ctypes.c_int(y.size), _p(y), ctypes.c_double(lam), _p(out))
    return out
