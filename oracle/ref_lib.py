"""ctypes loader for oracle/_ref/libref.so: the reference's vendored Eigen (BLAS + decompositions)
compiled from /root/reference/third_party/eigen by oracle/Makefile, called the way the reference's
solver core calls it (oracle/ref_driver.cc).  TEST INFRASTRUCTURE.  `available()` is False where the
library was never built (it is built in the container that has the reference tree and travels to
the GPU box as a file)."""

import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_ref", "libref.so")
_lib = None


def available():
    return os.path.exists(_LIB)


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(_LIB)
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _f(a):
    return np.asfortranarray(a, dtype=np.float64)


def dgemv(A, x, trans=False):
    """y = A x or A^T x through the reference's dgemv_ call (dense_matrix_impl.cc:55-67)."""
    A = _f(A)
    x = np.ascontiguousarray(x, dtype=np.float64)
    m, n = A.shape
    y = np.empty(n if trans else m)
    lib().ref_dgemv(ctypes.c_char(b"T" if trans else b"N"), ctypes.c_int(m), ctypes.c_int(n), _p(A), _p(x), _p(y))
    return y


def dgemm(A, B, ta=False, tb=False):
    """op(A) op(B) through the reference's dgemm_ call (linear_map_multiply.cc:14-37)."""
    A, B = _f(A), _f(B)
    m = A.shape[1] if ta else A.shape[0]
    k = A.shape[0] if ta else A.shape[1]
    n = B.shape[0] if tb else B.shape[1]
    C = np.empty((m, n), order="F")
    lib().ref_dgemm(ctypes.c_char(b"T" if ta else b"N"), ctypes.c_char(b"T" if tb else b"N"), ctypes.c_int(m),
                    ctypes.c_int(n), ctypes.c_int(k), _p(A), _p(B), _p(C))
    return C


def ldlt_inverse(A):
    """Eigen::LDLT + solve(Identity) (dense_matrix_impl.cc:21-30)."""
    A = _f(A)
    n = A.shape[0]
    out = np.empty((n, n), order="F")
    rc = lib().ref_ldlt_inverse(ctypes.c_int(n), _p(A), _p(out))
    if rc != 0:
        raise ArithmeticError("Eigen::LDLT did not succeed")
    return out


def llt_solve(A, b):
    A = _f(A)
    b = np.ascontiguousarray(b, dtype=np.float64)
    x = np.empty_like(b)
    if lib().ref_llt_solve(ctypes.c_int(A.shape[0]), _p(A), _p(b), _p(x)) != 0:
        raise ArithmeticError("Eigen::LLT did not succeed")
    return x


def gram_svd(Y):
    """(d, V, U) as prox/ortho_invariant.cc:36-50 computes them."""
    Y = _f(Y)
    m, n = Y.shape
    d, V, U = np.empty(n), np.empty((n, n), order="F"), np.empty((m, n), order="F")
    if lib().ref_gram_svd(ctypes.c_int(m), ctypes.c_int(n), _p(Y), _p(d), _p(V), _p(U)) != 0:
        raise ArithmeticError("Eigen::SelfAdjointEigenSolver did not succeed")
    return d, V, U


def lasso_sweeps(A, Minv, b, lam, state, k):
    """k sweeps of the compiled lasso on `state` (c_oracle.LassoState) with the mat-vecs through the
    reference tree's dgemv_ (ref_driver.cc: ref_lasso_sweeps); no stopping test."""
    m, n = A.shape
    assert A.flags.f_contiguous and Minv.flags.f_contiguous and A.dtype == np.float64
    b = np.ascontiguousarray(b, dtype=np.float64)
    t, w, g = np.empty(m), np.empty(m), np.empty(n)
    lib().ref_lasso_sweeps(ctypes.c_int(m), ctypes.c_int(n), _p(A), _p(Minv), _p(b), ctypes.c_double(lam),
                           _p(state.x0), _p(state.x1), _p(state.u), _p(state.y0), _p(state.y1),
                           ctypes.c_int(k), _p(t), _p(w), _p(g))
    state.iter += k
    return k
