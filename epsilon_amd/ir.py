"""Prox-affine IR constructors (py3) for the solver boundary.

The reference frontend (python-2 + cvxpy 0.3.6, not runnable here) emits the IR
through python/epopt/expression.py, linear_map.py and constant.py.  This module
restates only the constructors whose output the *solver* consumes, with the same
shapes, keys and byte packing, so tests / bench can hand the C-ABI exactly the
bytes the untouched frontend would:

  * constant.store            -> reference python/epopt/constant.py:7-41
  * linear-map atoms          -> reference python/epopt/linear_map.py:22-121
  * expression constructors   -> reference python/epopt/expression.py:149-160,217-287,383-395,427-433
"""

import hashlib

import numpy as np
import scipy.sparse as sp

from . import wire
from .wire import Cone, Constant, Expression, LinearMap, ProxFunction, Size, Variable


class Expr(object):
    """Expression proto + the {location: bytes} data it references.

    Mirrors reference python/epopt/expression.py:46-97 (proto with reference
    semantics for args plus a data dict).
    """

    def __init__(self, proto, data=None):
        self.proto = proto
        self.data = dict(data or {})

    @property
    def size(self):
        return tuple(self.proto.size.dim)

    def dim(self):
        d = self.proto.size.dim
        return d[0] * d[1]


class LMap(object):
    """LinearMap proto + data (reference python/epopt/linear_map.py:11-20)."""

    def __init__(self, proto, data=None):
        self.proto = proto
        self.data = dict(data or {})

    @property
    def m(self):
        return self.proto.m

    @property
    def n(self):
        return self.proto.n


# ---- constants (reference python/epopt/constant.py) --------------------------------------


def value_location(value_bytes):
    # reference uses "/mem/data/" + str(abs(hash(bytes))) (constant.py:7-8); python-3
    # string hashing is salted per process, so a content hash keeps keys stable.
    return "/mem/data/" + hashlib.sha1(value_bytes).hexdigest()[:16]


def value_data(value):
    """reference constant.py:10-34: dense -> float64 column-major bytes;
    sparse -> CSC indptr|indices|data as int32,int32,float64."""
    if isinstance(value, np.ndarray):
        value = np.asarray(value, dtype=np.float64)
        c = Constant(
            constant_type=Constant.DENSE_MATRIX,
            m=value.shape[0],
            n=1 if value.ndim == 1 else value.shape[1])
        value_bytes = value.tobytes(order="F")
    elif sp.issparse(value):
        csc = sp.csc_matrix(value)
        csc.sort_indices()
        c = Constant(
            constant_type=Constant.SPARSE_MATRIX,
            m=value.shape[0], n=value.shape[1], nnz=csc.nnz)
        value_bytes = (csc.indptr.astype(np.int32).tobytes() +
                       csc.indices.astype(np.int32).tobytes() +
                       csc.data.astype(np.float64).tobytes())
    else:
        raise ValueError("unknown value type " + str(type(value)))
    return c, value_bytes


def store(value, data):
    c, value_bytes = value_data(value)
    loc = value_location(value_bytes)
    data[loc] = value_bytes
    c.data_location = loc
    return c


def store_device(ptr, m, n, dtype, data, key):
    """Device-resident dense constant: the data map entry is a (ptr, m*n, dtype)
    tuple instead of host bytes; `_solve` turns it into an `eps_blob` with
    kind EPS_BLOB_DEVICE_F32/F64 (see include/epsilon_hip.h).  Extension of the
    reference data map for inputs that already live in HBM."""
    loc = "/dev/data/" + key
    data[loc] = ("device", int(ptr), int(m) * int(n), dtype)
    return Constant(constant_type=Constant.DENSE_MATRIX, m=m, n=n, data_location=loc)


# ---- linear maps (reference python/epopt/linear_map.py) ---------------------------------


def scalar(alpha, n):
    return LMap(LinearMap(linear_map_type=LinearMap.SCALAR, m=n, n=n, scalar=float(alpha)))


def identity(n):
    return scalar(1, n)


def negate_map(n):
    return scalar(-1, n)


def dense_matrix(value=None, constant=None, data=None):
    data = dict(data or {})
    if constant is None:
        constant = store(np.asarray(value, dtype=np.float64).reshape(
            (value.shape[0], -1)), data)
    return LMap(LinearMap(linear_map_type=LinearMap.DENSE_MATRIX,
                          m=constant.m, n=constant.n, constant=constant), data)


def sparse_matrix(value):
    data = {}
    c = store(value, data)
    return LMap(LinearMap(linear_map_type=LinearMap.SPARSE_MATRIX,
                          m=c.m, n=c.n, constant=c), data)


def diagonal_matrix(value):
    data = {}
    value = np.asarray(value, dtype=np.float64).reshape(-1, 1)
    c = store(value, data)
    n = c.m * c.n
    return LMap(LinearMap(linear_map_type=LinearMap.DIAGONAL_MATRIX,
                          m=n, n=n, constant=c), data)


def transpose(A):
    return LMap(LinearMap(linear_map_type=LinearMap.TRANSPOSE,
                          m=A.n, n=A.m, arg=[A.proto]), A.data)


def kronecker_product(A, B):
    # reference linear_map.py:23-41 (including its degenerate-case folding)
    if A.m * A.n == 1:
        return B
    if B.m * B.n == 1:
        return A
    if (A.proto.linear_map_type == LinearMap.SCALAR and
            B.proto.linear_map_type == LinearMap.SCALAR):
        return scalar(A.proto.scalar * B.proto.scalar, A.n * B.n)
    data = dict(A.data)
    data.update(B.data)
    return LMap(LinearMap(linear_map_type=LinearMap.KRONECKER_PRODUCT,
                          m=A.m * B.m, n=A.n * B.n,
                          arg=[A.proto, B.proto]), data)


def left_matrix_product(A, n):  # X -> A X, X has n columns (linear_map.py:112-113)
    return kronecker_product(identity(n), A)


def right_matrix_product(B, m):  # X -> X B, X has m rows (linear_map.py:115-116)
    return kronecker_product(transpose(B), identity(m))


# ---- expressions (reference python/epopt/expression.py) ---------------------------------


def variable(m, n, variable_id):
    return Expr(Expression(expression_type=Expression.VARIABLE,
                           size=Size(dim=[m, n]),
                           variable=Variable(variable_id=variable_id)))


def constant(value):
    value = np.asarray(value, dtype=np.float64)
    if value.ndim == 1:
        value = value.reshape(-1, 1)
    data = {}
    c = store(value, data)
    return Expr(Expression(expression_type=Expression.CONSTANT,
                           size=Size(dim=[value.shape[0], value.shape[1]]),
                           constant=c), data)


def scalar_constant(value, size=(1, 1)):
    return Expr(Expression(expression_type=Expression.CONSTANT,
                           size=Size(dim=list(size)),
                           constant=Constant(constant_type=Constant.SCALAR,
                                             scalar=float(value))))


def parameter(m, n, parameter_id):
    """reference expression.py:227-236: a CONSTANT whose value is bound per call
    through `_solve.solve(..., parameters, ...)`."""
    return Expr(Expression(expression_type=Expression.CONSTANT,
                           size=Size(dim=[m, n]),
                           constant=Constant(constant_type=Constant.DENSE_MATRIX,
                                             parameter_id=parameter_id, m=m, n=n)))


def _merge(args):
    data = {}
    for a in args:
        data.update(a.data)
    return data


def add(*args):
    size = args[0].proto.size
    return Expr(Expression(expression_type=Expression.ADD, size=size,
                           arg=[a.proto for a in args]), _merge(args))


def reshape(arg, m, n):
    return Expr(Expression(expression_type=Expression.RESHAPE,
                           size=Size(dim=[m, n]), arg=[arg.proto]), arg.data)


def linear_map(A, x):
    if A.n != x.dim():
        raise ValueError("linear map %dx%d applied to expression of dim %d" %
                         (A.m, A.n, x.dim()))
    data = dict(A.data)
    data.update(x.data)
    return Expr(Expression(expression_type=Expression.LINEAR_MAP,
                           size=Size(dim=[A.m, 1]),
                           linear_map=A.proto, arg=[x.proto]), data)


def negate(x):
    return linear_map(negate_map(x.dim()), x)


def zero(x):
    """indicator of {x == 0}; reference expression.py:282-287 with Cone.ZERO."""
    return Expr(Expression(expression_type=Expression.INDICATOR,
                           size=Size(dim=[1, 1]),
                           cone=Cone(cone_type=Cone.ZERO),
                           arg=[x.proto]), x.data)


def prox_function(f, *args, **kwargs):
    data = _merge(args)
    data.update(kwargs.get("data", {}))
    return Expr(Expression(expression_type=Expression.PROX_FUNCTION,
                           size=Size(dim=list(kwargs.get("size", (1, 1)))),
                           prox_function=f,
                           arg=[a.proto for a in args]), data)


def prox(type_, *args, **kwargs):
    """Convenience: ProxFunction(type, alpha, arg_size=[size of each arg], ...)."""
    alpha = kwargs.pop("alpha", 1.0)
    epigraph = kwargs.pop("epigraph", False)
    extra = {}
    for k in ("scaled_zone_params", "sum_largest_params", "has_axis", "axis"):
        if k in kwargs:
            extra[k] = kwargs.pop(k)
    arg_size = kwargs.pop("arg_size", None)
    if arg_size is None:
        arg_size = [list(a.size) for a in args]
    f = ProxFunction(prox_function_type=type_, alpha=float(alpha), epigraph=epigraph,
                     arg_size=[Size(dim=list(s)) for s in arg_size], **extra)
    return prox_function(f, *args, **kwargs)


class Problem(object):
    """reference python/epopt/expression.py:22-43."""

    def __init__(self, objective_terms, constraints=()):
        self.terms = list(objective_terms)
        self.constraints = list(constraints)

    def proto(self):
        obj = Expression(expression_type=Expression.ADD, size=Size(dim=[1, 1]),
                         arg=[t.proto for t in self.terms])
        return wire.Problem(objective=obj,
                            constraint=[c.proto for c in self.constraints])

    def SerializeToString(self):
        return self.proto().SerializeToString()

    def expression_data(self):
        data = {}
        for e in self.terms + self.constraints:
            data.update(e.data)
        return data


def get_variables(proto):
    """{variable_id: (m, n)} in lexicographic id order; reference
    src/epsilon/expression/expression_util.cc:11-31."""
    out = {}

    def walk(e):
        if e.expression_type == Expression.VARIABLE:
            out[e.variable.variable_id] = tuple(e.size.dim)
        for a in e.arg:
            walk(a)
        if e.expression_type == Expression.PROX_FUNCTION and e.prox_function is not None:
            pass
    if isinstance(proto, wire.Problem):
        walk(proto.objective)
        for c in proto.constraint:
            walk(c)
    else:
        walk(proto)
    return dict(sorted(out.items()))
