"""CPU oracle: a numpy (fp64) restatement of Epsilon's solver core.

TEST INFRASTRUCTURE ONLY.  Nothing under `epsilon_amd/` may import this file; it is
used by `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg as
the *checker* of the HIP path, never as the thing shipped or measured.

Each function cites the reference file:line it follows (paths relative to
/root/reference/src/epsilon unless they start with python/).  Pinning status:

  * linear-map algebra, Kronecker apply, BlockCholesky, ComputeFill -> pinned by the
    reference's own gtest expectations, restated in tests/test_oracle_reference_tests.py
    (linear/linear_map_test.cc:67-229, linear/kronecker_product_impl_test.cc:9-20,
     linear/dense_matrix_impl_test.cc:24-29, vector/block_cholesky_test.cc:23-104).
  * prox operators -> the reference only checks them against CVXPY at 1e-2
    (python/epopt/prox_test.py:250-266); restated against scipy minimisation of
    lam*f(x) + 0.5||x - v||^2 in tests/test_oracle_prox.py.
  * ADMM drivers -> the reference holds no fixtures for them (no C++ test exists,
    python tests need cvxpy 0.3.6/python 2).  PARITY UNPINNED against the reference
    binary: the reference C++ cannot be built here without writing stand-ins for
    glog / generated protobuf headers, which is not allowed.  The drivers are pinned
    instead the way python/epopt/solve_test.py:26-83 does it: objective at termination
    within (1+1e-2)*opt + 1e-4 of an independent high-accuracy solve.
  * TOTAL_VARIATION_1D -> the reference calls glmgen `tf_dp` (third_party/glmgen is an
    empty, un-pinned submodule: .gitmodules:4-6).  Restated here from Johnson's
    published dynamic program; parity unpinned, certified by the KKT conditions.
"""

import math

import numpy as np
import scipy.sparse as sp

from epsilon_amd import wire
from epsilon_amd.wire import Expression, ProxFunction
from epsilon_amd.wire import LinearMap as LinearMapProto

DENSE, SPARSE, DIAGONAL, SCALAR, KRONECKER = range(5)  # linear/linear_map.h:18-26
TYPE_NAMES = ["DENSE", "SPARSE", "DIAGONAL", "SCALAR", "KRONECKER"]


class CheckError(Exception):
    """Stands for a glog CHECK / LOG(FATAL) in the reference (-> _solve.error)."""


def check(cond, msg="CHECK failed"):
    if not cond:
        raise CheckError(msg)


# =============================================================================================
# typed linear maps  (linear/*.{h,cc})
# =============================================================================================


class LM(object):
    """LinearMapImpl restated (linear/linear_map.h:33-57)."""

    def __init__(self, type_, **kw):
        self.type = type_
        self.__dict__.update(kw)

    # ---- constructors -------------------------------------------------------------------
    @staticmethod
    def dense(A):
        A = np.array(A, dtype=np.float64)
        check(A.ndim == 2)
        return LM(DENSE, A=A)

    @staticmethod
    def sparse(A):
        return LM(SPARSE, A=sp.csc_matrix(A, dtype=np.float64))

    @staticmethod
    def diagonal(d):
        return LM(DIAGONAL, d=np.array(d, dtype=np.float64).reshape(-1))

    @staticmethod
    def scalar(alpha, n):  # linear/scalar_matrix_impl.h:12-13
        return LM(SCALAR, alpha=float(alpha), n_=int(n))

    @staticmethod
    def identity(n):  # linear/linear_map.cc:106-108
        return LM.scalar(1.0, n)

    @staticmethod
    def kron(A, B):
        return LM(KRONECKER, KA=A, KB=B)

    # ---- shape -----------------------------------------------------------------------------
    @property
    def m(self):
        if self.type in (DENSE, SPARSE):
            return self.A.shape[0]
        if self.type == DIAGONAL:
            return self.d.shape[0]
        if self.type == SCALAR:
            return self.n_
        return self.KA.m * self.KB.m  # kronecker_product_impl.h:18

    @property
    def n(self):
        if self.type in (DENSE, SPARSE):
            return self.A.shape[1]
        if self.type == DIAGONAL:
            return self.d.shape[0]
        if self.type == SCALAR:
            return self.n_
        return self.KA.n * self.KB.n

    def as_dense(self):
        if self.type == DENSE:
            return self.A
        if self.type == SPARSE:
            return self.A.toarray()
        if self.type == DIAGONAL:
            return np.diag(self.d)
        if self.type == SCALAR:
            return self.alpha * np.eye(self.n_)
        return np.kron(self.KA.as_dense(), self.KB.as_dense())  # kronecker_product_impl.cc:7-22

    def as_sparse(self):
        if self.type == SPARSE:
            return self.A
        if self.type == SCALAR:
            return sp.identity(self.n_, format="csc") * self.alpha
        if self.type == DIAGONAL:
            return sp.diags(self.d, format="csc")
        return sp.csc_matrix(self.as_dense())

    def T(self):
        if self.type == DENSE:  # dense_matrix_impl.h:41-43 (flag flip, shared data)
            return LM.dense(self.A.T)
        if self.type == SPARSE:
            return LM.sparse(self.A.T)
        if self.type == DIAGONAL:
            return LM.diagonal(self.d)
        if self.type == SCALAR:
            return LM.scalar(self.alpha, self.n_)
        return LM.kron(self.KA.T(), self.KB.T())  # kronecker_product_impl.h:28-30

    def inverse(self):
        if self.type == DENSE:  # dense_matrix_impl.cc:21-30: LDLT, assumes symmetric
            check(self.m == self.n)
            return LM.dense(np.linalg.solve(self.A, np.eye(self.n)))
        if self.type == SPARSE:  # sparse_matrix_impl.cc:60-78 densifies
            return LM.dense(np.linalg.inv(self.A.toarray()))
        if self.type == DIAGONAL:  # diagonal_matrix_impl.cc:14-22: 1/0 -> 0
            inv = np.zeros_like(self.d)
            nz = self.d != 0
            inv[nz] = 1.0 / self.d[nz]
            return LM.diagonal(inv)
        if self.type == SCALAR:  # scalar_matrix_impl.h:30-32
            with np.errstate(divide="ignore"):
                return LM.scalar(np.float64(1.0) / np.float64(self.alpha), self.n_)
        return LM.kron(self.KA.inverse(), self.KB.inverse())  # kronecker_product_impl.h:32-34

    def apply(self, x):
        x = np.asarray(x, dtype=np.float64)
        if self.type == DENSE:  # dense_matrix_impl.cc:55-67 dgemv_
            return self.A @ x
        if self.type == SPARSE:
            return self.A @ x
        if self.type == DIAGONAL:  # diagonal_matrix_impl.h:23
            return self.d * x
        if self.type == SCALAR:  # scalar_matrix_impl.h:24
            return self.alpha * x
        # kronecker_product_impl.cc:45-58: vec(B X A^T), X is B.n x A.n column-major
        X = x.reshape((self.KB.n, self.KA.n), order="F")
        BX = self.KB.as_dense() @ X if self.KB.type != SCALAR else self.KB.alpha * X
        Y = BX @ self.KA.as_dense().T if self.KA.type != SCALAR else self.KA.alpha * BX
        return Y.reshape(-1, order="F")

    def equals(self, o):
        if self.type != o.type or self.m != o.m or self.n != o.n:
            return False
        if self.type == DENSE:
            return np.array_equal(self.A, o.A)
        if self.type == SPARSE:
            return (self.A != o.A).nnz == 0
        if self.type == DIAGONAL:
            return np.array_equal(self.d, o.d)
        if self.type == SCALAR:
            return self.alpha == o.alpha
        return self.KA.equals(o.KA) and self.KB.equals(o.KB)

    def __repr__(self):
        return "LM(%s %dx%d)" % (TYPE_NAMES[self.type], self.m, self.n)


def lm_multiply(L, R):
    """kMultiplyTable, linear/linear_map_multiply.cc:14-311 (result *type* matters)."""
    check(L.n == R.m, "multiply shape mismatch %r * %r" % (L, R))
    a, b = L.type, R.type
    if a == SCALAR and b == SCALAR:
        return LM.scalar(L.alpha * R.alpha, L.n_)
    if a == SCALAR and b == KRONECKER:  # :188-198 stays Kronecker
        return LM.kron(lm_multiply(LM.scalar(L.alpha, R.KA.m), R.KA),
                       lm_multiply(LM.scalar(1.0, R.KB.m), R.KB))
    if a == KRONECKER and b == SCALAR:
        # :223-227 calls the Scalar*Kron routine directly with swapped arguments: the scalar
        # is re-sized from the Kronecker factors, so non-square Kronecker maps work too
        return LM.kron(lm_multiply(LM.scalar(R.alpha, L.KA.m), L.KA),
                       lm_multiply(LM.scalar(1.0, L.KB.m), L.KB))
    if a == KRONECKER and b == KRONECKER:  # :230-241
        if L.KA.n == R.KA.m and L.KB.n == R.KB.m:
            return LM.kron(lm_multiply(L.KA, R.KA), lm_multiply(L.KB, R.KB))
        return LM.sparse(L.as_sparse() @ R.as_sparse())
    if a == SCALAR:
        if b == DENSE:
            return LM.dense(L.alpha * R.A)
        if b == SPARSE:
            return LM.sparse(L.alpha * R.A)
        if b == DIAGONAL:
            return LM.diagonal(L.alpha * R.d)
    if b == SCALAR:
        if a == DENSE:
            return LM.dense(L.A * R.alpha)
        if a == SPARSE:
            return LM.sparse(L.A * R.alpha)
        if a == DIAGONAL:
            return LM.diagonal(L.d * R.alpha)
    if a == DIAGONAL and b == DIAGONAL:
        return LM.diagonal(L.d * R.d)
    if a == DENSE or b == DENSE:
        # Dense x {Dense,Sparse,Diag,Kron} and {Sparse,Diag,Kron} x Dense -> Dense
        if a == DIAGONAL:
            return LM.dense(L.d[:, None] * R.A)
        if b == DIAGONAL:
            return LM.dense(L.A * R.d[None, :])
        return LM.dense(np.asarray(L.as_dense() @ R.as_dense()))
    # remaining: Sparse/Diag/Kron mixtures -> Sparse
    return LM.sparse(L.as_sparse() @ R.as_sparse())


def lm_add(L, R):
    """kAddTable, linear/linear_map_add.cc:13-293."""
    a, b = L.type, R.type
    if a == SCALAR and b == SCALAR:
        return LM.scalar(L.alpha + R.alpha, L.n_)
    if a == KRONECKER and b == SCALAR:
        return lm_add(R, L)
    if a == SCALAR and b == KRONECKER:  # :167-187
        K = R
        if K.KA.type == SCALAR:
            s1 = LM.scalar(0.0, K.KA.n)
            s2 = LM.scalar(L.alpha / K.KA.alpha, K.KB.n)
            return LM.kron(lm_add(s1, K.KA), lm_add(s2, K.KB))
        if K.KB.type == SCALAR:
            s1 = LM.scalar(L.alpha / K.KB.alpha, K.KA.n)
            s2 = LM.scalar(0.0, K.KB.n)
            return LM.kron(lm_add(s1, K.KA), lm_add(s2, K.KB))
        return LM.sparse(L.as_sparse() + K.as_sparse())
    if a == KRONECKER and b == KRONECKER:  # :213-226
        if L.KA.equals(R.KA):
            return LM.kron(L.KA, lm_add(L.KB, R.KB))
        if L.KB.equals(R.KB):
            return LM.kron(lm_add(L.KA, R.KA), L.KB)
        return LM.sparse(L.as_sparse() + R.as_sparse())
    if a == DENSE or b == DENSE:
        return LM.dense(L.as_dense() + R.as_dense())
    if a == DIAGONAL and b == DIAGONAL:
        return LM.diagonal(L.d + R.d)
    if (a == DIAGONAL and b == SCALAR):
        return LM.diagonal(L.d + R.alpha)
    if (a == SCALAR and b == DIAGONAL):
        return LM.diagonal(R.d + L.alpha)
    return LM.sparse(L.as_sparse() + R.as_sparse())


def lm_scale(alpha, A):  # linear/linear_map.cc:33-35
    return lm_multiply(LM.scalar(alpha, A.m), A)


def compute_type(a, b):  # linear/linear_map.cc:141-149 (op type is ignored there too)
    if a <= SCALAR and b <= SCALAR:
        return min(a, b)
    return DENSE


def nonzeros(t, m, n):  # linear/linear_map.cc:151-164
    if t in (DENSE, SPARSE):
        return m * n
    if t == DIAGONAL:
        check(m == n)
        return n
    if t == SCALAR:
        return 1
    raise CheckError("Not implemented")


def get_scalar(A):  # linear/linear_map.cc:131-139
    check(A.type == SCALAR, "Non-scalar matrix")
    return A.alpha


def get_diagonal(A):  # linear/linear_map.cc:118-129
    if A.type == SCALAR:
        return np.full(A.n_, A.alpha)
    check(A.type == DIAGONAL, "Non-diagonal linear map")
    return A.d


# ---- proto -> linear map (linear/linear_map.cc:39-104), data decoders (vector/vector_util.cc) ---


PARAMS_KEY = "\0parameters"  # {parameter_id: Constant} bound for this call (solver.cc:109-116)


def resolve_constant(c, data):
    """A Constant carrying a parameter_id stands for the value bound to that CVXPY Parameter
    (reference algorithms/solver.cc:30-61,109-116 overwrites the proto in place)."""
    if c.parameter_id:
        bound = data.get(PARAMS_KEY, {})
        check(c.parameter_id in bound, "parameter %s has no value" % c.parameter_id)
        return bound[c.parameter_id]
    return c


def build_matrix(c, data):  # vector/vector_util.cc:247-259
    c = resolve_constant(c, data)
    check(c.constant_type == wire.Constant.DENSE_MATRIX)
    check(c.data_location in data, "missing data " + c.data_location)
    buf = data[c.data_location]
    check(len(buf) == c.m * c.n * 8, "dense blob size mismatch")
    return np.frombuffer(buf, dtype=np.float64).reshape((c.m, c.n), order="F")


def build_sparse_matrix(c, data):  # vector/vector_util.cc:261-281
    check(c.constant_type == wire.Constant.SPARSE_MATRIX)
    buf = data[c.data_location]
    m, n, nnz = c.m, c.n, c.nnz
    check(len(buf) == nnz * 8 + (n + nnz + 1) * 4, "sparse blob size mismatch")
    colptr = np.frombuffer(buf, dtype=np.int32, count=n + 1)
    rowidx = np.frombuffer(buf, dtype=np.int32, count=nnz, offset=4 * (n + 1))
    vals = np.frombuffer(buf, dtype=np.float64, count=nnz, offset=4 * (n + 1 + nnz))
    return sp.csc_matrix((vals, rowidx, colptr), shape=(m, n))


def build_linear_map(p, data):
    t = p.linear_map_type
    if t == LinearMapProto.DENSE_MATRIX:
        return LM.dense(build_matrix(p.constant, data))
    if t == LinearMapProto.SPARSE_MATRIX:
        return LM.sparse(build_sparse_matrix(p.constant, data))
    if t == LinearMapProto.DIAGONAL_MATRIX:
        return LM.diagonal(build_matrix(p.constant, data).reshape(-1, order="F"))
    if t == LinearMapProto.SCALAR:
        return LM.scalar(p.scalar, p.n)
    if t == LinearMapProto.KRONECKER_PRODUCT:
        check(len(p.arg) == 2)
        return LM.kron(build_linear_map(p.arg[0], data), build_linear_map(p.arg[1], data))
    if t == LinearMapProto.TRANSPOSE:
        check(len(p.arg) == 1)
        return build_linear_map(p.arg[0], data).T()
    raise CheckError("No linear map function for %d" % t)


# =============================================================================================
# block containers (vector/block_vector.cc, vector/block_matrix.cc)
# =============================================================================================


class BlockVector(object):
    """std::map<string, VectorXd> with lexicographic iteration (block_vector.h:13-75)."""

    def __init__(self, data=None):
        self.d = {}
        if data:
            for k, v in data.items():
                self.d[k] = np.array(v, dtype=np.float64).reshape(-1)

    def keys(self):
        return sorted(self.d)

    def items(self):
        return [(k, self.d[k]) for k in sorted(self.d)]

    def has_key(self, k):
        return k in self.d

    def __call__(self, k):
        check(k in self.d, k + " not in BlockVector")
        return self.d[k]

    def set(self, k, v):
        self.d[k] = np.array(v, dtype=np.float64).reshape(-1)

    def copy(self):
        return BlockVector(self.d)

    def insert_or_add(self, k, v):  # block_vector.cc:44-49
        if k in self.d:
            self.d[k] = self.d[k] + v
        else:
            self.d[k] = np.array(v, dtype=np.float64)

    def iadd(self, o):  # :9-13
        for k, v in o.items():
            self.insert_or_add(k, v)
        return self

    def isub(self, o):  # :15-19
        for k, v in o.items():
            self.insert_or_add(k, -v)
        return self

    def __add__(self, o):
        return self.copy().iadd(o)

    def __sub__(self, o):
        return self.copy().isub(o)

    def scaled(self, alpha):  # :21-26
        return BlockVector({k: alpha * v for k, v in self.d.items()})

    def select(self, keys):  # :69-77
        return BlockVector({k: self.d[k] for k in keys if k in self.d})

    def norm(self):  # :87-93
        return math.sqrt(sum(float(v @ v) for _, v in self.items()))

    def n(self):
        return sum(v.shape[0] for v in self.d.values())


class BlockMatrix(object):
    """map<col, map<row, LinearMap>> (block_matrix.h:33-72)."""

    def __init__(self):
        self.d = {}

    def copy(self):
        c = BlockMatrix()
        c.d = {col: dict(rows) for col, rows in self.d.items()}
        return c

    def col_keys(self):
        return sorted(self.d)

    def row_keys(self):
        return sorted({r for rows in self.d.values() for r in rows})

    def col(self, c):
        check(c in self.d)
        return [(r, self.d[c][r]) for r in sorted(self.d[c])]

    def has_key(self, r, c):
        return c in self.d and r in self.d[c]

    def get(self, r, c):
        check(c in self.d, "column: %s not found" % c)
        check(r in self.d[c], "row: %s not found" % r)
        return self.d[c][r]

    def set(self, r, c, v):
        self.d.setdefault(c, {})[r] = v

    def insert_or_add(self, r, c, v):  # block_matrix.cc:170-176
        rows = self.d.setdefault(c, {})
        if r in rows:
            rows[r] = lm_add(rows[r], v)
        else:
            rows[r] = v

    def remove(self, r, c):  # :241-250
        check(c in self.d and r in self.d[c])
        del self.d[c][r]
        if not self.d[c]:
            del self.d[c]

    def entries(self):
        for c in sorted(self.d):
            for r in sorted(self.d[c]):
                yield r, c, self.d[c][r]

    def T(self):  # :55-64
        t = BlockMatrix()
        for r, c, v in self.entries():
            t.insert_or_add(c, r, v.T())
        return t

    def m(self):  # :178-192
        seen, m = set(), 0
        for r, c, v in self.entries():
            if r not in seen:
                seen.add(r)
                m += v.m
        return m

    def n(self):  # :194-200
        return sum(self.d[c][sorted(self.d[c])[0]].n for c in self.d)

    def inverse(self):  # :9-27,66-74 (block diagonal only)
        check(self.m() == self.n(), "Inverting non square matrix")
        seen = set()
        for c in self.col_keys():
            check(len(self.d[c]) == 1, "Unable to invert matrix")
            r = next(iter(self.d[c]))
            check(r not in seen, "Unable to invert matrix")
            seen.add(r)
        inv = BlockMatrix()
        for c in self.col_keys():
            r = next(iter(self.d[c]))
            inv.insert_or_add(c, r, self.d[c][r].inverse())
        return inv

    def left_identity(self):  # :76-88
        C = BlockMatrix()
        for r, c, v in self.entries():
            if r not in C.d:
                C.insert_or_add(r, r, LM.identity(v.m))
        return C

    def right_identity(self):  # :90-100
        C = BlockMatrix()
        for c in self.col_keys():
            r = sorted(self.d[c])[0]
            C.insert_or_add(c, c, LM.identity(self.d[c][r].n))
        return C

    def __matmul__(self, B):  # operator*(BlockMatrix, BlockMatrix) :102-127
        C = BlockMatrix()
        for bc in B.col_keys():
            for br in sorted(B.d[bc]):
                if br not in self.d:
                    continue
                for ar in sorted(self.d[br]):
                    C.insert_or_add(ar, bc, lm_multiply(self.d[br][ar], B.d[bc][br]))
        return C

    def __add__(self, B):  # :129-137
        C = self.copy()
        for r, c, v in B.entries():
            C.insert_or_add(r, c, v)
        return C

    def scaled(self, alpha):  # :143-151
        C = BlockMatrix()
        for r, c, v in self.entries():
            C.insert_or_add(r, c, lm_scale(alpha, v))
        return C

    def __sub__(self, B):  # :139-141
        return self + B.scaled(-1.0)

    def apply(self, x):  # operator*(BlockMatrix, BlockVector) :155-168
        y = BlockVector()
        for k, xv in x.items():
            if k not in self.d:
                continue
            for r in sorted(self.d[k]):
                y.insert_or_add(r, self.d[k][r].apply(xv))
        return y

    def as_dense(self, row_keys=None, col_keys=None):
        rk = row_keys or self.row_keys()
        ck = col_keys or self.col_keys()
        rdim, cdim = {}, {}
        for r, c, v in self.entries():
            rdim[r] = v.m
            cdim[c] = v.n
        ro = np.cumsum([0] + [rdim[r] for r in rk])
        co = np.cumsum([0] + [cdim[c] for c in ck])
        D = np.zeros((ro[-1], co[-1]))
        for r, c, v in self.entries():
            i, j = rk.index(r), ck.index(c)
            D[ro[i]:ro[i + 1], co[j]:co[j + 1]] = v.as_dense()
        return D


# =============================================================================================
# block LDL^T with greedy min-fill ordering (vector/block_cholesky.cc)
# =============================================================================================

FILL_MAX = (1 << 64) - 1


def compute_fill(A, k):  # block_cholesky.cc:11-48
    keys, has_diag = [], False
    for r, _ in A.col(k):
        if r == k:
            has_diag = True
        else:
            keys.append(r)
    if not has_diag:
        return FILL_MAX
    fill = 0
    for i in keys:
        aik = compute_type(A.get(i, k).type, A.get(k, k).type)
        for j in keys:
            t = compute_type(aik, A.get(j, k).type)
            fill += nonzeros(t, A.get(i, k).m, A.get(j, k).m)
    return fill


def next_key(A):  # :51-64 (first strict minimum in lexicographic key order)
    best_key, best_fill = None, FILL_MAX
    for key in A.col_keys():
        f = compute_fill(A, key)
        if f < best_fill:
            best_key, best_fill = key, f
    check(best_fill != FILL_MAX, "no eliminable key")
    return best_key


def remove_key(A, key):  # :68-83
    V = BlockMatrix()
    to_remove = []
    for r, v in A.col(key):
        to_remove.append((r, key))
        if r != key:
            to_remove.append((key, r))
            V.set(r, key, v)
    for r, c in to_remove:
        A.remove(r, c)
    return V


def forward_sub(L, keys, b):  # :86-100
    b = b.copy()
    for jx, j in enumerate(keys):
        if b.has_key(j):
            neg_bj = -1.0 * b(j)
            for i in keys[jx + 1:]:
                if L.has_key(i, j):
                    b.insert_or_add(i, L.get(i, j).apply(neg_bj))
    return b


def back_sub(LT, keys, b):  # :103-117 (iterates keys in reverse)
    b = b.copy()
    rk = keys[::-1]
    for jx, j in enumerate(rk):
        if b.has_key(j):
            neg_bj = -1.0 * b(j)
            for i in rk[jx + 1:]:
                if LT.has_key(i, j):
                    b.insert_or_add(i, LT.get(i, j).apply(neg_bj))
    return b


class BlockCholesky(object):
    def __init__(self):
        self.p = []
        self.L = BlockMatrix()
        self.D_inv = BlockMatrix()
        self.LT = None

    def compute(self, A):  # :119-133
        A = A.copy()
        n_cols = len(A.col_keys())
        for _ in range(n_cols):
            key = next_key(A)
            Di_inv = BlockMatrix()
            Di_inv.set(key, key, A.get(key, key).inverse())
            V = remove_key(A, key)
            self.L = self.L + (V @ Di_inv)
            self.D_inv = self.D_inv + Di_inv
            A = A - (V @ Di_inv @ V.T())
            self.p.append(key)
        self.LT = self.L.T()
        return self

    def solve(self, b):  # :135-137
        return back_sub(self.LT, self.p, self.D_inv.apply(forward_sub(self.L, self.p, b)))


# =============================================================================================
# affine builder (affine/affine.cc) and IR helpers (expression/expression_util.cc)
# =============================================================================================


def constraint_key(i):  # affine.cc:131-136
    return "constraint:%d" % i


def arg_key(i):  # affine.cc:138-140
    return "arg:%d" % i


def get_dimension(e):  # expression_util.cc:50-53
    check(len(e.size.dim) == 2, "expression size must have 2 dims")
    return e.size.dim[0] * e.size.dim[1]


def get_variables(e, out=None):  # expression_util.cc:11-31 (set ordered by variable id)
    if out is None:
        out = {}
    if isinstance(e, wire.Problem):
        get_variables(e.objective, out)
        for c in e.constraint:
            get_variables(c, out)
        return dict(sorted(out.items()))
    if e.expression_type == Expression.VARIABLE:
        out.setdefault(e.variable.variable_id, e)
    for a in e.arg:
        get_variables(a, out)
    return dict(sorted(out.items()))


class AffineOperator(object):
    def __init__(self):
        self.A = BlockMatrix()
        self.b = BlockVector()


def build_affine_operator(e, data, row_key, A, b, L=None):  # affine.cc:22-129
    if L is None:
        L = LM.identity(get_dimension(e))
    t = e.expression_type
    if t in (Expression.ADD, Expression.RESHAPE):  # :30-39, :97 (RESHAPE is a no-op)
        for a in e.arg:
            build_affine_operator(a, data, row_key, A, b, L)
    elif t == Expression.VARIABLE:  # :41-49
        A.insert_or_add(row_key, e.variable.variable_id, L)
    elif t == Expression.CONSTANT:  # :51-69
        c = resolve_constant(e.constant, data)
        if c.data_location == "":
            b_dense = np.full(L.n, c.scalar)
        else:
            b_dense = build_matrix(c, data).reshape(-1, order="F")
        if b is not None:
            b.insert_or_add(row_key, L.apply(b_dense))
    elif t == Expression.LINEAR_MAP:  # :71-84
        check(len(e.arg) == 1)
        build_affine_operator(e.arg[0], data, row_key, A, b,
                              lm_multiply(L, build_linear_map(e.linear_map, data)))
    else:
        raise CheckError("No linear function for expression type %d" % t)


# =============================================================================================
# proximal operators (prox/*.cc)
# =============================================================================================


class ProxArg(object):  # prox/prox.h:11-35
    def __init__(self, f, data, H, A):
        self.f, self.data, self.H, self.A = f, data, H, A


def _bm_get_scalar(A):  # prox/vector_prox.cc:4-26
    alpha, first = None, True
    for c in A.col_keys():
        rows = A.d[c]
        if len(rows) != 1 or c != next(iter(rows)):
            return None
        Ai = rows[c]
        if Ai.type != SCALAR:
            return None
        if first:
            alpha, first = Ai.alpha, False
        elif alpha != Ai.alpha:
            return None
    return alpha


def _bm_get_diagonal(A):  # prox/vector_prox.cc:28-49
    alpha, first = None, True
    for c in A.col_keys():
        rows = A.d[c]
        if len(rows) != 1 or c != next(iter(rows)):
            return None
        Ai = rows[c]
        if Ai.type not in (SCALAR, DIAGONAL):
            return None
        ai = get_diagonal(Ai)
        if first:
            alpha, first = ai.copy(), False
        elif not np.array_equal(alpha, ai):
            return None
    return alpha


class VectorProx(object):
    """prox/vector_prox.cc:51-183: reduce a prox with scalar/diagonal H, A to a plain
    vector prox on v' = B v + g; x = C (x' - g) + D v."""

    def init(self, arg):
        if not self._init_scalar(arg) and not self._init_diagonal(arg):
            raise CheckError("Affine transformation is not scalar or diagonal")
        self.g = arg.H.b
        self.f = arg.f

    def _init_scalar(self, arg):  # :51-70
        alpha = arg.f.alpha
        H, A = arg.H.A, arg.A.A
        HT, AT = H.T(), A.T()
        beta = _bm_get_scalar(HT @ H)
        gamma = _bm_get_scalar(H @ AT @ A @ HT)
        if beta is None or gamma is None:
            return False
        self.B = (H @ AT).scaled(beta / gamma)
        self.C = HT.scaled(1.0 / beta)
        self.D = BlockMatrix()
        self.lam = alpha * beta * beta / gamma
        self.lam_vec = np.full(A.n(), self.lam)
        self.elementwise = False
        check(self.lam >= 0)
        return True

    def _init_diagonal(self, arg):  # :72-118
        alpha = arg.f.alpha
        H, A = arg.H.A, arg.A.A
        HT, AT = H.T(), A.T()
        beta = _bm_get_diagonal(HT @ H)
        gamma = _bm_get_diagonal(H @ AT @ A @ HT)
        if beta is None or gamma is None:
            return False
        n = beta.shape[0]
        lam = np.zeros(n)
        delta = np.zeros(n)
        beta = beta.copy()
        gamma = gamma.copy()
        for i in range(n):
            if gamma[i]:
                lam[i] = alpha * beta[i] * beta[i] / gamma[i]
            else:
                lam[i] = 0
                beta[i] = 1
                gamma[i] = 1
                delta[i] = 1
        B0, C0, D0 = LM.diagonal(beta / gamma), LM.diagonal(1.0 / beta), LM.diagonal(delta)
        Bs, Cs, Ds = BlockMatrix(), BlockMatrix(), BlockMatrix()
        for key in H.col_keys():
            Bs.set(key, key, B0)
            Cs.set(key, key, C0)
            Ds.set(key, key, D0)
        self.B = H @ Bs @ AT
        self.C = Cs @ HT
        self.D = (AT @ A).inverse() @ Ds @ AT
        self.lam_vec = lam
        self.lam = None
        self.elementwise = True
        return True

    def apply(self, v):  # :147-183
        vin = self.B.apply(v) + self.g
        f = self.f
        x = BlockVector()
        if f.has_axis:
            n = len(f.arg_size)
            V = [vin(arg_key(i)).reshape(tuple(f.arg_size[i].dim), order="F") for i in range(n)]
            X = [np.zeros(tuple(f.arg_size[i].dim)) for i in range(n)]
            k = f.arg_size[0].dim[1 - f.axis]
            for it in range(k):
                if f.axis == 0:
                    ins = [Vi[:, it] for Vi in V]
                else:
                    ins = [Vi[it, :] for Vi in V]
                outs = self.apply_vector(ins)
                for i, o in enumerate(outs):
                    if f.axis == 0:
                        X[i][:, it] = o
                    else:
                        X[i][it, :] = o
            for i in range(n):
                x.set(arg_key(i), X[i].reshape(-1, order="F"))
        else:
            nargs = len([k for k in vin.keys() if k.startswith("arg:")])
            ins = [vin(arg_key(i)) for i in range(nargs)]
            outs = self.apply_vector(ins)
            for i, o in enumerate(outs):
                x.set(arg_key(i), o)
        return self.C.apply(x - self.g) + self.D.apply(v)

    def lam_scalar(self):
        check(not self.elementwise)
        return self.lam

    def lam_vec_for(self, v):
        """lambda_vec() (vector_prox.cc:189-191) at the length of the argument it scales: the
        reference returns the whole vector whatever slice is being processed; scalar lambda
        promoted to the slice length is the well-defined case."""
        if self.elementwise:
            check(self.lam_vec.shape[0] == v.shape[0], "elementwise lambda with an axis")
            return self.lam_vec
        return np.full(v.shape[0], self.lam)

    def apply_vector(self, ins):
        raise NotImplementedError


def scaled_zone_params(f, data):  # prox/scaled_zone.cc:34-76
    if f.has_axis:
        n = f.arg_size[0].dim[f.axis]
    else:
        n = f.arg_size[0].dim[0] * f.arg_size[0].dim[1]
    t = f.prox_function_type
    one, zero = np.ones(n), np.zeros(n)
    if t == ProxFunction.NORM_1:
        return one, one, 0.0, 0.0
    if t == ProxFunction.SUM_DEADZONE:
        return one, one, f.scaled_zone_params.m, 0.0
    if t == ProxFunction.SUM_HINGE:
        return one, zero, 0.0, 0.0
    if t == ProxFunction.SUM_QUANTILE:
        tmp = BlockVector()
        build_affine_operator(f.scaled_zone_params.alpha_expr, data, "alpha", None, tmp)
        build_affine_operator(f.scaled_zone_params.beta_expr, data, "beta", None, tmp)

        def promote(x):  # :26-32
            if x.shape[0] == n:
                return x
            check(x.shape[0] == 1 and n != 1)
            return np.full(n, x[0])
        return promote(tmp("alpha")), promote(tmp("beta")), 0.0, 0.0
    raise CheckError("Unknown prox type")


def apply_scaled_zone(alpha, beta, M, C, lam, v):  # prox/scaled_zone.cc:78-104
    x = v - C
    if np.ndim(lam):
        lam = lam[:x.shape[0]]  # the reference loop indexes lambda(i), i < v.rows()
    up = x > M + lam * alpha
    dn = x < -M - lam * beta
    inside = np.abs(x) <= M
    out = np.where(inside, x,
                   np.where(up, x - lam * alpha,
                            np.where(dn, x + lam * beta,
                                     np.where(x > 0, M, -M))))
    return out


class ScaledZoneProx(VectorProx):  # prox/scaled_zone.cc:106-121
    def init(self, arg):
        VectorProx.init(self, arg)
        self.alpha, self.beta, self.M, self.Cc = scaled_zone_params(arg.f, arg.data)

    def apply_vector(self, ins):
        return [apply_scaled_zone(self.alpha, self.beta, self.M, self.Cc, self.lam_vec, ins[0])]


def largest_real_cubic_root(b, c, d):
    """prox/newton.cc:293-323: Durand-Kerner on x^3 + b x^2 + c x + d, largest real root."""
    eps = 1e-12
    p = complex(0.4, 0.9)
    q, r = p * p, p * p * p
    for _ in range(100):
        fp = p * p * p + b * p * p + c * p + d
        fq = q * q * q + b * q * q + c * q + d
        fr = r * r * r + b * r * r + c * r + d
        if abs(fp) < eps and abs(fq) < eps and abs(fr) < eps:
            break
        p, q, r = (p - fp / ((p - q) * (p - r)), q - fq / ((q - p) * (q - r)),
                   r - fr / ((r - p) * (r - q)))
    m = -1e41
    for z in (p, q, r):
        if abs(z.imag) < eps and z.real > m:
            m = z.real
    return m


class SumSquareEpigraph(VectorProx):  # prox/sum_square.cc:42-57
    def apply_vector(self, ins):
        u, s = ins[0], float(ins[1][0])
        lam = largest_real_cubic_root(1 + s, 0.25 + s, (s - float(u @ u)) / 4)
        if lam < 0:
            lam = 0.0
        return [u / (1 + 2 * lam), np.array([s + lam])]


class ScaledZoneEpigraph(VectorProx):
    """prox/scaled_zone.cc:123-279: projection of (v, s) onto {(x, t): f(x) <= t}.  The
    reference finds the multiplier lam with a randomised 3-way-partition selection
    (`random()`, :198); the multiplier itself is the unique root of
        sum_i w_i^2 max(k_i - lam, 0) = s + lam ,   k_i = (|y_i| - M) / w_i ,  w = alpha or beta
    which is computed here by sorting (any exact method gives the same number)."""

    def init(self, arg):
        VectorProx.init(self, arg)
        self.alpha, self.beta, self.M, self.Cc = scaled_zone_params(arg.f, arg.data)

    def apply_vector(self, ins):
        v, s = ins[0], float(ins[1][0])
        y = v - self.Cc
        n = y.shape[0]
        a, b, M = self.alpha[:n], self.beta[:n], self.M
        w = np.where(y > 0, a, b)
        act = (np.abs(y) > M) & (w != 0)
        fval = float(np.sum(w[act] * (np.abs(y[act]) - M)))
        if fval <= s:
            return [v.copy(), np.array([s])]
        k = (np.abs(y[act]) - M) / w[act]
        w2 = w[act] ** 2
        order = np.argsort(-k)
        k, w2 = k[order], w2[order]
        acc = np.cumsum(w2 * k) - s
        div = np.cumsum(w2) + 1.0
        lam_c = acc / div  # candidate with the j+1 largest keys active
        nxt = np.append(k[1:], -np.inf)
        j = np.nonzero((lam_c < k) & (lam_c >= nxt))[0][0]
        lam = float(lam_c[j])
        x = apply_scaled_zone(self.alpha, self.beta, M, self.Cc, np.full(n, lam), v)
        return [x, np.array([s + lam])]


class Norm2Prox(VectorProx):  # prox/norm_2.cc:4-19
    def apply_vector(self, ins):
        lam = self.lam_scalar()
        v = ins[0]
        nv = math.sqrt(float(v @ v))
        if nv >= lam:
            return [(1 - lam / nv) * v]
        return [np.zeros_like(v)]


class NonNegativeProx(VectorProx):  # prox/non_negative.cc:3-11
    def apply_vector(self, ins):
        return [np.maximum(ins[0], 0.0)]


def tv1d_prox(y, lam):
    """Exact minimiser of 0.5||x-y||^2 + lam*sum|x[i+1]-x[i]|.

    The reference calls glmgen's `tf_dp` (prox/total_variation_1d.cc:8,21), whose source is
    absent (empty submodule).  This restates N. Johnson's published dynamic program
    ("A dynamic programming algorithm for the fused lasso and L0-segmentation", JCGS 2013):
    delta_k(b) = d/db of the optimal cost of the first k points with x_k = b is piecewise
    linear and increasing; delta_{k+1}(b) = (b - y_{k+1}) + clip(delta_k(b), -lam, lam); the
    clip points (tm_k, tp_k) are the back-pointers x_k = clip(x_{k+1}, tm_k, tp_k).
    """
    y = np.asarray(y, dtype=np.float64)
    n = y.shape[0]
    if n == 0:
        return y.copy()
    if n == 1 or lam == 0:
        return y.copy()
    # knots of the piecewise-linear derivative kept in a double-ended array
    x = np.zeros(2 * n)
    a = np.zeros(2 * n)
    b = np.zeros(2 * n)
    tm = np.zeros(n - 1)
    tp = np.zeros(n - 1)
    tm[0] = -lam + y[0]
    tp[0] = lam + y[0]
    lo_i, hi_i = n - 1, n
    x[lo_i], x[hi_i] = tm[0], tp[0]
    a[lo_i], b[lo_i] = 1.0, -y[0] + lam
    a[hi_i], b[hi_i] = -1.0, y[0] + lam
    afirst, bfirst = 1.0, -lam - y[1]
    alast, blast = -1.0, -lam + y[1]
    for k in range(1, n - 1):
        alo, blo = afirst, bfirst
        lo = lo_i
        while lo <= hi_i:
            if alo * x[lo] + blo > -lam:
                break
            alo += a[lo]
            blo += b[lo]
            lo += 1
        tm[k] = (-lam - blo) / alo
        lo_i = lo - 1
        x[lo_i] = tm[k]
        ahi, bhi = alast, blast
        hi = hi_i
        while hi >= lo_i:
            if -ahi * x[hi] - bhi < lam:
                break
            ahi += a[hi]
            bhi += b[hi]
            hi -= 1
        tp[k] = (lam + bhi) / (-ahi)
        hi_i = hi + 1
        x[hi_i] = tp[k]
        a[lo_i], b[lo_i] = alo, blo + lam
        a[hi_i], b[hi_i] = ahi, bhi + lam
        afirst, bfirst = 1.0, -lam - y[k + 1]
        alast, blast = -1.0, -lam + y[k + 1]
    alo, blo = afirst, bfirst
    lo = lo_i
    while lo <= hi_i:
        if alo * x[lo] + blo > 0:
            break
        alo += a[lo]
        blo += b[lo]
        lo += 1
    beta = np.zeros(n)
    beta[n - 1] = -blo / alo
    for k in range(n - 2, -1, -1):
        if beta[k + 1] > tp[k]:
            beta[k] = tp[k]
        elif beta[k + 1] < tm[k]:
            beta[k] = tm[k]
        else:
            beta[k] = beta[k + 1]
    return beta


def tv1d_kkt_violation(x, v, lam):
    """KKT certificate for tv1d: with c_k = sum_{i<=k}(x_i - v_i),
    need |c_k| <= lam, c_k = lam*sign(x_{k+1}-x_k) where x jumps, and c_n = 0.
    Returns (max bound violation, max jump-sign violation, |c_n|)."""
    x = np.asarray(x, dtype=np.float64)
    v = np.asarray(v, dtype=np.float64)
    c = np.cumsum(x - v)
    n = x.shape[0]
    if n == 1:
        return 0.0, 0.0, abs(c[-1])
    ck = c[:-1]
    bound = max(0.0, float(np.max(np.abs(ck)) - lam))
    d = np.diff(x)
    jump = d != 0
    # stationarity: (x_k - v_k) + lam*(s_{k-1} - s_k) = 0 with s_k in sign(x_{k+1}-x_k)
    # => c_k = lam * s_k
    js = float(np.max(np.abs(ck[jump] - lam * np.sign(d[jump])))) if jump.any() else 0.0
    return bound, js, abs(float(c[-1]))


class TotalVariation1DProx(VectorProx):  # prox/total_variation_1d.cc:7-25
    def apply_vector(self, ins):
        return [tv1d_prox(ins[0], self.lam_scalar())]


def _block_kkt_prox(M, b0, var_keys):
    chol = BlockCholesky().compute(M)

    def apply(v):
        sol = chol.solve(b0 + v)
        return sol.select(var_keys) if var_keys is not None else sol
    return chol, apply


class SumSquareProx(object):  # prox/sum_square.cc:10-40
    def init(self, arg):
        H, g, A = arg.H.A, arg.H.b, arg.A.A
        alpha = math.sqrt(2 * arg.f.alpha)
        M = ((H + H.T()).scaled(alpha) + (A + A.T())
             - H.left_identity() - A.left_identity())
        self.chol = BlockCholesky().compute(M)
        self.b = g.scaled(-alpha)
        self.var_keys = H.col_keys()

    def apply(self, v):
        return self.chol.solve(self.b + v).select(self.var_keys)


class ZeroProx(object):  # prox/zero.cc:10-36
    def init(self, arg):
        H, g, A = arg.H.A, arg.H.b, arg.A.A
        M = H + H.T() + A + A.T() - A.left_identity()
        self.chol = BlockCholesky().compute(M)
        self.b = g.scaled(-1.0)
        self.var_keys = H.col_keys()

    def apply(self, v):
        return self.chol.solve(self.b + v).select(self.var_keys)


class AffineProx(object):  # prox/affine.cc:8-49
    def init(self, arg):
        A, b = arg.A.A, arg.A.b
        alpha = arg.f.alpha
        c = BlockVector()
        if arg.f.prox_function_type == ProxFunction.AFFINE:
            for r, col, v in arg.H.A.entries():  # GetLinear :8-17
                c.set(col, v.as_dense().T.reshape(-1, order="F"))
            c = c.scaled(alpha)
        M = A + A.T() - A.left_identity()
        self.chol = BlockCholesky().compute(M)
        self.g = b.scaled(-1.0) - c

    def apply(self, v):
        return self.chol.solve(self.g + v)


class OrthoInvariantProx(VectorProx):  # prox/ortho_invariant.cc:7-116
    eigen_prox_type = ProxFunction.NORM_1
    symmetric_part = False
    add_residual = False
    epigraph = False

    def init(self, arg):
        VectorProx.init(self, arg)
        self.m_ = arg.f.arg_size[0].dim[0]
        self.n_ = arg.f.arg_size[0].dim[1]
        self.eigen_prox = None

    def _init_eigen_prox(self, lam):  # :76-98
        # The reference sizes the nested prox with min(m, n) (:77) but feeds it the n
        # eigenvalues of Y^T Y (:36-50); for m < n its parameter vectors are then indexed out of
        # bounds (undefined behaviour).  The well-defined reading - n entries - is used here.
        n = self.n_
        nargs = 2 if self.epigraph else 1
        self.alpha_ = 1.0 if self.epigraph else 1.0 / math.sqrt(lam)
        f = ProxFunction(prox_function_type=self.eigen_prox_type, alpha=1.0,
                         arg_size=[wire.Size(dim=[n, 1])])
        H, A = AffineOperator(), AffineOperator()
        for i in range(nargs):
            # the reference gives the epigraph's t argument n entries too (:88-92); only its
            # size-agnostic scalar maps are used, so 1 entry is the same operator
            ni = n if i == 0 else 1
            H.A.set(arg_key(i), arg_key(i), LM.identity(ni))
            A.A.set(arg_key(i), arg_key(i), LM.scalar(self.alpha_, ni))
        self.eigen_prox = create_prox_operator(self.eigen_prox_type, self.epigraph)
        self.eigen_prox.init(ProxArg(f, {}, H, A))

    def apply_vector(self, ins):  # :13-73
        if self.eigen_prox is None:
            self._init_eigen_prox(1.0 if self.epigraph else self.lam_scalar())
        Y = ins[0].reshape((self.m_, self.n_), order="F")
        R = (Y - Y.T) / 2 if self.add_residual else None
        if self.symmetric_part:
            d, V = np.linalg.eigh((Y + Y.T) / 2)
            U = V
        else:
            G = Y.T @ Y + 1e-15 * np.eye(self.n_)
            d, V = np.linalg.eigh(G)
            d = np.sqrt(np.maximum(d, 0.0))
            dinv = np.zeros_like(d)
            nz = d != 0
            dinv[nz] = 1.0 / d[nz]
            U = Y @ V @ np.diag(dinv)
        if self.epigraph:  # :52-60,107-116
            s = float(ins[1][0])
            out = self.eigen_prox.apply(BlockVector({arg_key(0): d, arg_key(1): np.array([s])}))
            X = U @ np.diag(out(arg_key(0))) @ V.T
            return [X.reshape(-1, order="F"), np.array([float(out(arg_key(1))[0])])]
        inp = BlockVector({arg_key(0): self.alpha_ * d})  # :100-105
        x_tilde = self.eigen_prox.apply(inp)(arg_key(0))
        X = U @ np.diag(x_tilde) @ V.T
        if self.add_residual:
            X = X + R
        elif self.symmetric_part:
            X = (X + X.T) / 2
        return [X.reshape(-1, order="F")]


def _ortho(eigen_type, symmetric=False, residual=False, epi=False):
    return type("OrthoInvariant_%d_%d%d%d" % (eigen_type, symmetric, residual, epi),
                (OrthoInvariantProx,),
                dict(eigen_prox_type=eigen_type, symmetric_part=symmetric, add_residual=residual,
                     epigraph=epi))


# ---- sort-based vector operators -------------------------------------------------------------


class MaxProx(VectorProx):  # prox/max.cc:7-43
    def apply_vector(self, ins):
        v, lam = ins[0], self.lam_scalar()
        y = np.sort(v)[::-1]
        t, acc, div = 0.0, -lam, 0.0
        for yi in y:
            if yi * div < acc:
                break
            acc += yi
            div += 1
            t = acc / div
        return [np.minimum(v, t)]


class MaxEpigraph(VectorProx):  # prox/max.cc:46-87
    def apply_vector(self, ins):
        v, s = ins[0], float(ins[1][0])
        y = np.sort(v)[::-1]
        if s >= y[0]:
            return [v.copy(), np.array([s])]
        delta, acc, div = 0.0, 0.0, 1.0
        for yi in y:
            if div * (yi - s) < acc:
                break
            acc += yi - s
            div += 1
            delta = acc / div
        t = s + delta
        return [v - np.maximum(0.0, v - t), np.array([t])]


def apply_sum_largest(v, lam, k):  # prox/sum_largest.cc:17-62
    n = v.shape[0]
    y = np.sort(v)[::-1]
    q, acc, inside, i, j = 0.0, -k * lam, 0, 0, 0
    while i < n and j < n:
        if y[i] * inside <= acc and (y[j] - lam) * inside <= acc:
            break
        if y[i] >= y[j] - lam:
            acc += y[i]
            inside += 1
            i += 1
        else:
            acc += -y[j] + lam
            inside -= 1
            j += 1
        with np.errstate(divide="ignore", invalid="ignore"):
            q = float(np.float64(acc) / np.float64(inside))  # inside == 0 -> +-inf as in C++
    return v - np.maximum(0.0, np.minimum(lam, v - q))


def eval_sum_largest(x, k):  # prox/sum_largest.cc:64-78
    return float(np.sum(np.sort(x)[::-1][:k]))


class SumLargestProx(VectorProx):
    def init(self, arg):
        VectorProx.init(self, arg)
        self.k = arg.f.sum_largest_params.k

    def apply_vector(self, ins):
        return [apply_sum_largest(ins[0], self.lam_scalar(), self.k)]


class SumLargestEpigraph(SumLargestProx):  # BisectionEpigraph, prox/newton.cc:239-288
    def apply_vector(self, ins):
        v, s = ins[0], float(ins[1][0])
        if eval_sum_largest(v, self.k) <= s:
            return [v.copy(), np.array([s])]
        lam, eps = 1.0, 1e-5
        upper, lower, upper_fixed = lam, 0.0, False
        x = v
        for _ in range(100):
            x = apply_sum_largest(v, lam, self.k)
            g = eval_sum_largest(x, self.k) - (lam + s)
            if abs(g) <= eps:
                return [x, np.array([lam + s])]
            if g > 0 and not upper_fixed:
                lam *= 2
                upper = lam
            elif g > 0:
                lower = lam
                lam = (lam + upper) / 2
            else:
                upper = lam
                lam = (lam + lower) / 2
                upper_fixed = True
        # the reference leaves output 1 unset when the bisection runs out (:288)
        return [x, np.array([lam + s])]


class SecondOrderConeProx(object):  # prox/second_order_cone.cc:6-124
    def init(self, arg):
        f = arg.f
        check(len(f.arg_size) == 2)
        self.m_, self.n_ = f.arg_size[1].dim[0], f.arg_size[1].dim[1]
        H, g = arg.H.A, arg.H.b
        self.t_key = self.x_key = None
        for col in H.col_keys():  # GetArgKeys :6-19
            for row, _ in H.col(col):
                if row == arg_key(0):
                    self.t_key = col
                elif row == arg_key(1):
                    self.x_key = col
                else:
                    raise CheckError("Unknown row key " + row)
        at = get_scalar(H.get(arg_key(0), self.t_key))
        ax = get_scalar(H.get(arg_key(1), self.x_key))
        # BlockVector::Get(key, n): zeros when the key is absent (block_vector.cc:57-63)
        bt = g(arg_key(0)) if g.has_key(arg_key(0)) else np.zeros(self.m_)
        bx = g(arg_key(1)) if g.has_key(arg_key(1)) else np.zeros(self.m_ * self.n_)
        self.a = at / abs(ax)
        self.bx = bx / ax
        self.bt = bt / abs(ax)
        A = arg.A.A  # InitConstraints :109-122
        AT = A.T()
        ATA = AT @ A
        alphat = get_scalar(ATA.get(self.t_key, self.t_key))
        alphax = get_scalar(ATA.get(self.x_key, self.x_key))
        check(alphat == alphax, "A'A not scalar matrix")
        D = BlockMatrix()
        D.set(self.x_key, self.x_key, LM.scalar(1 / alphat, self.m_ * self.n_))
        D.set(self.t_key, self.t_key, LM.scalar(1 / alphat, self.m_))
        self.AT = D @ AT

    def apply(self, v):  # :46-56
        u = self.AT.apply(v)
        X = (u(self.x_key) + self.bx).reshape((self.m_, self.n_), order="F")
        t = u(self.t_key) + self.bt / self.a
        beta = self.a
        v_norm = np.sqrt(np.sum(X * X, axis=1))
        beta2 = beta * beta
        with np.errstate(divide="ignore", invalid="ignore"):
            alpha = (1 / (beta2 + 1)) * (beta2 + beta * t / v_norm)
        t = t.copy()
        for i in range(self.m_):  # :60-77
            if np.isnan(alpha[i]) or alpha[i] > 1:
                alpha[i] = 1
            elif alpha[i] < 0:
                alpha[i] = 0
                t[i] = 0
            else:
                t[i] = (1 / beta) * alpha[i] * v_norm[i]
        X = alpha[:, None] * X
        x = BlockVector()
        x.set(self.x_key, X.reshape(-1, order="F") - self.bx)
        x.set(self.t_key, t - self.bt / self.a)
        return x


# ---- smooth functions and the Newton family (prox/newton.{h,cc}) --------------------------------


class SmoothFunction(object):  # newton.h:6-22
    def proj_feasible(self, x):
        return x


class ElemwiseSmoothFunction(SmoothFunction):  # newton.h:24-34
    def hess_inv(self, lam, x, v):
        return v / (1.0 + lam * self.hessf(x))


class SumExp(ElemwiseSmoothFunction):  # prox/sum_exp.cc:11-36
    def eval(self, x):
        return float(np.sum(np.exp(x)))

    def gradf(self, x):
        return np.exp(x)

    hessf = gradf


class Logistic(ElemwiseSmoothFunction):  # prox/sum_logistic.cc:8-33
    def eval(self, x):
        return float(np.sum(np.log(1 + np.exp(x))))

    def gradf(self, x):
        return np.exp(x) / (1 + np.exp(x))

    def hessf(self, x):
        return np.exp(x) / (1 + np.exp(x)) ** 2


class NegativeEntropy(ElemwiseSmoothFunction):  # prox/sum_neg_entr.cc:11-42
    def eval(self, x):
        pos = x > 0
        return float(np.sum(x[pos] * np.log(x[pos])))

    def gradf(self, x):
        return 1 + np.log(x)

    def hessf(self, x):
        return 1 / x

    def proj_feasible(self, x):
        return np.maximum(x, 1e-6)


class InvPos(ElemwiseSmoothFunction):  # prox/sum_inv_pos.cc:11-39
    def eval(self, x):
        return float(np.sum(1 / x))

    def gradf(self, x):
        return -1 / (x * x)

    def hessf(self, x):
        return 2 / (x * x * x)

    def proj_feasible(self, x):
        return np.maximum(x, 1e-6)


class LogSumExp(SmoothFunction):  # prox/log_sum_exp.cc:20-68
    def eval(self, x):
        mx = float(np.max(x))
        return mx + math.log(float(np.sum(np.exp(x - mx))))

    def gradf(self, x):
        w = np.exp(x - np.max(x))
        return w / np.sum(w)

    def hess_inv(self, lam, x, v):
        lam = float(np.atleast_1d(lam)[0])
        w = self.gradf(x)
        t = float(np.sum(w * w / (1 + lam * w)))
        r = float(np.sum(v * w / (1 + lam * w)))
        s = lam * r / (1 - lam * t)
        return v / (1 + lam * w) + w / (1 + lam * w) * s


def _prox_residual(f, lam, x, v):  # newton.cc:8-14
    return x - v + lam * f.gradf(x)


def apply_newton_prox(f, lam, v):  # newton.cc:49-103 (lam: vector or scalar)
    n = v.shape[0]
    lam = np.full(n, lam) if np.isscalar(lam) else lam
    eps = max(1e-12, 1e-10 / n)
    x = f.proj_feasible(v)
    for _ in range(100):
        gx = _prox_residual(f, lam, x, v)
        dx = f.hess_inv(lam, x, gx)
        beta, gamma, theta = 0.001, 0.5, 1.0
        x_res = float(np.linalg.norm(gx))
        while theta > eps:
            nx = f.proj_feasible(x - theta * dx)
            nx_res = float(np.linalg.norm(_prox_residual(f, lam, nx, v)))
            if nx_res <= (1 - beta * theta) * x_res:
                x, x_res = nx, nx_res
                break
            theta *= gamma
        if x_res < eps:
            break
    return x


def _make_newton_prox(fcls):
    class _NewtonProx(VectorProx):  # newton.cc:105-112
        def apply_vector(self, ins):
            return [apply_newton_prox(fcls(), self.lam_vec_for(ins[0]), ins[0])]
    return _NewtonProx


def _epigraph_residual(f, lam, x, t, v, s):  # newton.cc:16-27
    return np.concatenate([x - v + lam * f.gradf(x), [t - s - lam, f.eval(x) - t]])


def _make_newton_epigraph(fcls):
    class _NewtonEpigraph(VectorProx):  # newton.cc:114-194
        def apply_vector(self, ins):
            f = fcls()
            v, s = ins[0], float(ins[1][0])
            n = v.shape[0]
            eps = max(1e-12, 1e-10 / n)
            x = f.proj_feasible(v)
            if float(np.linalg.norm(v - x)) < eps and f.eval(x) <= s:
                return [v.copy(), np.array([s])]
            t, lam = s, 1.0
            for _ in range(100):
                g = _epigraph_residual(f, lam, x, t, v, s)
                nt_step = f.hess_inv(lam, x, g[:n])
                nt_res = float(g[:n] @ nt_step)
                scale = (-nt_res + g[n] + g[n + 1]) / (nt_res + 1)
                step_x = (1 + scale) * nt_step
                step_t = g[n] - scale
                step_l = -scale
                beta, gamma, theta = 0.001, 0.5, 1.0
                x_res = float(np.linalg.norm(g))
                while theta > eps:
                    nx = f.proj_feasible(x - theta * step_x)
                    nt = t - theta * step_t
                    nlam = max(lam - theta * step_l, eps)
                    nx_res = float(np.linalg.norm(_epigraph_residual(f, nlam, nx, nt, v, s)))
                    if nx_res <= (1 - beta * theta) * x_res:
                        x, t, lam, x_res = nx, nt, nlam, nx_res
                        break
                    theta *= gamma
                if x_res < eps:
                    break
            return [x, np.array([t])]
    return _NewtonEpigraph


def _make_implicit_newton_epigraph(fcls):
    class _ImplicitNewtonEpigraph(VectorProx):  # newton.cc:196-237
        def apply_vector(self, ins):
            f = fcls()
            v, s = ins[0], float(ins[1][0])
            x = f.proj_feasible(v)
            if f.eval(x) <= s:
                return [x, np.array([s])]
            lam = 1.0
            for _ in range(100):
                x = apply_newton_prox(f, lam, v)
                gx = f.gradf(x)
                glam = f.eval(x) - lam - s
                hlam = -float(f.hess_inv(lam, x, gx) @ gx) - 1
                if abs(glam) < 1e-10:
                    break
                lam = lam - glam / hlam
                if lam < 0:
                    lam = 1e-6
            return [apply_newton_prox(f, lam, v), np.array([s + lam])]
    return _ImplicitNewtonEpigraph


def apply_neg_log_prox(lam, v):  # prox/sum_neg_log.cc:9-24
    z = np.sqrt(v * v + 4 * lam)
    return np.where(v >= 0, (v + z) / 2, 2 * lam / (-v + z))


class SumNegLogProx(VectorProx):  # prox/sum_neg_log.cc:27-38
    def apply_vector(self, ins):
        return [apply_neg_log_prox(self.lam_vec_for(ins[0]), ins[0])]


class SumNegLogEpigraph(VectorProx):  # prox/sum_neg_log.cc:42-90
    def apply_vector(self, ins):
        v, s = ins[0], float(ins[1][0])
        n = v.shape[0]
        eps, lam = 1e-10, 1.0
        for _ in range(1000):
            z = np.sqrt(v * v + 4 * lam)
            pos = v >= 0
            g = -lam - s - float(np.sum(np.log((v[pos] + z[pos]) / 2))) \
                + float(np.sum(np.log((-v[~pos] + z[~pos]) / (2 * lam))))
            h = -1.0 - float(np.sum(1.0 / (v[pos] * (v[pos] + z[pos]) / 2 + 2 * lam))) \
                - float(np.sum(1.0 / (v[~pos] * 2 * lam / (-v[~pos] + z[~pos]) + 2 * lam)))
            if abs(g) <= eps:
                break
            if h >= -1e-10:
                h = -1e-10
            lam -= g / h
            if lam <= 1e-10:
                lam = 1e-10
        return [apply_neg_log_prox(np.full(n, lam), v), np.array([s + lam])]


def apply_kl_div_prox(lam, u, v):  # prox/sum_kl_div.cc:6-54
    eps = 1e-13
    n = u.shape[0]
    x, y = np.zeros(n), np.zeros(n)
    for i in range(n):
        li, ui, vi = float(lam[i]), float(u[i]), float(v[i])
        yhat = max((0.5 + li - vi) / li, eps)
        if abs(ui) < eps * eps and abs(vi) < eps * eps:
            x[i], y[i] = ui, vi
            continue
        for _ in range(1000):
            f = li * yhat * yhat + (vi - li) * yhat - ui + li * math.log(yhat)
            F = 2 * li * yhat + (vi - li) + li / yhat
            res = f / F
            if abs(res) < eps or (yhat <= eps * 2 and res > 0) or \
                    (li * yhat + vi - li <= eps * 2 and res > 0):
                break
            yhat = yhat - res
            if yhat < eps:
                yhat = eps
            if li * yhat + vi - li < eps:
                yhat = (eps + li - vi) / li
        y[i] = li * yhat + vi - li
        x[i] = y[i] * yhat
    return x, y


class SumKLDivProx(VectorProx):  # prox/sum_kl_div.cc:56-69
    def apply_vector(self, ins):
        x, y = apply_kl_div_prox(self.lam_vec_for(ins[0]), ins[0], ins[1])
        return [x, y]


class SumKLDivEpigraph(VectorProx):  # prox/sum_kl_div.cc:71-126
    def apply_vector(self, ins):
        u, v, s = ins[0], ins[1], float(ins[2][0])
        n = u.shape[0]
        eps, lam = 1e-10, 1.0
        for _ in range(100):
            x, y = apply_kl_div_prox(np.full(n, lam), u, v)
            glam, hlam = -s - lam, -1.0
            for i in range(n):
                glam += x[i] * math.log(x[i] / y[i]) - x[i] + y[i]
                g = np.array([math.log(x[i] / y[i]), -x[i] / y[i] + 1])
                h = np.eye(2) + lam * np.array([[1 / x[i], -1 / y[i]],
                                                [-1 / y[i], x[i] / (y[i] * y[i])]])
                hlam -= float(g @ np.linalg.solve(h, g))
            if abs(glam) < eps or (lam <= eps * 2 and glam / hlam > 0):
                break
            lam = lam - glam / hlam
            if lam < eps:
                lam = eps
        x, y = apply_kl_div_prox(np.full(n, lam), u, v)
        return [x, y, np.array([s + lam])]


class ExpEpigraph(VectorProx):  # prox/exp.cc:12-77 (elementwise: x, t and s are all vectors)
    def apply_vector(self, ins):
        v, s = ins[0], ins[1]
        x, t, l = v.copy(), s.copy(), np.ones(v.shape[0])
        eps = 1e-12
        for _ in range(100):
            ex = np.exp(x)
            r0 = (x - v) + l * ex
            r1 = t - s - l
            r2 = ex - t
            if max(np.max(np.abs(r0)), np.max(np.abs(r1)), np.max(np.abs(r2))) < eps:
                break
            h = 1 + l * ex
            d = ex
            dl = (-d * r0 + h * (r1 + r2)) / (d * d + h)
            dx = -(d * dl + r0) / h
            dt = dl - r1
            x, t, l = x + dx, t + dt, l + dl
        easy = np.exp(v) <= s
        x[easy] = v[easy]
        t[easy] = s[easy]
        return [x, t]


_PROX_REGISTRY = {
    (ProxFunction.SUM_SQUARE, True): SumSquareEpigraph,
    (ProxFunction.NORM_1, True): ScaledZoneEpigraph,
    (ProxFunction.SUM_DEADZONE, True): ScaledZoneEpigraph,
    (ProxFunction.SUM_HINGE, True): ScaledZoneEpigraph,
    (ProxFunction.SUM_QUANTILE, True): ScaledZoneEpigraph,
    (ProxFunction.NORM_1, False): ScaledZoneProx,
    (ProxFunction.SUM_DEADZONE, False): ScaledZoneProx,
    (ProxFunction.SUM_HINGE, False): ScaledZoneProx,
    (ProxFunction.SUM_QUANTILE, False): ScaledZoneProx,
    (ProxFunction.NORM_2, False): Norm2Prox,
    (ProxFunction.NON_NEGATIVE, False): NonNegativeProx,
    (ProxFunction.TOTAL_VARIATION_1D, False): TotalVariation1DProx,
    (ProxFunction.SUM_SQUARE, False): SumSquareProx,
    (ProxFunction.ZERO, False): ZeroProx,
    (ProxFunction.AFFINE, False): AffineProx,
    (ProxFunction.CONSTANT, False): AffineProx,
    (ProxFunction.NORM_NUCLEAR, False): OrthoInvariantProx,
    (ProxFunction.NORM_NUCLEAR, True): _ortho(ProxFunction.NORM_1, False, False, True),
    (ProxFunction.LAMBDA_MAX, False): _ortho(ProxFunction.MAX, True),
    (ProxFunction.LAMBDA_MAX, True): _ortho(ProxFunction.MAX, True, False, True),
    (ProxFunction.NEG_LOG_DET, False): _ortho(ProxFunction.SUM_NEG_LOG, True),
    (ProxFunction.NEG_LOG_DET, True): _ortho(ProxFunction.SUM_NEG_LOG, True, False, True),
    (ProxFunction.SEMIDEFINITE, False): _ortho(ProxFunction.NON_NEGATIVE, True, True),
    (ProxFunction.MAX, False): MaxProx,
    (ProxFunction.MAX, True): MaxEpigraph,
    (ProxFunction.SUM_LARGEST, False): SumLargestProx,
    (ProxFunction.SUM_LARGEST, True): SumLargestEpigraph,
    (ProxFunction.SECOND_ORDER_CONE, False): SecondOrderConeProx,
    (ProxFunction.SUM_EXP, False): _make_newton_prox(SumExp),
    (ProxFunction.SUM_EXP, True): _make_newton_epigraph(SumExp),
    (ProxFunction.SUM_LOGISTIC, False): _make_newton_prox(Logistic),
    (ProxFunction.SUM_LOGISTIC, True): _make_newton_epigraph(Logistic),
    (ProxFunction.SUM_INV_POS, False): _make_newton_prox(InvPos),
    (ProxFunction.SUM_INV_POS, True): _make_newton_epigraph(InvPos),
    (ProxFunction.SUM_NEG_ENTR, False): _make_newton_prox(NegativeEntropy),
    (ProxFunction.SUM_NEG_ENTR, True): _make_implicit_newton_epigraph(NegativeEntropy),
    (ProxFunction.LOG_SUM_EXP, False): _make_newton_prox(LogSumExp),
    (ProxFunction.LOG_SUM_EXP, True): _make_newton_epigraph(LogSumExp),
    (ProxFunction.SUM_NEG_LOG, False): SumNegLogProx,
    (ProxFunction.SUM_NEG_LOG, True): SumNegLogEpigraph,
    (ProxFunction.SUM_KL_DIV, False): SumKLDivProx,
    (ProxFunction.SUM_KL_DIV, True): SumKLDivEpigraph,
    (ProxFunction.EXP, True): ExpEpigraph,
}


def create_prox_operator(type_, epigraph):  # prox/prox.cc:29-38
    cls = _PROX_REGISTRY.get((type_, bool(epigraph)))
    if cls is None:
        raise CheckError("No proximal operator for %s (epigraph: %d)" %
                         (ProxFunction.type_name(type_), epigraph))
    return cls()


# =============================================================================================
# ADMM drivers (algorithms/prox_admm.cc, algorithms/prox_admm_two_block.cc)
# =============================================================================================


class ProxADMMSolver(object):
    def __init__(self, problem, data, params):
        self.problem, self.data, self.params = problem, data, params
        self.status = wire.SolverStatus(residuals=wire.Residuals())
        self.initialized = False
        self.trace = None  # optional callback(iter, solver)

    def init_constraints(self):  # prox_admm.cc:25-43
        self.A, self.b = BlockMatrix(), BlockVector()
        for i, constr in enumerate(self.problem.constraint):
            check(constr.expression_type == Expression.INDICATOR)
            check(constr.cone is not None and constr.cone.cone_type == wire.Cone.ZERO)
            check(len(constr.arg) == 1)
            build_affine_operator(constr.arg[0], self.data, constraint_key(i), self.A, self.b)
        self.AT = self.A.T()
        self.m, self.n = self.A.m(), self.A.n()

    def init_prox_operators(self):  # :45-94
        obj = self.problem.objective
        check(obj.expression_type == Expression.ADD)
        self.N = len(obj.arg)
        check(self.params.rho == 1)
        sqrt_rho = math.sqrt(self.params.rho)
        self.prox, self.AiT = [], []
        constr_vars = set(self.A.col_keys())
        for i in range(self.N):
            f_expr = obj.arg[i]
            H = AffineOperator()
            for k, a in enumerate(f_expr.arg):
                build_affine_operator(a, self.data, arg_key(k), H.A, H.b)
            A = AffineOperator()
            for var_id in get_variables(f_expr):
                if var_id not in constr_vars:
                    continue
                for r, v in self.A.col(var_id):
                    A.A.set(r, var_id, lm_scale(sqrt_rho, v))
            pf = f_expr.prox_function
            op = create_prox_operator(pf.prox_function_type, pf.epigraph)
            op.init(ProxArg(pf, self.data, H, A))
            self.prox.append(op)
            self.AiT.append(A.A.T())

    def init_variables(self):  # :96-108
        self.x = [BlockVector() for _ in range(self.N)]
        self.y = [BlockVector() for _ in range(self.N)]
        self.u = BlockVector()
        for i, constr in enumerate(self.problem.constraint):
            self.u.set(constraint_key(i), np.zeros(get_dimension(constr.arg[0])))

    def init(self):  # :110-129
        self.init_constraints()
        self.init_prox_operators()
        if not self.params.warm_start or not self.initialized:
            self.init_variables()
            self.initialized = True

    def sweep(self):  # :135-147
        self.y_prev = [y.copy() for y in self.y]
        self.u.isub(self.b)
        for i in range(self.N):
            self.u.isub(self.y[i])
        for i in range(self.N):
            self.u.iadd(self.y[i])
            self.x[i] = self.prox[i].apply(self.u)
            self.y[i] = self.A.apply(self.x[i])
            self.u.isub(self.y[i])

    def solve(self):  # :131-169
        self.init()
        p = self.params
        it = 0
        while it < p.max_iterations:
            self.iter = it
            self.sweep()
            if self.trace:
                self.trace(it, self)
            if it % p.epoch_iterations == 0:
                self.compute_residuals()
                if self.status.state == wire.SolverStatus.OPTIMAL:
                    break
            it += 1
        self.iter = it
        if it == p.max_iterations:
            self.compute_residuals()
            self.status.state = wire.SolverStatus.MAX_ITERATIONS_REACHED
        return self.get_solution()

    def get_solution(self):  # :171-176
        r = BlockVector()
        for i in range(self.N):
            r.iadd(self.x[i])
        return r

    def compute_residuals(self):  # :178-217
        p = self.params
        Ax_b = self.b.copy()
        max_norm = self.b.norm()
        for i in range(self.N):
            Ai_xi = self.A.apply(self.x[i])
            max_norm = max(max_norm, Ai_xi.norm())
            Ax_b.iadd(Ai_xi)
        s2 = 0.0
        Ax_diff = BlockVector()
        for i in range(self.N - 2, -1, -1):
            Ax_diff.iadd(self.y[i + 1] - self.y_prev[i + 1])
            s_i = self.AiT[i].apply(Ax_diff).norm()
            s2 += s_i * s_i
        r = self.status.residuals
        r.r_norm = Ax_b.norm()
        r.s_norm = p.rho * math.sqrt(s2)
        r.epsilon_primal = p.abs_tol * math.sqrt(self.m) + p.rel_tol * max_norm
        r.epsilon_dual = p.abs_tol * math.sqrt(self.n) + p.rel_tol * p.rho * self.AT.apply(self.u).norm()
        if r.r_norm <= r.epsilon_primal and r.s_norm <= r.epsilon_dual:
            self.status.state = wire.SolverStatus.OPTIMAL
        else:
            self.status.state = wire.SolverStatus.RUNNING
        self.status.num_iterations = self.iter


class ProxADMMTwoBlockSolver(object):
    def __init__(self, problem, data, params):
        self.problem, self.data, self.params = problem, data, params
        self.status = wire.SolverStatus(residuals=wire.Residuals())
        self.trace = None

    def init(self):  # prox_admm_two_block.cc:21-94
        p = self.params
        sqrt_rho = math.sqrt(p.rho)
        H, A = AffineOperator(), AffineOperator()
        self.z, self.u = BlockVector(), BlockVector()
        for i, constr in enumerate(self.problem.constraint):
            check(constr.expression_type == Expression.INDICATOR)
            check(constr.cone is not None and constr.cone.cone_type == wire.Cone.ZERO)
            check(len(constr.arg) == 1)
            build_affine_operator(constr.arg[0], self.data, constraint_key(i), H.A, H.b)
            for var_id, e in get_variables(constr).items():
                A.A.set(var_id, var_id, lm_scale(sqrt_rho, LM.identity(get_dimension(e))))
                self.z.set(var_id, np.zeros(get_dimension(e)))
        self.constr_prox = create_prox_operator(ProxFunction.ZERO, False)
        self.constr_prox.init(ProxArg(ProxFunction(), self.data, H, A))
        self.m, self.n = H.A.m(), H.A.n()
        obj = self.problem.objective
        check(obj.expression_type == Expression.ADD)
        self.N = len(obj.arg)
        self.prox = []
        for i in range(self.N):
            f_expr = obj.arg[i]
            Hi = AffineOperator()
            for k, a in enumerate(f_expr.arg):
                build_affine_operator(a, self.data, arg_key(k), Hi.A, Hi.b)
            Ai = AffineOperator()
            for var_id, e in get_variables(f_expr).items():
                Ai.A.set(var_id, var_id, lm_scale(sqrt_rho, LM.identity(get_dimension(e))))
            pf = f_expr.prox_function
            op = create_prox_operator(pf.prox_function_type, pf.epigraph)
            op.init(ProxArg(pf, self.data, Hi, Ai))
            self.prox.append(op)

    def solve(self):  # :96-133
        self.init()
        p = self.params
        it = 0
        while it < p.max_iterations:
            self.iter = it
            self.z_prev = self.z.copy()
            zu = self.z - self.u
            self.x = BlockVector()
            for i in range(self.N):
                self.x.iadd(self.prox[i].apply(zu))
            self.z = self.constr_prox.apply(self.x + self.u)
            self.u.iadd(self.x - self.z)
            if self.trace:
                self.trace(it, self)
            if it % p.epoch_iterations == 0:
                self.compute_residuals()
                if self.status.state == wire.SolverStatus.OPTIMAL:
                    break
            it += 1
        self.iter = it
        if it == p.max_iterations:
            self.compute_residuals()
            self.status.state = wire.SolverStatus.MAX_ITERATIONS_REACHED
        return self.x

    def compute_residuals(self):  # :135-156
        p = self.params
        r = self.status.residuals
        r.r_norm = (self.x - self.z).norm()
        r.s_norm = p.rho * (self.z - self.z_prev).norm()
        r.epsilon_primal = p.abs_tol * math.sqrt(self.n) + p.rel_tol * max(self.x.norm(), self.z.norm())
        r.epsilon_dual = p.abs_tol * math.sqrt(self.n) + p.rel_tol * p.rho * self.u.norm()
        if r.r_norm <= r.epsilon_primal and r.s_norm <= r.epsilon_dual:
            self.status.state = wire.SolverStatus.OPTIMAL
        else:
            self.status.state = wire.SolverStatus.RUNNING
        self.status.num_iterations = self.iter


# =============================================================================================
# entry points (python/epopt/solvemodule.cc)
# =============================================================================================


def create_solver(problem, data, params):  # solvemodule.cc:74-87
    if params.solver == wire.SolverParams.PROX_ADMM:
        return ProxADMMSolver(problem, data, params)
    if params.solver == wire.SolverParams.PROX_ADMM_TWO_BLOCK:
        return ProxADMMTwoBlockSolver(problem, data, params)
    raise CheckError("Unknown solver")


def solve(problem_bytes, parameters, params_bytes, data, trace=None):
    """solvemodule.cc:110-187 -> (SolverStatus bytes, {var_id: float64 bytes})."""
    problem = wire.Problem.FromString(problem_bytes)
    params = wire.SolverParams.FromString(params_bytes)
    if parameters:
        data = dict(data)
        data[PARAMS_KEY] = {pid: wire.Constant.FromString(cb) for pid, cb in parameters}
    solver = create_solver(problem, data, params)
    solver.trace = trace
    x = solver.solve()
    out = {}
    for var_id in get_variables(problem):
        out[var_id] = x(var_id).astype(np.float64).tobytes()
    return solver.status.SerializeToString(), out


def eval_prox(f_expr_bytes, lam, data, v_map):
    """solvemodule.cc:189-242 -> {var_id: float64 bytes}."""
    f_expr = wire.Expression.FromString(f_expr_bytes)
    check(f_expr.expression_type == Expression.PROX_FUNCTION)
    H, A = AffineOperator(), AffineOperator()
    for i, a in enumerate(f_expr.arg):
        build_affine_operator(a, data, arg_key(i), H.A, H.b)
    i = 0
    for var_id, e in get_variables(f_expr).items():
        A.A.set(constraint_key(i), var_id,
                lm_scale(1.0 / math.sqrt(lam), LM.identity(get_dimension(e))))
        i += 1
    vin = BlockVector({k: np.frombuffer(b, dtype=np.float64) for k, b in v_map.items()})
    v = A.A.apply(vin)
    pf = f_expr.prox_function
    op = create_prox_operator(pf.prox_function_type, pf.epigraph)
    op.init(ProxArg(pf, data, H, A))
    x = op.apply(v)
    return {k: val.astype(np.float64).tobytes() for k, val in x.items()}
