"""Compiled (separable prox-affine) forms of the benchmark problems.

The reference builds these with cvxpy + its compiler (python/epopt/problems/*.py ->
compiler/transforms/{prox,separate}.py); neither runs on python 3.  The *compiled*
IR each problem ends up as is fixed by the prox rules and is written out in
SURVEY.md 3.3 / 3.4b, so it is constructed directly here, with the same data recipes:

  lasso        python/epopt/problems/lasso.py:8-15 + problem_util.py:9-42
               compiled form docs/solver.rst:33-39
  tv_1d        python/epopt/problems/tv_1d.py:5-20, compiler_test.py:51-57
  robust_pca   python/epopt/problems/robust_pca.py:5-22
  mnist hinge  python/epopt/functions.py:36-40, docs/notebooks/mnist.rst:118-129

The RNG stream of 2016 numpy is not reproduced: same distributions, own fixed seed.
"""

import numpy as np
import scipy.sparse as sp

from . import ir
from . import wire
from .wire import ProxFunction


def regression_data(m, n, rho=1.0, sigma=0.05, seed=0):
    """A = randn(m,n) with unit-l2 columns, x0 sparse-uniform support (density rho) with
    randn values, b = A x0 + sigma*randn(m).  reference problem_util.py:9-42."""
    rng = np.random.RandomState(seed)
    A = rng.randn(m, n)
    A /= np.sqrt(np.sum(A ** 2, 0))
    if rho >= 1:
        x0 = rng.randn(n)
    else:
        x0 = np.zeros(n)
        nnz = int(round(rho * n))
        idx = rng.choice(n, nnz, replace=False)
        x0[idx] = rng.randn(nnz)
    b = A.dot(x0) + sigma * rng.randn(m)
    return A, b


# The separate pass gives the FIRST function sharing a variable (sum_square) a copy named
# "separate:<old id>:<function node>", tied to the original by zero(copy - original); the user's
# variable stays with norm_1 (reference compiler/transforms/separate.py:64-85).
LASSO_COPY = "separate:var:x:sum_square"
LASSO_VAR = "var:x"


def lasso_ir(A_map, b_expr, lam, n):
    """sum_square(A x' - b) + lam*norm_1(x)  s.t.  x' - x = 0   (docs/solver.rst:33-39)."""
    x = ir.variable(n, 1, LASSO_COPY)
    y = ir.variable(n, 1, LASSO_VAR)
    m = A_map.m
    f0 = ir.prox(ProxFunction.SUM_SQUARE,
                 ir.add(ir.linear_map(A_map, x),
                        ir.linear_map(ir.scalar(-1, m), b_expr)),
                 alpha=1.0)
    f1 = ir.prox(ProxFunction.NORM_1, y, alpha=lam)
    c = ir.zero(ir.add(x, ir.linear_map(ir.scalar(-1, n), y)))
    return ir.Problem([f0, f1], [c])


def lasso(m, n, rho=1.0, seed=0):
    A, b = regression_data(m, n, rho=rho, seed=seed)
    lam = 0.5 * np.abs(A.T.dot(b)).max()
    prob = lasso_ir(ir.dense_matrix(A), ir.constant(b), lam, n)
    return prob, dict(A=A, b=b, lam=lam)


CONSENSUS_Z = "var:z"


def consensus_row_range(m, rank, world):
    per = -(-m // world)
    lo = min(m, rank * per)
    return lo, min(m, lo + per)


def consensus_lasso_local(A_g, b_g, lam, x_key="var:x_local"):
    """One rank's problem of the consensus-form lasso (examples split by rows):
    sum_square(A_g x_g - b_g) + lam*norm_1(z)  s.t.  x_g - z = 0.  x_g and the constraint row are
    sharded keys, z is replicated (include/epsilon_hip.h, eps_shard_consensus_terms)."""
    mg, n = A_g.shape
    x = ir.variable(n, 1, x_key)
    z = ir.variable(n, 1, CONSENSUS_Z)
    f = ir.prox(ProxFunction.SUM_SQUARE,
                ir.add(ir.linear_map(ir.dense_matrix(A_g), x),
                       ir.linear_map(ir.scalar(-1, mg), ir.constant(b_g))), alpha=1.0)
    h = ir.prox(ProxFunction.NORM_1, z, alpha=lam)
    c = ir.zero(ir.add(x, ir.linear_map(ir.scalar(-1, n), z)))
    return ir.Problem([f, h], [c])


def consensus_lasso(A, b, lam, world):
    """The same problem stacked for ONE process: terms f_1..f_G, h and G consensus constraints
    (what the per-rank solves must reproduce sweep for sweep)."""
    m, n = A.shape
    z = ir.variable(n, 1, CONSENSUS_Z)
    terms, cons = [], []
    for g in range(world):
        lo, hi = consensus_row_range(m, g, world)
        x = ir.variable(n, 1, "var:x_%d" % g)
        terms.append(ir.prox(ProxFunction.SUM_SQUARE,
                             ir.add(ir.linear_map(ir.dense_matrix(A[lo:hi]), x),
                                    ir.linear_map(ir.scalar(-1, hi - lo), ir.constant(b[lo:hi]))),
                             alpha=1.0))
        cons.append(ir.zero(ir.add(x, ir.linear_map(ir.scalar(-1, n), z))))
    terms.append(ir.prox(ProxFunction.NORM_1, z, alpha=lam))
    return ir.Problem(terms, cons)


def lasso_objective(A, b, lam, x):
    r = A.dot(x) - b
    return float(r.dot(r) + lam * np.abs(x).sum())


def tv_1d_data(n, seed=0):
    """reference tv_1d.py:5-17; the piecewise-constant signal is built from a difference
    array + prefix sum instead of the O(k n) python loop."""
    rng = np.random.RandomState(seed)
    k = max(int(np.sqrt(n) / 2), 1)
    idxs = rng.randint(0, n, (k, 2))
    idxs.sort()
    steps = 10 * (rng.rand(k) - 0.5)
    diff = np.zeros(n + 1)
    np.add.at(diff, idxs[:, 0], steps)
    np.add.at(diff, idxs[:, 1], -steps)
    x0 = 1.0 + np.cumsum(diff[:n])
    b = x0 + rng.randn(n)
    return b, float(np.sqrt(n))


def tv_1d(n, seed=0):
    """0.5*sum_square(x' - b) + lam*total_variation_1d(x)  s.t.  x' - x = 0."""
    b, lam = tv_1d_data(n, seed)
    xp = ir.variable(n, 1, "separate:var:x:sum_square")
    x = ir.variable(n, 1, "var:x")
    f0 = ir.prox(ProxFunction.TOTAL_VARIATION_1D, x, alpha=lam)
    f1 = ir.prox(ProxFunction.SUM_SQUARE,
                 ir.add(xp, ir.linear_map(ir.scalar(-1, n), ir.constant(b))), alpha=0.5)
    c = ir.zero(ir.add(xp, ir.linear_map(ir.scalar(-1, n), x)))
    return ir.Problem([f0, f1], [c]), dict(b=b, lam=lam)


def tv_1d_objective(b, lam, x):
    return float(0.5 * np.sum((x - b) ** 2) + lam * np.abs(np.diff(x)).sum())


def tv_1d_prox_expr(n, lam_alpha=1.0):
    """Single TOTAL_VARIATION_1D term for `eval_prox` (prox_test.py:209 cp.tv(x))."""
    x = ir.variable(n, 1, "var:x")
    return ir.prox(ProxFunction.TOTAL_VARIATION_1D, x, alpha=lam_alpha)


def robust_pca_data(n, r=10, density=0.1, seed=0):
    rng = np.random.RandomState(seed)
    L0 = rng.randn(n, r).dot(rng.randn(r, n))
    S0 = sp.rand(n, n, density, random_state=rng)
    S0.data = 10 * rng.randn(len(S0.data))
    return np.asarray(L0 + S0.toarray())


def robust_pca_ir(M, lam):
    """norm_nuclear(L) + lam*norm_1(S)  s.t.  L + S - M = 0   for an m x n block M:
    the whole matrix, or one rank's block of rows in a row-sharded solve (keys var:L, var:S and
    constraint:0 sharded; the nuclear-norm prox then runs its row-sharded SVD)."""
    m, n = M.shape
    L = ir.variable(m, n, "var:L")
    S = ir.variable(m, n, "var:S")
    f0 = ir.prox(ProxFunction.NORM_NUCLEAR, L, alpha=1.0)
    f1 = ir.prox(ProxFunction.NORM_1, S, alpha=lam)
    c = ir.zero(ir.add(L, S, ir.linear_map(ir.scalar(-1, m * n), ir.constant(M.reshape(-1, 1, order="F")))))
    return ir.Problem([f0, f1], [c])


def robust_pca(n, r=10, density=0.1, seed=0, lam=0.1):
    M = robust_pca_data(n, r, density, seed)
    return robust_pca_ir(M, lam), dict(M=M, lam=lam)


def robust_pca_objective(lam, Lm, Sm):
    return float(np.linalg.svd(Lm, compute_uv=False).sum() + lam * np.abs(Sm).sum())


def multiclass_hinge_data(m, nf, k, seed=0):
    rng = np.random.RandomState(seed)
    X = rng.rand(m, nf)
    y = rng.randint(0, k, m)
    Y = np.zeros((m, k))
    Y[np.arange(m), y] = 1.0
    return X, Y


def multiclass_hinge(X, Y, lam, c_vec=None):
    """Compiled form of sum_i max_j (X Theta + 1 - Y)_ij - <X^T Y, Theta> + lam||Theta||^2
    (functions.py:36-40) as printed at docs/notebooks/mnist.rst:118-129:

      affine(1^T t) + non_negative(y) + affine(-<X^T Y, Z>) + sum_square(W)[lam]
      zero(kron(1_k, I_m) t - (kron(I_k, X) W + 1 - vec(Y)) - y),   zero(Z - W)
    """
    m, nf = X.shape
    k = Y.shape[1]
    t = ir.variable(m, 1, "max_entries:t")
    y = ir.variable(m * k, 1, "non_negative:y")
    Z = ir.variable(nf * k, 1, "separate:var:Theta:affine")
    W = ir.variable(nf * k, 1, "var:Theta")
    f0 = ir.prox(ProxFunction.AFFINE,
                 ir.linear_map(ir.dense_matrix(np.ones((1, m))), t), alpha=1.0)
    f1 = ir.prox(ProxFunction.NON_NEGATIVE, y, alpha=1.0)
    if c_vec is None:  # a sample-sharded rank passes the GLOBAL -X^T Y here
        c_vec = -(X.T.dot(Y)).reshape(1, -1, order="F")
    f2 = ir.prox(ProxFunction.AFFINE, ir.linear_map(ir.dense_matrix(c_vec), Z), alpha=1.0)
    f3 = ir.prox(ProxFunction.SUM_SQUARE, W, alpha=lam)
    ones_k = ir.dense_matrix(np.ones((1, k)))
    lift_t = ir.kronecker_product(ir.transpose(ones_k), ir.identity(m))  # t 1^T
    XW = ir.linear_map(ir.kronecker_product(ir.identity(k), ir.dense_matrix(X)), W)
    const = ir.constant((1.0 - Y).reshape(-1, 1, order="F"))
    inner = ir.add(XW, const)
    c0 = ir.zero(ir.add(ir.linear_map(lift_t, t),
                        ir.linear_map(ir.scalar(-1, m * k), inner),
                        ir.linear_map(ir.scalar(-1, m * k), y)))
    c1 = ir.zero(ir.add(Z, ir.linear_map(ir.scalar(-1, nf * k), W)))
    return ir.Problem([f0, f1, f2, f3], [c0, c1]), dict(X=X, Y=Y, lam=lam)


def multiclass_hinge_objective(X, Y, lam, Theta):
    S = X.dot(Theta) + 1 - Y
    return float(S.max(axis=1).sum() - np.sum(X.T.dot(Y) * Theta) + lam * np.sum(Theta ** 2))


def group_lasso(m, n, k, lam=None, seed=0):
    """sum_square(A X' - B) + lam * sum_i ||X_i||_2 over the rows X_i  s.t.  X' - X = 0
    (reference python/epopt/problems/group_lasso.py: the NORM_2 prox carries axis = 1)."""
    rng = np.random.RandomState(seed)
    A = rng.randn(m, n)
    X0 = rng.randn(n, k) * (rng.rand(n, 1) < 0.2)
    B = A.dot(X0) + 0.1 * rng.randn(m, k)
    if lam is None:
        lam = 0.3 * np.sqrt((A.T.dot(B) ** 2).sum(axis=1)).max()
    Xp = ir.variable(n, k, "separate:var:X:sum_square")
    X = ir.variable(n, k, "var:X")
    f0 = ir.prox(ProxFunction.SUM_SQUARE,
                 ir.add(ir.linear_map(ir.left_matrix_product(ir.dense_matrix(A), k),
                                      ir.reshape(Xp, n * k, 1)),
                        ir.linear_map(ir.scalar(-1, m * k), ir.constant(B.reshape(-1, 1, order="F")))),
                 alpha=1.0, arg_size=[(m * k, 1)])
    f1 = ir.prox(ProxFunction.NORM_2, X, alpha=lam, has_axis=True, axis=1)
    c = ir.zero(ir.add(ir.reshape(Xp, n * k, 1),
                       ir.linear_map(ir.scalar(-1, n * k), ir.reshape(X, n * k, 1))))
    return ir.Problem([f0, f1], [c]), dict(A=A, B=B, lam=lam)


def group_lasso_objective(A, B, lam, X):
    return float(np.sum((A.dot(X) - B) ** 2) + lam * np.sqrt((X ** 2).sum(axis=1)).sum())


def mv_lasso_ir(A, B, lam):
    """sum_square(A X' - B) + lam * norm_1(X)  s.t.  X' - X = 0 for a matrix variable X (n x k):
    the data map is the Kronecker product I_k (x) A acting on vec(X') (reference lasso.py with
    k > 1, mnist.py:51-64; linear_map.proto KRONECKER_PRODUCT)."""
    m, n = A.shape
    k = B.shape[1]
    Xp = ir.variable(n, k, "separate:var:X:sum_square")
    X = ir.variable(n, k, "var:X")
    f0 = ir.prox(ProxFunction.SUM_SQUARE,
                 ir.add(ir.linear_map(ir.left_matrix_product(ir.dense_matrix(A), k), ir.reshape(Xp, n * k, 1)),
                        ir.linear_map(ir.scalar(-1, m * k), ir.constant(B.reshape(-1, 1, order="F")))),
                 alpha=1.0, arg_size=[(m * k, 1)])
    f1 = ir.prox(ProxFunction.NORM_1, X, alpha=lam)
    c = ir.zero(ir.add(ir.reshape(Xp, n * k, 1), ir.linear_map(ir.scalar(-1, n * k), ir.reshape(X, n * k, 1))))
    return ir.Problem([f0, f1], [c])


def mv_lasso(m, n, k, rho=0.01, sigma=0.05, seed=0):
    """Multivariate lasso (reference problems/lasso.py with k > 1, benchmark.py:46: m=1500,
    n=5000, k=10, rho=0.01): A with unit-l2 columns, X0 with support density rho,
    B = A X0 + sigma randn, lam = 0.5 max|A^T B|."""
    rng = np.random.RandomState(seed)
    A = rng.randn(m, n)
    A /= np.sqrt(np.sum(A ** 2, 0))
    X0 = np.zeros((n, k))
    nnz = int(round(rho * n * k))
    idx = rng.choice(n * k, nnz, replace=False)
    X0.reshape(-1)[idx] = rng.randn(nnz)
    B = A.dot(X0) + sigma * rng.randn(m, k)
    lam = 0.5 * np.abs(A.T.dot(B)).max()
    return mv_lasso_ir(A, B, lam), dict(A=A, B=B, lam=lam)


def mv_lasso_objective(A, B, lam, X):
    return float(np.sum((A.dot(X) - B) ** 2) + lam * np.abs(X).sum())


def mnist_random_features(X, n, rng):
    """cos(Xp W + B) kitchen-sink features of reference problems/mnist.py:27-44: Xp = the data
    projected on 50 eigenvectors of its scatter matrix when it has more rows than columns (the
    reference takes the FIRST 50 columns of eigh's output - the smallest eigenvalues - and so does
    this), sigma = the median distance of m^1.5 random pairs."""
    X = np.asarray(X, dtype=np.float64)
    if X.shape[0] >= X.shape[1]:
        Xc = X - np.mean(X, axis=0)
        _, D = np.linalg.eigh(Xc.T.dot(Xc))
        Xp = X.dot(D[:, :50])
    else:
        Xp = X
    mrows = Xp.shape[0]
    kk = int(mrows ** 1.5)
    I, J = rng.randint(0, mrows, kk), rng.randint(0, mrows, kk)
    sigma = np.sort(np.linalg.norm(Xp[I] - Xp[J], axis=1))[kk // 2]
    W = rng.randn(Xp.shape[1], n) / sigma / np.sqrt(2)
    Bv = rng.uniform(0, 2 * np.pi, n)
    return np.cos(Xp.dot(W) + Bv)


def mnist_features_lasso(X, y, n=1000, lam=0.1, seed=0):
    """The reference's "mnist" benchmark problem (problems/mnist.py:51-64, benchmark.py:45 on the
    2000-sample mnist_small data): sum_squares(F Theta - Y) + 0.1 norm_1(Theta), F = n random
    features, Y the one-hot labels, Theta n x 10."""
    rng = np.random.RandomState(seed)
    F = mnist_random_features(X, n, rng)
    y = np.asarray(y).ravel().astype(int)
    Y = np.zeros((len(y), int(y.max()) + 1))
    Y[np.arange(len(y)), y] = 1.0
    return mv_lasso_ir(F, Y, lam), dict(A=F, B=Y, lam=lam)


def fused_lasso(m, ni, k, rho=0.05, sigma=0.05, seed=0):
    """sum_squares(A x - b) + lam norm_1(x) + lam tv(x) (reference problems/fused_lasso.py,
    benchmark.py:30: m=1000, ni=10, k=1000), in prox form: the least-squares term and the l1 term
    work on copies tied to x by equality constraints, the total variation keeps x itself."""
    rng = np.random.RandomState(seed)
    n = ni * k
    A = rng.randn(m, n)
    A /= np.sqrt(np.sum(A ** 2, 0))
    x0 = np.zeros(n)
    for i in range(k):
        if rng.rand() < rho:
            x0[i * ni:(i + 1) * ni] = rng.rand()
    b = A.dot(x0) + sigma * rng.randn(m)
    lam = 0.1 * sigma * np.sqrt(m * np.log(n))
    xs = ir.variable(n, 1, "separate:var:x:sum_square")
    xl = ir.variable(n, 1, "separate:var:x:norm_1")
    x = ir.variable(n, 1, "var:x")
    f0 = ir.prox(ProxFunction.SUM_SQUARE,
                 ir.add(ir.linear_map(ir.dense_matrix(A), xs), ir.linear_map(ir.scalar(-1, m), ir.constant(b))),
                 alpha=1.0)
    f1 = ir.prox(ProxFunction.NORM_1, xl, alpha=lam)
    f2 = ir.prox(ProxFunction.TOTAL_VARIATION_1D, x, alpha=lam)
    c0 = ir.zero(ir.add(xs, ir.linear_map(ir.scalar(-1, n), x)))
    c1 = ir.zero(ir.add(xl, ir.linear_map(ir.scalar(-1, n), x)))
    return ir.Problem([f0, f1, f2], [c0, c1]), dict(A=A, b=b, lam=lam)


def fused_lasso_objective(A, b, lam, x):
    return float(np.sum((A.dot(x) - b) ** 2) + lam * np.abs(x).sum() + lam * np.abs(np.diff(x)).sum())


def logreg_l1(m, n, lam=None, seed=0):
    """sum_i logistic(-y_i a_i^T x) + lam ||x||_1 in graph form
    (reference python/epopt/problems/logreg_l1.py, compiled like docs/notebooks):

      sum_logistic(z) + norm_1(x)[lam] + zero(C x' - z')   s.t.  x' - x = 0,  z' - z = 0
    with C = -diag(y) A."""
    rng = np.random.RandomState(seed)
    A = rng.randn(m, n)
    x0 = rng.randn(n) * (rng.rand(n) < 0.3)
    y = np.sign(A.dot(x0) + 0.1 * rng.randn(m))
    C = -y[:, None] * A
    if lam is None:
        lam = 0.1 * np.abs(C.T.dot(np.full(m, 0.5))).max()
    z = ir.variable(m, 1, "var:z")
    zp = ir.variable(m, 1, "separate:var:z:zero")
    x = ir.variable(n, 1, "var:x")
    xp = ir.variable(n, 1, "separate:var:x:zero")
    f0 = ir.prox(ProxFunction.SUM_LOGISTIC, z, alpha=1.0)
    f1 = ir.prox(ProxFunction.NORM_1, x, alpha=lam)
    f2 = ir.prox(ProxFunction.ZERO,
                 ir.add(ir.linear_map(ir.dense_matrix(C), xp), ir.linear_map(ir.scalar(-1, m), zp)))
    c0 = ir.zero(ir.add(xp, ir.linear_map(ir.scalar(-1, n), x)))
    c1 = ir.zero(ir.add(zp, ir.linear_map(ir.scalar(-1, m), z)))
    return ir.Problem([f0, f1, f2], [c0, c1]), dict(C=C, lam=lam)


def logreg_l1_objective(C, lam, x):
    return float(np.sum(np.logaddexp(0, C.dot(x))) + lam * np.abs(x).sum())


def covsel(n, m=None, lam=0.1, seed=0):
    """Sparse inverse covariance selection, -log_det(X) + <S, X'> + lam ||X''||_1
    (reference python/epopt/problems/covsel.py) in consensus form:

      neg_log_det(X) + affine(<S, X1>) + norm_1(X2)[lam]   s.t.  X1 - X = 0,  X2 - X = 0"""
    rng = np.random.RandomState(seed)
    m = m or 10 * n
    P = np.eye(n) + 0.3 * sp.rand(n, n, 0.2, random_state=rng).toarray()
    P = (P + P.T) / 2 + n * 0.05 * np.eye(n)
    Z = rng.multivariate_normal(np.zeros(n), np.linalg.inv(P), size=m)
    S = Z.T.dot(Z) / m
    X = ir.variable(n, n, "var:X")
    X1 = ir.variable(n * n, 1, "separate:var:X:affine")
    X2 = ir.variable(n * n, 1, "separate:var:X:norm_1")
    f0 = ir.prox(ProxFunction.NEG_LOG_DET, X, alpha=1.0)
    f1 = ir.prox(ProxFunction.AFFINE,
                 ir.linear_map(ir.dense_matrix(S.reshape(1, -1, order="F")), X1), alpha=1.0)
    f2 = ir.prox(ProxFunction.NORM_1, X2, alpha=lam)
    Xv = ir.reshape(X, n * n, 1)
    c0 = ir.zero(ir.add(X1, ir.linear_map(ir.scalar(-1, n * n), Xv)))
    c1 = ir.zero(ir.add(X2, ir.linear_map(ir.scalar(-1, n * n), Xv)))
    return ir.Problem([f0, f1, f2], [c0, c1]), dict(S=S, lam=lam)


def covsel_objective(S, lam, X):
    sign, logdet = np.linalg.slogdet(X)
    return float(-logdet + np.sum(S * X) + lam * np.abs(X).sum()) if sign > 0 else float("inf")


# ---- LP-representable benchmark problems in graph form (reference python/epopt/problems/
#      basis_pursuit.py, least_abs_dev.py, lp.py, hinge_l1.py, quantile.py) -------------------------
# Each data-dependent affine relation z = C x (+ d) is one ZERO term over private copies
# (x', z'), tied to the variables of the separable terms by consensus constraints - the shape
# the reference's compiler produces (compiler/transforms/separate.py).


def _graph_form(f_terms, C, d, x_key="var:x", z_key="var:z", x_in_terms=True):
    """terms(x, z) + zero(C x' - z' + d)  s.t.  x' - x = 0, z' - z = 0  (d may be None).  When no
    separable term touches x (`x_in_terms=False`) it lives in the ZERO term alone, uncopied."""
    m, n = C.shape
    x = ir.variable(n, 1, x_key)
    z = ir.variable(m, 1, z_key)
    xp = ir.variable(n, 1, "separate:%s:zero" % x_key) if x_in_terms else x
    zp = ir.variable(m, 1, "separate:%s:zero" % z_key)
    parts = [ir.linear_map(ir.dense_matrix(C), xp), ir.linear_map(ir.scalar(-1, m), zp)]
    if d is not None:
        parts.append(ir.constant(np.asarray(d, dtype=np.float64)))
    terms = list(f_terms(x, z)) + [ir.prox(ProxFunction.ZERO, ir.add(*parts))]
    cons = [ir.zero(ir.add(zp, ir.linear_map(ir.scalar(-1, m), z)))]
    if x_in_terms:
        cons.insert(0, ir.zero(ir.add(xp, ir.linear_map(ir.scalar(-1, n), x))))
    return ir.Problem(terms, cons)


def basis_pursuit(m, n, seed=0):
    """min ||x||_1  s.t.  A x = b."""
    rng = np.random.RandomState(seed)
    A = rng.randn(m, n)
    b = A.dot(rng.randn(n) * (rng.rand(n) < 0.2))
    xp = ir.variable(n, 1, "separate:var:x:zero")
    x = ir.variable(n, 1, "var:x")
    f0 = ir.prox(ProxFunction.NORM_1, x)
    f1 = ir.prox(ProxFunction.ZERO, ir.add(ir.linear_map(ir.dense_matrix(A), xp),
                                           ir.linear_map(ir.scalar(-1, m), ir.constant(b))))
    c = ir.zero(ir.add(xp, ir.linear_map(ir.scalar(-1, n), x)))
    return ir.Problem([f0, f1], [c]), dict(A=A, b=b)


def least_abs_dev(m, n, seed=0):
    """min ||A x - b||_1: norm_1(z) with z = A x - b."""
    rng = np.random.RandomState(seed)
    A = rng.randn(m, n)
    b = A.dot(rng.randn(n)) + rng.laplace(size=m)
    prob = _graph_form(lambda x, z: [ir.prox(ProxFunction.NORM_1, z)], A, -b, x_in_terms=False)
    return prob, dict(A=A, b=b)


def hinge_l1(m, n, lam=None, seed=0):
    """sum_i max(0, 1 - y_i a_i^T x) + lam ||x||_1: sum_hinge(1 - z) with z = diag(y) A x."""
    rng = np.random.RandomState(seed)
    A = rng.randn(m, n)
    y = np.sign(A.dot(rng.randn(n) * (rng.rand(n) < 0.3)) + 0.1 * rng.randn(m))
    C = y[:, None] * A
    if lam is None:
        lam = 0.1 * np.abs(C.sum(axis=0)).max()

    def terms(x, z):
        arg = ir.add(ir.linear_map(ir.scalar(-1, m), z), ir.scalar_constant(1.0, (m, 1)))
        return [ir.prox(ProxFunction.SUM_HINGE, arg), ir.prox(ProxFunction.NORM_1, x, alpha=lam)]
    return _graph_form(terms, C, None), dict(C=C, lam=lam)


def quantile(m, n, tau=0.3, seed=0):
    """Quantile regression sum_i rho_tau(b_i - a_i^T x): sum_quantile(z), z = A x - b, with
    alpha = 1 - tau on the positive and beta = tau on the negative part of z."""
    rng = np.random.RandomState(seed)
    A = np.hstack([np.ones((m, 1)), rng.randn(m, n - 1)])
    b = A.dot(rng.randn(n)) + rng.randn(m) * (1 + np.abs(A[:, 1]))
    qa, qb = ir.scalar_constant(1.0 - tau, (1, 1)), ir.scalar_constant(tau, (1, 1))

    def terms(x, z):
        return [ir.prox(ProxFunction.SUM_QUANTILE, z,
                        scaled_zone_params=wire.ProxScaledZoneParams(alpha_expr=qa.proto,
                                                                     beta_expr=qb.proto))]
    return _graph_form(terms, A, -b, x_in_terms=False), dict(A=A, b=b, tau=tau)


def lp(m, n, seed=0):
    """min c^T x  s.t.  A x <= b  (bounded, feasible):  affine(c^T x) + non_negative(s) with
    s = b - A x."""
    rng = np.random.RandomState(seed)
    A = rng.randn(m, n)
    x0 = rng.randn(n)
    b = A.dot(x0) + rng.rand(m)
    c = -A.T.dot(rng.rand(m))  # c in -cone(A^T): the LP is bounded

    def terms(x, s):
        return [ir.prox(ProxFunction.AFFINE, ir.linear_map(ir.dense_matrix(c.reshape(1, -1)), x)),
                ir.prox(ProxFunction.NON_NEGATIVE, s)]
    return _graph_form(terms, -A, b, z_key="var:s"), dict(A=A, b=b, c=c)


# ---- the reference's solver-level known answers (python/epopt/constant_atoms_test.py) ------------

def constant_atom(prox_name, arg_columns, k=None, alpha=None, beta=None, arg_scale=None, linear=None, axis=None):
    """minimise f(x)  s.t.  x - c = 0  for ONE prox function f of this path: the hand-compiled form
    of the reference's "atoms with variable arguments" test (constant_atoms_test.py:283-292: one
    variable per argument, tied to the constant by an equality constraint).  `arg_columns` is the
    constant as CVXPY reads a literal: a list of columns.  Returns (problem, c) with c the m x n
    constant."""
    c = np.array(arg_columns, dtype=np.float64).T
    if c.ndim == 1:
        c = c.reshape(-1, 1)
    m, n = c.shape
    x = ir.variable(m, n, "var:x")
    kw = {}
    if k is not None:
        kw["sum_largest_params"] = wire.SumLargestParams(k=int(k))
    if alpha is not None:
        a, b = ir.constant(np.full(m * n, float(alpha))), ir.constant(np.full(m * n, float(beta)))
        data = dict(a.data)
        data.update(b.data)
        kw["scaled_zone_params"] = wire.ProxScaledZoneParams(alpha_expr=a.proto, beta_expr=b.proto)
        kw["data"] = data
    if axis is not None:  # one value per column (axis 0) / row (axis 1), summed
        kw["has_axis"] = True
        kw["axis"] = int(axis)
    arg = x
    if arg_scale is not None:  # f(arg_scale * x): neg(x) = pos(-x)
        arg = ir.linear_map(ir.scalar(float(arg_scale), m * n), ir.reshape(x, m * n, 1))
    if linear is not None:     # the AFFINE prox of a linear functional l^T vec(x) (sum_entries, trace)
        arg = ir.linear_map(ir.dense_matrix(np.asarray(linear, dtype=np.float64).reshape(1, -1)),
                            ir.reshape(x, m * n, 1))
    if arg_scale is not None:
        kw["arg_size"] = [(m, n)]  # (the scaled argument keeps the variable's shape)
    f = ir.prox(getattr(ProxFunction, prox_name), arg, alpha=1.0, **kw)
    con = ir.zero(ir.add(x, ir.linear_map(ir.scalar(-1, m * n), ir.constant(c.reshape(-1, 1, order="F")))))
    return ir.Problem([f], [con]), c


def constant_atom_value(prox_name, X, k=None, alpha=None, beta=None, arg_scale=None, linear=None, axis=None):
    """f(X) in numpy, for the objective the reference's test evaluates at the returned variable."""
    X = np.asarray(X, dtype=np.float64)
    if arg_scale is not None:
        X = arg_scale * X
    v = X.reshape(-1, order="F")
    if prox_name == "AFFINE":
        return float(np.dot(np.asarray(linear, dtype=np.float64), v))
    if axis is not None and prox_name == "MAX":
        return float(X.max(axis=axis).sum())
    if axis is not None and prox_name == "NORM_2":
        return float(np.sqrt((X ** 2).sum(axis=axis)).sum())
    if prox_name == "NORM_1":
        return float(np.abs(v).sum())
    if prox_name == "NORM_2":
        return float(np.sqrt((v ** 2).sum()))
    if prox_name == "NORM_NUCLEAR":
        return float(np.linalg.svd(X, compute_uv=False).sum())
    if prox_name == "LAMBDA_MAX":
        return float(np.linalg.eigvalsh(0.5 * (X + X.T)).max())
    if prox_name == "SUM_SQUARE":
        return float((v ** 2).sum())
    if prox_name == "TOTAL_VARIATION_1D":
        return float(np.abs(np.diff(v)).sum())
    if prox_name == "MAX":
        return float(v.max())
    if prox_name == "SUM_LARGEST":
        return float(np.sort(v)[::-1][:k].sum())
    if prox_name == "LOG_SUM_EXP":
        return float(np.log(np.exp(v - v.max()).sum()) + v.max())
    if prox_name == "SUM_EXP":
        return float(np.exp(v).sum())
    if prox_name == "SUM_INV_POS":
        return float((1.0 / v).sum())
    if prox_name == "SUM_HINGE":
        return float(np.maximum(v, 0).sum())
    if prox_name == "SUM_QUANTILE":
        return float((alpha * np.maximum(v, 0) + beta * np.maximum(-v, 0)).sum())
    if prox_name == "SUM_NEG_LOG":
        return float(-np.log(v).sum())
    if prox_name == "NEG_LOG_DET":
        return float(-np.linalg.slogdet(0.5 * (X + X.T))[1])
    raise ValueError(prox_name)
