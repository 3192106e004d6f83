"""Pins the oracle's dense kernels against oracle/_ref: the reference's vendored Eigen (BLAS and
decompositions) compiled from the reference tree and called the way the reference's solver core
calls it (oracle/ref_driver.cc).  This is the part of the reference that builds in this image
without stand-ins; the solver core itself does not.  The `-m gpu` half holds
the device to the same library."""

import numpy as np
import pytest

from epsilon_amd import ir, problems
from epsilon_amd.wire import ProxFunction
from oracle import epsilon_oracle as orc
from oracle import ref_lib

pytestmark = pytest.mark.skipif(not ref_lib.available(), reason="oracle/_ref/libref.so not built")


def _omap(lmap):
    return orc.build_linear_map(lmap.proto, lmap.data)


def test_dense_apply_and_adjoint_vs_reference_blas():
    rng = np.random.RandomState(0)
    for m, n in [(7, 5), (1, 9), (64, 33), (300, 1)]:
        A, x, y = rng.randn(m, n), rng.randn(n), rng.randn(m)
        M = _omap(ir.dense_matrix(A))
        np.testing.assert_allclose(M.apply(x), ref_lib.dgemv(A, x), rtol=1e-13, atol=1e-13)
        np.testing.assert_allclose(M.T().apply(y), ref_lib.dgemv(A, y, trans=True), rtol=1e-13, atol=1e-13)


def test_dense_times_dense_vs_reference_dgemm():
    rng = np.random.RandomState(1)
    A, B = rng.randn(6, 9), rng.randn(9, 4)
    C = orc.lm_multiply(_omap(ir.dense_matrix(A)), _omap(ir.dense_matrix(B))).as_dense()
    np.testing.assert_allclose(C, ref_lib.dgemm(A, B), rtol=1e-13, atol=1e-13)
    # the Gram product of the least-squares prox: A A^T with the transpose flag on the right
    G = orc.lm_multiply(_omap(ir.dense_matrix(A)), _omap(ir.transpose(ir.dense_matrix(A)))).as_dense()
    np.testing.assert_allclose(G, ref_lib.dgemm(A, A, tb=True), rtol=1e-13, atol=1e-13)


@pytest.mark.parametrize("kind", ["spd", "negative_definite", "indefinite"])
def test_dense_inverse_vs_reference_ldlt(kind):
    rng = np.random.RandomState(2)
    n = 12
    Q = rng.randn(n, n)
    if kind == "spd":
        A = Q.dot(Q.T) + n * np.eye(n)
    elif kind == "negative_definite":
        A = -(Q.dot(Q.T) + n * np.eye(n))
    else:
        A = Q + Q.T  # symmetric, eigenvalues of both signs
        assert (np.linalg.eigvalsh(A) > 0).any() and (np.linalg.eigvalsh(A) < 0).any()
    got = _omap(ir.dense_matrix(A)).inverse().as_dense()
    np.testing.assert_allclose(got, ref_lib.ldlt_inverse(A), rtol=1e-9, atol=1e-10)


def test_block_ldl_solve_vs_reference_llt():
    """the reference's own check of its block factorisation: [[10 I, A12], [A12^T, 10 I]] solved
    against Eigen::LLT (vector/block_cholesky_test.cc:73-104)."""
    rng = np.random.RandomState(3)
    n1, n2 = 5, 4
    A12 = rng.randn(n1, n2)
    Kd = np.block([[10 * np.eye(n1), A12], [A12.T, 10 * np.eye(n2)]])
    b = rng.randn(n1 + n2)
    A = orc.BlockMatrix()
    A.set("0", "0", _omap(ir.scalar(10, n1)))
    A.set("1", "1", _omap(ir.scalar(10, n2)))
    A.set("0", "1", _omap(ir.dense_matrix(A12)))
    A.set("1", "0", _omap(ir.transpose(ir.dense_matrix(A12))))
    rhs = orc.BlockVector()
    rhs.set("0", b[:n1])
    rhs.set("1", b[n1:])
    chol = orc.BlockCholesky()
    chol.compute(A)
    x = chol.solve(rhs)
    got = np.concatenate([x("0"), x("1")])
    np.testing.assert_allclose(got, ref_lib.llt_solve(Kd, b), rtol=1e-10, atol=1e-12)


def test_lasso_sweep_through_reference_blas_matches_the_oracles():
    """ref_lasso_sweeps (the unrolled sweep with its three mat-vecs through the reference tree's
    dgemv_: what bench.py times as the reference's CPU path) against the plain-C restatement and the
    generic numpy oracle's own iterates on the same instance."""
    from oracle import c_oracle
    from epsilon_amd import wire
    prob, info = problems.lasso(40, 90, seed=5)
    A = np.asfortranarray(info["A"], dtype=np.float64)
    b, lam = np.asarray(info["b"], dtype=np.float64).ravel(), float(info["lam"])
    m, n = A.shape
    Minv = np.asfortranarray(ref_lib.ldlt_inverse(np.eye(m) + 2 * ref_lib.dgemm(A, A, tb=True)))
    s_ref, s_c = c_oracle.LassoState(n), c_oracle.LassoState(n)
    ref_lib.lasso_sweeps(A, Minv, b, lam, s_ref, 11)
    c_oracle.lasso_run(A, Minv, b, lam, s_c, 11, abs_tol=0, rel_tol=0)
    for f in ("x0", "x1", "u", "y0", "y1"):
        np.testing.assert_allclose(getattr(s_ref, f), getattr(s_c, f), rtol=0, atol=1e-12, err_msg=f)
    # the generic oracle after the same 11 sweeps (stopping rule off)
    sb = wire.SolverParams(max_iterations=11, abs_tol=0.0, rel_tol=0.0).SerializeToString()
    _, x = orc.solve(prob.SerializeToString(), [], sb, prob.expression_data())
    got = np.frombuffer(x[sorted(x)[0]])
    assert min(np.abs(got - s_ref.x0).max(), np.abs(got - s_ref.x1).max()) < 1e-9


@pytest.mark.parametrize("shape", [(6, 6), (12, 7), (20, 20)])
def test_nuclear_norm_prox_vs_reference_eigensolver(shape):
    """The oracle's nuclear-norm prox against the reference's own route: eigenvectors of
    Y^T Y + 1e-15 I from Eigen's SelfAdjointEigenSolver, U = Y V diag(1/d), soft threshold of d
    (prox/ortho_invariant.cc:36-66 with the NORM_1 eigen prox of :76-105)."""
    rng = np.random.RandomState(4)
    m, n = shape
    Y = rng.randn(m, n)
    lam = 0.8
    X = ir.variable(m, n, "var:X")
    f = ir.prox(ProxFunction.NORM_NUCLEAR, X, alpha=1.0, arg_size=[(m, n)])
    out = orc.eval_prox(f.proto.SerializeToString(), lam, f.data, {"var:X": Y.tobytes(order="F")})
    got = np.frombuffer(out["var:X"]).reshape(m, n, order="F")
    d, V, U = ref_lib.gram_svd(Y)
    want = (U * np.maximum(d - lam, 0)).dot(V.T)
    np.testing.assert_allclose(got, want, rtol=1e-8, atol=1e-8)


# ---- the device against the same library -----------------------------------------------------------


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["spd", "negative_definite", "indefinite"])
def test_device_dense_inverse_vs_reference_ldlt(solve_mod, kind):
    rng = np.random.RandomState(5)
    n = 40
    Q = rng.randn(n, n)
    if kind == "spd":
        A = Q.dot(Q.T) + n * np.eye(n)
    elif kind == "negative_definite":
        A = -(Q.dot(Q.T) + n * np.eye(n))
    else:
        A = Q + Q.T
    solve_mod.set_option("dtype", "f64")
    try:
        got = solve_mod.linear_map_inverse(ir.dense_matrix(A))
    finally:
        solve_mod.set_option("dtype", "f32")
    np.testing.assert_allclose(got, ref_lib.ldlt_inverse(A), rtol=1e-7, atol=1e-9)


@pytest.mark.gpu
def test_device_gemv_gemm_vs_reference_blas(solve_mod):
    rng = np.random.RandomState(6)
    A, x, y = rng.randn(130, 77), rng.randn(77), rng.randn(130)
    B = rng.randn(77, 50)
    solve_mod.set_option("dtype", "f64")
    try:
        np.testing.assert_allclose(solve_mod.linear_map_apply(ir.dense_matrix(A), x), ref_lib.dgemv(A, x),
                                   rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(solve_mod.linear_map_apply(ir.dense_matrix(A), y, transpose=True),
                                   ref_lib.dgemv(A, y, trans=True), rtol=1e-12, atol=1e-12)
        t, C = solve_mod.linear_map_binary("*", ir.dense_matrix(A), ir.dense_matrix(B))
        np.testing.assert_allclose(C, ref_lib.dgemm(A, B), rtol=1e-12, atol=1e-12)
    finally:
        solve_mod.set_option("dtype", "f32")


@pytest.mark.gpu
def test_device_nuclear_norm_prox_vs_reference_eigensolver(solve_mod):
    rng = np.random.RandomState(7)
    m, n = 30, 18
    Y = rng.randn(m, n)
    lam = 0.8
    X = ir.variable(m, n, "var:X")
    f = ir.prox(ProxFunction.NORM_NUCLEAR, X, alpha=1.0, arg_size=[(m, n)])
    solve_mod.set_option("dtype", "f64")
    try:
        out = solve_mod.eval_prox(f.proto.SerializeToString(), lam, f.data, {"var:X": Y.tobytes(order="F")})
    finally:
        solve_mod.set_option("dtype", "f32")
    got = np.frombuffer(out["var:X"]).reshape(m, n, order="F")
    d, V, U = ref_lib.gram_svd(Y)
    np.testing.assert_allclose(got, (U * np.maximum(d - lam, 0)).dot(V.T), rtol=1e-7, atol=1e-8)
