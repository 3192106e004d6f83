"""Pins the oracle's proximal operators the way the reference's python tests do
(python/epopt/prox_test.py:250-266: eval_prox vs an independent solve of
lam*f(x) + 1/2||x - v||^2, rtol = atol = 1e-2), with scipy / closed forms standing in for
CVXPY, at a much tighter tolerance."""

import numpy as np
import pytest
from scipy import optimize

from epsilon_amd import ir, wire
from epsilon_amd.wire import ProxFunction
from oracle import epsilon_oracle as orc

N = 10


def eval_prox(expr, lam, v_map):
    vb = {k: np.asarray(v, dtype=np.float64).tobytes() for k, v in v_map.items()}
    out = orc.eval_prox(expr.proto.SerializeToString(), lam, expr.data, vb)
    return {k: np.frombuffer(b) for k, b in out.items()}


def numeric_prox(f, v, lam, smooth=False):
    """argmin lam*f(x) + 0.5||x-v||^2 by a derivative-free / quasi-Newton polish from many
    starts (small n)."""
    obj = lambda x: lam * f(x) + 0.5 * np.sum((x - v) ** 2)
    best = None
    for x0 in (v, np.zeros_like(v)):
        r = optimize.minimize(obj, x0, method="Powell", options=dict(xtol=1e-10, ftol=1e-14, maxiter=200000, maxfev=400000))
        if best is None or r.fun < best.fun:
            best = r
    return best.x, best.fun


@pytest.mark.parametrize("trial", range(5))
def test_norm1_closed_form(trial):
    rng = np.random.RandomState(trial)
    v, lam = rng.randn(N), abs(rng.randn())
    x = ir.variable(N, 1, "var:x")
    got = eval_prox(ir.prox(ProxFunction.NORM_1, x), lam, {"var:x": v})["var:x"]
    np.testing.assert_allclose(got, np.sign(v) * np.maximum(np.abs(v) - lam, 0), atol=1e-12)


@pytest.mark.parametrize("trial", range(3))
def test_weighted_norm1_with_zero_weight(trial):  # prox_test.py:86-89
    rng = np.random.RandomState(trial)
    v, lam = rng.randn(N), abs(rng.randn()) + 0.1
    w = rng.randn(N)
    w[0] = 0
    x = ir.variable(N, 1, "var:x")
    got = eval_prox(ir.prox(ProxFunction.NORM_1, ir.linear_map(ir.diagonal_matrix(w), x)), lam,
                    {"var:x": v})["var:x"]
    want = np.sign(v) * np.maximum(np.abs(v) - lam * np.abs(w), 0)
    np.testing.assert_allclose(got, want, atol=1e-12)


@pytest.mark.parametrize("kind", ["deadzone", "hinge", "hinge_1mx", "quantile"])
def test_scaled_zone_family(kind):  # prox_test.py:60-75,195-205
    rng = np.random.RandomState(4)
    v, lam = rng.randn(N), 0.7
    x = ir.variable(N, 1, "var:x")
    if kind == "deadzone":
        eps = 0.4
        e = ir.prox(ProxFunction.SUM_DEADZONE, x, scaled_zone_params=wire.ProxScaledZoneParams(m=eps))
        f = lambda z: np.sum(np.maximum(np.abs(z) - eps, 0))
    elif kind == "hinge":
        e = ir.prox(ProxFunction.SUM_HINGE, x)
        f = lambda z: np.sum(np.maximum(z, 0))
    elif kind == "hinge_1mx":
        e = ir.prox(ProxFunction.SUM_HINGE,
                    ir.add(ir.linear_map(ir.scalar(-1, N), x), ir.scalar_constant(1.0, (N, 1))))
        f = lambda z: np.sum(np.maximum(1 - z, 0))
    else:
        a, b = np.abs(rng.randn(N)), np.abs(rng.randn(N))
        qa, qb = ir.constant(a), ir.constant(b)
        data = dict(qa.data)
        data.update(qb.data)
        e = ir.prox(ProxFunction.SUM_QUANTILE, x, data=data,
                    scaled_zone_params=wire.ProxScaledZoneParams(alpha_expr=qa.proto, beta_expr=qb.proto))
        f = lambda z: np.sum(a * np.maximum(z, 0) + b * np.maximum(-z, 0))
    got = eval_prox(e, lam, {"var:x": v})["var:x"]
    # separable: solve each coordinate on a fine grid + golden section
    want = np.empty(N)
    for i in range(N):
        fi = lambda t, i=i: lam * f(np.where(np.arange(N) == i, t, 0.0)) - lam * f(np.zeros(N)) * 0 + 0.5 * (t - v[i]) ** 2
        g = lambda t: lam * (f(np.where(np.arange(N) == i, t, got)) ) + 0.5 * (t - v[i]) ** 2
        r = optimize.minimize_scalar(g, bracket=(v[i] - 5, v[i] + 5), tol=1e-13)
        want[i] = r.x
    np.testing.assert_allclose(got, want, atol=2e-6)


@pytest.mark.parametrize("trial", range(3))
def test_norm2_and_nonneg(trial):
    rng = np.random.RandomState(trial)
    v, lam = rng.randn(N), abs(rng.randn()) + 0.05
    x = ir.variable(N, 1, "var:x")
    got = eval_prox(ir.prox(ProxFunction.NORM_2, x), lam, {"var:x": v})["var:x"]
    nv = np.linalg.norm(v)
    np.testing.assert_allclose(got, max(0, 1 - lam / nv) * v, atol=1e-12)
    got = eval_prox(ir.prox(ProxFunction.NON_NEGATIVE, ir.linear_map(ir.scalar(-1.3, N), x)), lam,
                    {"var:x": v})["var:x"]
    np.testing.assert_allclose(got, np.minimum(v, 0), atol=1e-12)  # -1.3 x >= 0  <=>  x <= 0


@pytest.mark.parametrize("m", [20, 5])
def test_sum_square_least_squares(m):  # prox_test.py:76-79,206-207
    rng = np.random.RandomState(m)
    A, b, v, lam = rng.randn(m, N), rng.randn(m), rng.randn(N), 0.8
    x = ir.variable(N, 1, "var:x")
    e = ir.prox(ProxFunction.SUM_SQUARE, ir.add(ir.linear_map(ir.dense_matrix(A), x),
                                                ir.linear_map(ir.scalar(-1, m), ir.constant(b))))
    got = eval_prox(e, lam, {"var:x": v})["var:x"]
    want = np.linalg.solve(2 * lam * A.T @ A + np.eye(N), 2 * lam * A.T @ b + v)
    np.testing.assert_allclose(got, want, atol=1e-10)


def test_zero_projection():  # prox_test.py:210-220 (C_linear_equality)
    rng = np.random.RandomState(7)
    A = rng.randn(5, N)
    b = A @ rng.randn(N)
    v = rng.randn(N)
    x = ir.variable(N, 1, "var:x")
    e = ir.prox(ProxFunction.ZERO, ir.add(ir.linear_map(ir.dense_matrix(A), x),
                                          ir.linear_map(ir.scalar(-1, 5), ir.constant(b))))
    got = eval_prox(e, 1.3, {"var:x": v})["var:x"]
    want = v - A.T @ np.linalg.solve(A @ A.T, A @ v - b)
    np.testing.assert_allclose(got, want, atol=1e-10)


def test_affine():
    rng = np.random.RandomState(8)
    c, v, lam = rng.randn(1, N), rng.randn(N), 0.6
    x = ir.variable(N, 1, "var:x")
    got = eval_prox(ir.prox(ProxFunction.AFFINE, ir.linear_map(ir.dense_matrix(c), x)), lam, {"var:x": v})["var:x"]
    np.testing.assert_allclose(got, v - lam * c.ravel(), atol=1e-12)


def test_tv1d_known_answer_and_certificate():
    """n=2, v=(0,10), lam=1 -> x=(1,9); plus the KKT certificate and an
    independent dual solve (min 1/2||D^T z - v||^2, |z| <= lam) on random inputs."""
    np.testing.assert_allclose(orc.tv1d_prox(np.array([0.0, 10.0]), 1.0), [1.0, 9.0])
    for seed in range(4):
        rng = np.random.RandomState(seed)
        n = 60
        v = np.cumsum(rng.randn(n)) * 0.5 + rng.randn(n)
        lam = 0.3 + seed
        x = orc.tv1d_prox(v, lam)
        bound, jump, end = orc.tv1d_kkt_violation(x, v, lam)
        assert bound < 1e-10 and jump < 1e-10 and end < 1e-10
        D = np.diff(np.eye(n), axis=0)  # (n-1) x n
        r = optimize.lsq_linear(D.T, v, bounds=(-lam, lam), tol=1e-14, max_iter=2000)
        np.testing.assert_allclose(x, v - D.T @ r.x, atol=2e-6)
    x = ir.variable(N, 1, "var:x")
    v = np.random.RandomState(0).randn(N)
    got = eval_prox(ir.prox(ProxFunction.TOTAL_VARIATION_1D, x), 0.5, {"var:x": v})["var:x"]
    np.testing.assert_allclose(got, orc.tv1d_prox(v, 0.5), atol=1e-12)


def test_nuclear_norm():  # prox_test.py:190
    rng = np.random.RandomState(9)
    for (m, n) in [(3, 3), (12, 7)]:
        V, lam = rng.randn(m, n), 0.4
        X = ir.variable(m, n, "var:X")
        got = eval_prox(ir.prox(ProxFunction.NORM_NUCLEAR, X), lam, {"var:X": V.reshape(-1, order="F")})["var:X"]
        U, s, Vt = np.linalg.svd(V, full_matrices=False)
        want = (U * np.maximum(s - lam, 0)) @ Vt
        np.testing.assert_allclose(got.reshape((m, n), order="F"), want, atol=1e-7)


@pytest.mark.parametrize("kind", ["norm_1", "hinge", "sum_square"])
def test_epigraph_projection_vs_constrained_solve(kind):
    """prox_test.py:232-246 epigraph cases: eval_prox of an epigraph operator is the Euclidean
    projection of (v, s) onto {(x, t): f(x) <= t}; checked against SLSQP."""
    n = 8
    rng = np.random.RandomState(3)
    typ, f = {"norm_1": (ProxFunction.NORM_1, lambda z: np.abs(z).sum()),
              "hinge": (ProxFunction.SUM_HINGE, lambda z: np.maximum(z, 0).sum()),
              "sum_square": (ProxFunction.SUM_SQUARE, lambda z: (z * z).sum())}[kind]
    x, t = ir.variable(n, 1, "var:x"), ir.variable(1, 1, "var:t")
    for _ in range(3):
        v, s = rng.randn(n), 0.5 * rng.randn()
        got = eval_prox(ir.prox(typ, x, t, epigraph=True), 1.0, {"var:x": v, "var:t": [s]})
        obj = lambda z: 0.5 * np.sum((z[:n] - v) ** 2) + 0.5 * (z[n] - s) ** 2
        cons = {"type": "ineq", "fun": lambda z: z[n] - f(z[:n])}
        z0 = np.concatenate([0.1 * v, [f(0.1 * v) + 1]])
        r = optimize.minimize(obj, z0, constraints=[cons], method="SLSQP", options=dict(ftol=1e-14, maxiter=500))
        np.testing.assert_allclose(got["var:x"], r.x[:n], atol=2e-4)
        np.testing.assert_allclose(got["var:t"][0], r.x[n], atol=2e-4)
        assert f(got["var:x"]) <= got["var:t"][0] + 1e-9
