"""Pins the oracle's restatement of the remaining proximal / epigraph operators (SURVEY.md 8(f)
f2) the way python/epopt/prox_test.py:166-246 does - eval_prox against an independent solve of
lam*f(x) + 1/2||x - v||^2 - with scipy / closed forms / optimality conditions standing in for
CVXPY and tolerances far below the reference's 1e-2."""

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


def project_epigraph(f, v, s, x0=None, extra_cons=()):
    n = v.shape[0]
    obj = lambda z: 0.5 * np.sum((z[:n] - v) ** 2) + 0.5 * (z[n] - s) ** 2
    cons = [{"type": "ineq", "fun": lambda z: z[n] - f(z[:n])}] + list(extra_cons)
    if x0 is None:
        x0 = v
    z0 = np.concatenate([x0, [f(x0) + 1]])
    r = optimize.minimize(obj, z0, constraints=cons, method="SLSQP",
                          options=dict(ftol=1e-15, maxiter=1000))
    return r.x[:n], r.x[n]


# ---- elementwise smooth family: optimality condition x + lam f'(x) = v ------------------------

SMOOTH = {
    "sum_exp": (ProxFunction.SUM_EXP, np.exp, lambda x: np.sum(np.exp(x)), False),
    "sum_logistic": (ProxFunction.SUM_LOGISTIC, lambda x: 1 / (1 + np.exp(-x)),
                     lambda x: np.sum(np.log1p(np.exp(x))), False),
    "sum_inv_pos": (ProxFunction.SUM_INV_POS, lambda x: -1 / x ** 2, lambda x: np.sum(1 / x), True),
    "sum_neg_entr": (ProxFunction.SUM_NEG_ENTR, lambda x: 1 + np.log(x),
                     lambda x: np.sum(x * np.log(x)), True),
    "sum_neg_log": (ProxFunction.SUM_NEG_LOG, lambda x: -1 / x, lambda x: -np.sum(np.log(x)), True),
}


@pytest.mark.parametrize("name", sorted(SMOOTH))
@pytest.mark.parametrize("trial", range(3))
def test_smooth_elementwise_prox(name, trial):  # prox_test.py:187,191-194
    typ, grad, _, positive = SMOOTH[name]
    rng = np.random.RandomState(trial)
    v, lam = rng.randn(N), abs(rng.randn()) + 0.05
    x = ir.variable(N, 1, "var:x")
    got = eval_prox(ir.prox(typ, x), lam, {"var:x": v})["var:x"]
    if positive:
        assert np.all(got > 0)
    np.testing.assert_allclose(got + lam * grad(got), v, atol=1e-8)


@pytest.mark.parametrize("name", sorted(SMOOTH))
def test_smooth_elementwise_epigraph(name):  # prox_test.py:228,230,234-238
    typ, _, f, positive = SMOOTH[name]
    rng = np.random.RandomState(11)
    n = 6
    x, t = ir.variable(n, 1, "var:x"), ir.variable(1, 1, "var:t")
    for trial in range(3):
        v, s = rng.randn(n), rng.randn()
        got = eval_prox(ir.prox(typ, x, t, epigraph=True), 1.0, {"var:x": v, "var:t": [s]})
        x0 = np.maximum(v, 0.5) if positive else v
        bounds = [{"type": "ineq", "fun": lambda z: z[:n] - 1e-9}] if positive else []
        wx, wt = project_epigraph(f, v, s, x0=x0, extra_cons=bounds)
        np.testing.assert_allclose(got["var:x"], wx, atol=5e-4)
        np.testing.assert_allclose(got["var:t"][0], wt, atol=5e-4)
        assert f(got["var:x"]) <= got["var:t"][0] + 1e-6


def test_smooth_epigraph_feasible_point_is_fixed():
    x, t = ir.variable(4, 1, "var:x"), ir.variable(1, 1, "var:t")
    v = np.array([0.1, -0.3, 0.2, 0.0])
    for typ in (ProxFunction.SUM_EXP, ProxFunction.SUM_LOGISTIC, ProxFunction.LOG_SUM_EXP,
                ProxFunction.MAX, ProxFunction.SUM_LARGEST):
        kw = dict(sum_largest_params=wire.SumLargestParams(k=2)) if typ == ProxFunction.SUM_LARGEST else {}
        got = eval_prox(ir.prox(typ, x, t, epigraph=True, **kw), 1.0, {"var:x": v, "var:t": [50.0]})
        np.testing.assert_allclose(got["var:x"], v, atol=1e-12)
        np.testing.assert_allclose(got["var:t"], [50.0], atol=1e-12)


@pytest.mark.parametrize("trial", range(4))
def test_kl_div_prox(trial):  # prox_test.py:190
    rng = np.random.RandomState(trial)
    n, lam = 5, abs(rng.randn()) + 0.1
    u, v = rng.randn(n), rng.randn(n)
    p, q = ir.variable(n, 1, "var:p"), ir.variable(n, 1, "var:q")
    got = eval_prox(ir.prox(ProxFunction.SUM_KL_DIV, p, q), lam, {"var:p": u, "var:q": v})
    x, y = got["var:p"], got["var:q"]
    assert np.all(x > 0) and np.all(y > 0)
    # stationarity of lam*(x log(x/y) - x + y) + 1/2 (x-u)^2 + 1/2 (y-v)^2 where the solution is
    # interior; entries pushed to the boundary of the domain (x, y -> 0+) are compared with a
    # bounded solve
    for i in range(n):
        if min(x[i], y[i]) > 1e-6:
            np.testing.assert_allclose(x[i] - u[i] + lam * np.log(x[i] / y[i]), 0, atol=1e-7)
            np.testing.assert_allclose(y[i] - v[i] + lam * (1 - x[i] / y[i]), 0, atol=1e-7)
        else:
            def obj(z):
                kl = z[0] * np.log(z[0] / z[1]) - z[0] + z[1] if z[0] > 0 else z[1]
                return lam * kl + 0.5 * (z[0] - u[i]) ** 2 + 0.5 * (z[1] - v[i]) ** 2
            r = optimize.minimize(obj, [0.5, 0.5], method="L-BFGS-B",
                                  bounds=[(1e-14, None), (1e-14, None)], options=dict(ftol=1e-15, gtol=1e-12))
            assert obj([x[i], y[i]]) <= r.fun + 1e-6


def test_kl_div_epigraph():  # prox_test.py:231
    rng = np.random.RandomState(5)
    p, q, t = ir.variable(1, 1, "var:p"), ir.variable(1, 1, "var:q"), ir.variable(1, 1, "var:t")
    f = lambda z: z[0] * np.log(z[0] / z[1]) - z[0] + z[1]
    for trial in range(4):
        u, v, s = rng.randn() + 1, rng.randn() + 1, -abs(rng.randn())
        got = eval_prox(ir.prox(ProxFunction.SUM_KL_DIV, p, q, t, epigraph=True), 1.0,
                        {"var:p": [u], "var:q": [v], "var:t": [s]})
        z = np.array([got["var:p"][0], got["var:q"][0]])
        # the projection lies on the boundary t = f(x, y): minimise the distance over (x, y) > 0

        def dist(w):
            x, y = np.exp(w)
            tt = max(s, f([x, y]))
            return 0.5 * ((x - u) ** 2 + (y - v) ** 2 + (tt - s) ** 2)
        best = min((optimize.minimize(dist, w0, method="Nelder-Mead",
                                      options=dict(xatol=1e-12, fatol=1e-15, maxiter=20000))
                    for w0 in ([0, 0], [-1, -1], [1, 1], [-3, 0])), key=lambda r: r.fun)
        np.testing.assert_allclose(z, np.exp(best.x), atol=1e-5)
        np.testing.assert_allclose(got["var:t"][0], max(s, f(np.exp(best.x))), atol=1e-5)


def test_exp_epigraph_elementwise():  # prox_test.py:222
    rng = np.random.RandomState(6)
    n = 7
    x, z = ir.variable(n, 1, "var:x"), ir.variable(n, 1, "var:z")
    v, s = rng.randn(n), rng.randn(n)
    got = eval_prox(ir.prox(ProxFunction.EXP, x, z, epigraph=True), 1.0, {"var:x": v, "var:z": s})
    for i in range(n):
        wx, wt = project_epigraph(lambda w: np.exp(w[0]), v[i:i + 1], s[i])
        np.testing.assert_allclose(got["var:x"][i], wx[0], atol=2e-5)
        np.testing.assert_allclose(got["var:z"][i], wt, atol=2e-5)


# ---- sort-based operators -----------------------------------------------------------------------


@pytest.mark.parametrize("trial", range(5))
def test_max_prox_closed_form(trial):  # prox_test.py:171
    rng = np.random.RandomState(trial)
    v, lam = rng.randn(N), abs(rng.randn()) + 0.05
    x = ir.variable(N, 1, "var:x")
    got = eval_prox(ir.prox(ProxFunction.MAX, x), lam, {"var:x": v})["var:x"]
    # Moreau: prox_{lam max}(v) = v - lam * proj_simplex(v / lam)
    u = np.sort(v / lam)[::-1]
    css = np.cumsum(u) - 1
    rho = np.nonzero(u - css / (np.arange(N) + 1) > 0)[0][-1]
    proj = np.maximum(v / lam - css[rho] / (rho + 1), 0)
    np.testing.assert_allclose(got, v - lam * proj, atol=1e-12)


@pytest.mark.parametrize("trial", range(4))
def test_max_epigraph(trial):  # prox_test.py:227
    rng = np.random.RandomState(trial)
    v, s = rng.randn(N), rng.randn() - 0.5
    x, t = ir.variable(N, 1, "var:x"), ir.variable(1, 1, "var:t")
    got = eval_prox(ir.prox(ProxFunction.MAX, x, t, epigraph=True), 1.0, {"var:x": v, "var:t": [s]})
    # KKT: x = min(v, t), t - s = sum (v - t)_+
    tt = got["var:t"][0]
    np.testing.assert_allclose(got["var:x"], np.minimum(v, tt), atol=1e-12)
    np.testing.assert_allclose(tt - s, np.sum(np.maximum(v - tt, 0)), atol=1e-12)


def sum_largest(z, k):
    return np.sum(np.sort(z)[::-1][:k])


def project_sum_largest_epigraph(v, s, k):
    """Projection onto {sum_largest(x, k) <= t} through the smooth lifting
    k q + sum r <= t, r >= x - q, r >= 0 (max is the case k = 1)."""
    n = v.shape[0]
    # z = [x (n), t, q, r (n)]
    obj = lambda z: 0.5 * np.sum((z[:n] - v) ** 2) + 0.5 * (z[n] - s) ** 2
    jac = lambda z: np.concatenate([z[:n] - v, [z[n] - s, 0.0], np.zeros(n)])
    cons = [{"type": "ineq", "fun": lambda z: z[n] - k * z[n + 1] - np.sum(z[n + 2:])},
            {"type": "ineq", "fun": lambda z: z[n + 2:] - z[:n] + z[n + 1]},
            {"type": "ineq", "fun": lambda z: z[n + 2:]}]
    q0 = np.sort(v)[::-1][k - 1]
    r0 = np.maximum(v - q0, 0)
    z0 = np.concatenate([v, [k * q0 + r0.sum() + 1, q0], r0])
    r = optimize.minimize(obj, z0, jac=jac, constraints=cons, method="SLSQP",
                          options=dict(ftol=1e-16, maxiter=2000))
    return r.x[:n], r.x[n]


@pytest.mark.parametrize("trial", range(4))
@pytest.mark.parametrize("k", [1, 4, 9])
def test_sum_largest_prox(trial, k):  # prox_test.py:192
    rng = np.random.RandomState(trial)
    v, lam = rng.randn(N), abs(rng.randn()) + 0.05
    x = ir.variable(N, 1, "var:x")
    e = ir.prox(ProxFunction.SUM_LARGEST, x, sum_largest_params=wire.SumLargestParams(k=k))
    got = eval_prox(e, lam, {"var:x": v})["var:x"]
    obj = lambda z: lam * sum_largest(z, k) + 0.5 * np.sum((z - v) ** 2)
    # the prox removes u in {0 <= u <= lam, sum u = k lam}: x = v - u (Moreau, dual of the
    # support function); check optimality by comparing objective values with a polished solve
    r = optimize.minimize(obj, got + 1e-3 * rng.randn(N), method="Powell",
                          options=dict(xtol=1e-12, ftol=1e-15, maxiter=100000))
    assert obj(got) <= r.fun + 1e-9
    u = v - got
    assert np.all(u >= -1e-12) and np.all(u <= lam + 1e-12)
    np.testing.assert_allclose(np.sum(u), k * lam, atol=1e-10)


def test_sum_largest_epigraph():  # prox_test.py:233
    rng = np.random.RandomState(2)
    k = 4
    x, t = ir.variable(N, 1, "var:x"), ir.variable(1, 1, "var:t")
    e = ir.prox(ProxFunction.SUM_LARGEST, x, t, epigraph=True,
                sum_largest_params=wire.SumLargestParams(k=k))
    for trial in range(3):
        v, s = rng.randn(N), rng.randn() - 1
        got = eval_prox(e, 1.0, {"var:x": v, "var:t": [s]})
        wx, wt = project_sum_largest_epigraph(v, s, k)
        # bisection stops at |g| <= 1e-5 (newton.cc:252)
        np.testing.assert_allclose(got["var:x"], wx, atol=2e-4)
        np.testing.assert_allclose(got["var:t"][0], wt, atol=2e-4)


# ---- log-sum-exp --------------------------------------------------------------------------------


def lse(z):
    m = np.max(z)
    return m + np.log(np.sum(np.exp(z - m)))


@pytest.mark.parametrize("trial", range(4))
def test_log_sum_exp_prox(trial):  # prox_test.py:170
    rng = np.random.RandomState(trial)
    v, lam = rng.randn(N), abs(rng.randn()) + 0.05
    x = ir.variable(N, 1, "var:x")
    got = eval_prox(ir.prox(ProxFunction.LOG_SUM_EXP, x), lam, {"var:x": v})["var:x"]
    w = np.exp(got - lse(got))
    np.testing.assert_allclose(got + lam * w, v, atol=1e-8)


@pytest.mark.parametrize("axis", [None, 0, 1])
def test_log_sum_exp_epigraph(axis):  # prox_test.py:223-225
    rng = np.random.RandomState(8)
    if axis is None:
        v, s = rng.randn(N), np.array([rng.randn()])
        x, t = ir.variable(N, 1, "var:x"), ir.variable(1, 1, "var:t")
        got = eval_prox(ir.prox(ProxFunction.LOG_SUM_EXP, x, t, epigraph=True), 1.0,
                        {"var:x": v, "var:t": s})
        wx, wt = project_epigraph(lse, v, s[0])
        np.testing.assert_allclose(got["var:x"], wx, atol=5e-5)
        np.testing.assert_allclose(got["var:t"][0], wt, atol=5e-5)
        return
    m, n = 3, 4
    V = rng.randn(m, n)
    k = n if axis == 0 else m
    s = rng.randn(k)
    X = ir.variable(m, n, "var:X")
    t = ir.variable(1, n, "var:t") if axis == 0 else ir.variable(m, 1, "var:t")
    e = ir.prox(ProxFunction.LOG_SUM_EXP, X, t, epigraph=True, has_axis=True, axis=axis)
    got = eval_prox(e, 1.0, {"var:X": V.reshape(-1, order="F"), "var:t": s})
    GX = got["var:X"].reshape((m, n), order="F")
    for i in range(k):
        vi = V[:, i] if axis == 0 else V[i, :]
        gi = GX[:, i] if axis == 0 else GX[i, :]
        wx, wt = project_epigraph(lse, vi, s[i])
        np.testing.assert_allclose(gi, wx, atol=5e-5)
        np.testing.assert_allclose(got["var:t"][i], wt, atol=5e-5)


# ---- second-order cone ----------------------------------------------------------------------------


def soc_project(v, s):
    nv = np.linalg.norm(v)
    if nv <= s:
        return v, s
    if nv <= -s:
        return np.zeros_like(v), 0.0
    a = (nv + s) / 2
    return a * v / nv, a


def test_second_order_cone_fixed_cases():  # prox_test.py:276-288
    x, t = ir.variable(N, 1, "var:x"), ir.variable(1, 1, "var:t")
    e = ir.prox(ProxFunction.SECOND_ORDER_CONE, t, x, arg_size=[(1, 1), (1, N)])
    cases = [(np.zeros(N), 0.0), (np.arange(N), 100.0), (np.arange(N), 10.0),
             (np.arange(N), -100.0), (np.arange(N), -10.0)]
    for v, s in cases:
        got = eval_prox(e, 1.0, {"var:x": v.astype(float), "var:t": [s]})
        wx, wt = soc_project(v.astype(float), s)
        np.testing.assert_allclose(got["var:x"], wx, atol=1e-12)
        np.testing.assert_allclose(got["var:t"][0], wt, atol=1e-12)


@pytest.mark.parametrize("trial", range(4))
def test_second_order_cone_scaled_translated(trial):  # prox_test.py:155-162,179-181
    rng = np.random.RandomState(trial)
    ax, at, bx, bt = rng.randn(), abs(rng.randn()) + 0.1, rng.randn(), rng.randn()
    v, s = rng.randn(N), rng.randn()
    x, t = ir.variable(N, 1, "var:x"), ir.variable(1, 1, "var:t")
    targ = ir.add(ir.linear_map(ir.scalar(at, 1), t), ir.scalar_constant(bt, (1, 1)))
    xarg = ir.add(ir.linear_map(ir.scalar(ax, N), x), ir.scalar_constant(bx, (N, 1)))
    e = ir.prox(ProxFunction.SECOND_ORDER_CONE, targ, xarg, arg_size=[(1, 1), (1, N)])
    got = eval_prox(e, 1.0, {"var:x": v, "var:t": [s]})
    obj = lambda z: 0.5 * np.sum((z[:N] - v) ** 2) + 0.5 * (z[N] - s) ** 2
    cons = {"type": "ineq", "fun": lambda z: at * z[N] + bt - np.linalg.norm(ax * z[:N] + bx)}
    z0 = np.concatenate([v, [(np.linalg.norm(ax * v + bx) - bt) / at + 1]])
    r = optimize.minimize(obj, z0, constraints=[cons], method="SLSQP", options=dict(ftol=1e-15, maxiter=1000))
    np.testing.assert_allclose(got["var:x"], r.x[:N], atol=2e-5)
    np.testing.assert_allclose(got["var:t"][0], r.x[N], atol=2e-5)


def test_second_order_cone_rows():
    """m cones at once: rows of X against the entries of t (second_order_cone.cc:21-29)."""
    rng = np.random.RandomState(3)
    m, n = 5, 4
    V, s = rng.randn(m, n), rng.randn(m)
    X, t = ir.variable(m, n, "var:X"), ir.variable(m, 1, "var:t")
    e = ir.prox(ProxFunction.SECOND_ORDER_CONE, t, X, arg_size=[(m, 1), (m, n)])
    got = eval_prox(e, 1.0, {"var:X": V.reshape(-1, order="F"), "var:t": s})
    GX = got["var:X"].reshape((m, n), order="F")
    for i in range(m):
        wx, wt = soc_project(V[i], s[i])
        np.testing.assert_allclose(GX[i], wx, atol=1e-12)
        np.testing.assert_allclose(got["var:t"][i], wt, atol=1e-12)


# ---- norm_2 with an axis (group lasso) --------------------------------------------------------------


@pytest.mark.parametrize("axis", [0, 1])
def test_norm2_axis(axis):
    rng = np.random.RandomState(4)
    m, n, lam = 4, 6, 0.9
    V = rng.randn(m, n)
    X = ir.variable(m, n, "var:X")
    got = eval_prox(ir.prox(ProxFunction.NORM_2, X, has_axis=True, axis=axis), lam,
                    {"var:X": V.reshape(-1, order="F")})["var:X"].reshape((m, n), order="F")
    nrm = np.sqrt(np.sum(V * V, axis=axis, keepdims=True))
    np.testing.assert_allclose(got, np.maximum(1 - lam / nrm, 0) * V, atol=1e-12)


# ---- symmetric matrix functions -----------------------------------------------------------------------


def sym(rng, n):
    A = rng.randn(n, n)
    return (A + A.T) / 2


def test_semidefinite_projection():  # prox_test.py:184
    rng = np.random.RandomState(1)
    n = 5
    V = rng.randn(n, n)  # not symmetric: the skew part is handed back (add_residual)
    X = ir.variable(n, n, "var:X")
    got = eval_prox(ir.prox(ProxFunction.SEMIDEFINITE, X), 1.0,
                    {"var:X": V.reshape(-1, order="F")})["var:X"].reshape((n, n), order="F")
    d, Q = np.linalg.eigh((V + V.T) / 2)
    want = (Q * np.maximum(d, 0)) @ Q.T + (V - V.T) / 2
    np.testing.assert_allclose(got, want, atol=1e-10)


def test_neg_log_det_prox():  # prox_test.py:172
    rng = np.random.RandomState(2)
    n, lam = 4, 0.6
    V = sym(rng, n)
    X = ir.variable(n, n, "var:X")
    got = eval_prox(ir.prox(ProxFunction.NEG_LOG_DET, X), lam,
                    {"var:X": V.reshape(-1, order="F")})["var:X"].reshape((n, n), order="F")
    # stationarity: X - lam X^{-1} = V
    np.testing.assert_allclose(got - lam * np.linalg.inv(got), V, atol=1e-9)
    assert np.all(np.linalg.eigvalsh(got) > 0)


def test_lambda_max_prox():  # prox_test.py:169
    rng = np.random.RandomState(3)
    n, lam = 4, 0.5
    V = sym(rng, n)
    X = ir.variable(n, n, "var:X")
    got = eval_prox(ir.prox(ProxFunction.LAMBDA_MAX, X), lam,
                    {"var:X": V.reshape(-1, order="F")})["var:X"].reshape((n, n), order="F")
    d, Q = np.linalg.eigh(V)
    u = np.sort(d / lam)[::-1]
    css = np.cumsum(u) - 1
    rho = np.nonzero(u - css / (np.arange(n) + 1) > 0)[0][-1]
    dd = d - lam * np.maximum(d / lam - css[rho] / (rho + 1), 0)
    np.testing.assert_allclose(got, (Q * dd) @ Q.T, atol=1e-10)


@pytest.mark.parametrize("kind", ["lambda_max", "neg_log_det", "norm_nuclear"])
def test_matrix_epigraphs(kind):  # prox_test.py:221,226,229
    rng = np.random.RandomState(4)
    n = 3
    X, t = ir.variable(n, n, "var:X"), ir.variable(1, 1, "var:t")
    typ = {"lambda_max": ProxFunction.LAMBDA_MAX, "neg_log_det": ProxFunction.NEG_LOG_DET,
           "norm_nuclear": ProxFunction.NORM_NUCLEAR}[kind]
    for trial in range(2):
        V = sym(rng, n) if kind != "norm_nuclear" else rng.randn(n, n)
        s = rng.randn()
        got = eval_prox(ir.prox(typ, X, t, epigraph=True), 1.0,
                        {"var:X": V.reshape(-1, order="F"), "var:t": [s]})
        G = got["var:X"].reshape((n, n), order="F")
        tt = got["var:t"][0]
        # spectral reduction: same eigen/singular vectors, spectrum = vector epigraph projection
        if kind == "norm_nuclear":
            U, d, Vt = np.linalg.svd(V)
            wx, wt = project_epigraph(lambda z: np.abs(z).sum(), d, s)
            want = (U * wx) @ Vt
        else:
            d, Q = np.linalg.eigh(V)
            if kind == "lambda_max":
                wx, wt = project_sum_largest_epigraph(d, s, 1)
            else:
                pos = [{"type": "ineq", "fun": lambda z: z[:n] - 1e-9}]
                wx, wt = project_epigraph(lambda z: -np.sum(np.log(z)), d, s,
                                          x0=np.maximum(d, 0.5), extra_cons=pos)
            want = (Q * wx) @ Q.T
        np.testing.assert_allclose(G, want, atol=5e-4)
        np.testing.assert_allclose(tt, wt, atol=5e-4)
