"""GPU parity for the remaining proximal / epigraph operators: every case
goes through the C ABI (eps_eval_prox) and is compared with the oracle on the same seeded input.

The device algorithms differ from the reference's host ones (thresholds by Newton on the
piecewise-linear equation instead of std::sort; per-element / scalar Newton instead of a global
damped Newton), so agreement is to the tolerance both converge to, stated per test: fp64 mode
1e-8 (the reference stops its Newton iterations at residuals of 1e-10 ... 1e-12), fp32 storage
5e-4 relative.  SUM_LARGEST's epigraph is a bisection to 1e-5 in the reference itself."""

import numpy as np
import pytest

from epsilon_amd import ir, wire
from epsilon_amd.wire import ProxFunction
from oracle import epsilon_oracle as orc
from .test_oracle_prox_more import project_sum_largest_epigraph

pytestmark = pytest.mark.gpu

N = 10


@pytest.fixture(params=["f64", "f32"])
def dtype(request, solve_mod):
    solve_mod.set_option("dtype", request.param)
    yield request.param
    solve_mod.set_option("dtype", "f32")


def tol_for(dtype, f64=1e-8, f32=5e-4):
    return dict(rtol=f64, atol=f64) if dtype == "f64" else dict(rtol=f32, atol=f32)


def run(solve_mod, expr, lam, v_map, tol):
    fb = expr.proto.SerializeToString()
    vb = {k: np.asarray(v, dtype=np.float64).tobytes(order="F") for k, v in v_map.items()}
    got = solve_mod.eval_prox(fb, lam, expr.data, vb)
    want = orc.eval_prox(fb, lam, expr.data, vb)
    assert set(got) == set(want)
    for k in want:
        np.testing.assert_allclose(np.frombuffer(got[k]), np.frombuffer(want[k]), err_msg=k, **tol)
    return {k: np.frombuffer(v) for k, v in got.items()}


def both(solve_mod, expr, lam, v_map):
    fb = expr.proto.SerializeToString()
    vb = {k: np.asarray(v, dtype=np.float64).tobytes(order="F") for k, v in v_map.items()}
    got = {k: np.frombuffer(v) for k, v in solve_mod.eval_prox(fb, lam, expr.data, vb).items()}
    want = {k: np.frombuffer(v) for k, v in orc.eval_prox(fb, lam, expr.data, vb).items()}
    assert set(got) == set(want)
    return got, want


VECTOR_PROX = {
    "max": (ProxFunction.MAX, {}),
    "sum_largest_1": (ProxFunction.SUM_LARGEST, dict(sum_largest_params=wire.SumLargestParams(k=1))),
    "sum_largest_4": (ProxFunction.SUM_LARGEST, dict(sum_largest_params=wire.SumLargestParams(k=4))),
    "sum_largest_9": (ProxFunction.SUM_LARGEST, dict(sum_largest_params=wire.SumLargestParams(k=9))),
    "log_sum_exp": (ProxFunction.LOG_SUM_EXP, {}),
    "sum_exp": (ProxFunction.SUM_EXP, {}),
    "sum_logistic": (ProxFunction.SUM_LOGISTIC, {}),
    "sum_neg_entr": (ProxFunction.SUM_NEG_ENTR, {}),
    "sum_inv_pos": (ProxFunction.SUM_INV_POS, {}),
    "sum_neg_log": (ProxFunction.SUM_NEG_LOG, {}),
}


@pytest.mark.parametrize("name", sorted(VECTOR_PROX))
@pytest.mark.parametrize("n", [N, 1, 700])
def test_vector_prox(solve_mod, dtype, name, n):
    """prox_test.py:170-194 (MAX, SUM_LARGEST, LOG_SUM_EXP, the smooth sums); n = 700 takes the
    256-lane group path, n = 1 the single-lane one."""
    typ, kw = VECTOR_PROX[name]
    for trial in range(3):
        rng = np.random.RandomState(trial)
        x = ir.variable(n, 1, "var:x")
        v, lam = rng.randn(n), abs(rng.randn()) + 0.05
        run(solve_mod, ir.prox(typ, x, **kw), lam, {"var:x": v}, tol_for(dtype))


@pytest.mark.parametrize("name", ["max", "sum_largest_4", "log_sum_exp", "sum_exp"])
def test_vector_prox_scaled_argument(solve_mod, dtype, name):
    """f(a*x + b): VectorProx pre/post scaling (vector_prox.cc:51-70) around the new operators."""
    typ, kw = VECTOR_PROX[name]
    rng = np.random.RandomState(7)
    x = ir.variable(N, 1, "var:x")
    arg = ir.add(ir.linear_map(ir.scalar(-1.7, N), x), ir.scalar_constant(0.3, (N, 1)))
    run(solve_mod, ir.prox(typ, arg, alpha=0.6, **kw), 0.8, {"var:x": rng.randn(N)}, tol_for(dtype))


def test_sum_largest_prox_defining_equation(solve_mod, dtype):
    """x = v - u with 0 <= u <= lam and sum(u) = k lam, also for a lam so large that every entry
    sits inside the window - where the reference's sweep stops early (sum_largest.cc:36-55:
    sum(u) - k lam = -0.059 on this input) and the device result is the correct one."""
    rng = np.random.RandomState(3)
    v = np.abs(rng.randn(N)) + 0.2
    x = ir.variable(N, 1, "var:x")
    e = ir.prox(ProxFunction.SUM_LARGEST, x, sum_largest_params=wire.SumLargestParams(k=4))
    for lam in (0.5, 1.0, 1.5, 2.0, 5.0):
        got, want = both(solve_mod, e, lam, {"var:x": v})
        u = v - got["var:x"]
        tol = 1e-9 if dtype == "f64" else 2e-5
        assert u.min() >= -tol and u.max() <= lam + tol
        np.testing.assert_allclose(u.sum(), 4 * lam, atol=10 * tol)
        if lam <= 1.5:
            np.testing.assert_allclose(got["var:x"], want["var:x"], **tol_for(dtype))


def test_max_prox_ties_and_zero_lambda(solve_mod, dtype):
    x = ir.variable(6, 1, "var:x")
    v = np.array([1.0, 3.0, 3.0, -2.0, 3.0, 0.5])
    got = run(solve_mod, ir.prox(ProxFunction.MAX, x), 0.75, {"var:x": v}, tol_for(dtype))
    np.testing.assert_allclose(got["var:x"], np.minimum(v, 2.75), atol=1e-6)


def test_smooth_prox_elementwise_lambda(solve_mod, dtype):
    """diagonal scaling -> per-element lambda (vector_prox.cc:72-118), incl. a zero weight."""
    rng = np.random.RandomState(3)
    w = rng.rand(N) + 0.5
    x = ir.variable(N, 1, "var:x")
    for typ in (ProxFunction.SUM_EXP, ProxFunction.SUM_LOGISTIC, ProxFunction.SUM_NEG_LOG):
        e = ir.prox(typ, ir.linear_map(ir.diagonal_matrix(w), x))
        run(solve_mod, e, 0.9, {"var:x": rng.randn(N)}, tol_for(dtype))


# Smooth epigraphs: the reference's joint Newton (newton.cc:114-194) builds its step from the
# Schur complement r_x'(I + lam H)^-1 r_x - the residual where the gradient belongs (:147-155) -
# so it is not a Newton direction; with the Armijo search it still descends but often stops at
# its 100-iteration cap with KKT residuals of 1e-4 ... 1e+2 (kkt_norm of the oracle below; the
# reference's own test accepts 1e-2, prox_test.py:258).  The device result is therefore held
# to the KKT conditions themselves (<= 1e-9 in fp64) and compared with the oracle to the
# accuracy the oracle reached, or not at all where it did not converge.
SMOOTH_F = {"sum_exp": orc.SumExp, "sum_logistic": orc.Logistic, "sum_inv_pos": orc.InvPos,
            "sum_neg_entr": orc.NegativeEntropy, "log_sum_exp": orc.LogSumExp}


def kkt_norm(f, x, t, v, s):
    lam = t - s
    r = np.concatenate([x - v + lam * f.gradf(x), [f.eval(x) - t]])
    return float(np.linalg.norm(r)), lam


EPIGRAPHS = {
    "max": (ProxFunction.MAX, {}),
    "sum_largest_4": (ProxFunction.SUM_LARGEST, dict(sum_largest_params=wire.SumLargestParams(k=4))),
    "log_sum_exp": (ProxFunction.LOG_SUM_EXP, {}),
    "sum_exp": (ProxFunction.SUM_EXP, {}),
    "sum_logistic": (ProxFunction.SUM_LOGISTIC, {}),
    "sum_neg_entr": (ProxFunction.SUM_NEG_ENTR, {}),
    "sum_inv_pos": (ProxFunction.SUM_INV_POS, {}),
    "sum_neg_log": (ProxFunction.SUM_NEG_LOG, {}),
}


def check_epigraph(dtype, name, got, want, vx, vs, xkey, tkey, i=None):
    """One projection (slice i of an axis case): parity with the oracle where it converged, KKT
    residual of the device result where a smooth function is involved."""
    gx, gt, wx, wt = got[xkey], got[tkey], want[xkey], want[tkey]
    if name == "sum_largest_4":  # bisection to |g| <= 1e-5 in the reference (newton.cc:252)
        tol = dict(rtol=5e-5, atol=5e-5) if dtype == "f64" else dict(rtol=1e-3, atol=1e-3)
        # The reference's sorted sweep (sum_largest.cc:36-55) leaves its loop when every entry has
        # entered the window, before sum_i clip(v_i - q, 0, lam) = k lam holds; the bisection on
        # top of it then settles on a wrong multiplier (9e-3 off on one of these inputs, inside
        # the reference's own 1e-2 test tolerance).  The independent QP decides.
        qx, qt = project_sum_largest_epigraph(vx, vs, 4)
        qp_tol = dict(rtol=2e-4, atol=2e-4) if dtype == "f64" else dict(rtol=1e-3, atol=1e-3)
        np.testing.assert_allclose(gx, qx, **qp_tol)
        np.testing.assert_allclose(gt[0], qt, **qp_tol)
        if np.max(np.abs(wx - qx)) > 1e-4:
            return "reference inexact"
    elif name in ("max", "sum_neg_log"):
        tol = tol_for(dtype, f64=1e-8, f32=1e-3)
    else:
        tol = tol_for(dtype, f64=1e-5, f32=1e-3)
    if name in SMOOTH_F:
        f = SMOOTH_F[name]()
        interior = name not in ("sum_neg_entr", "sum_inv_pos") or np.all(gx > 1.5e-6)
        r_orc, _ = kkt_norm(f, wx, wt[0], vx, vs)
        if interior and wt[0] > vs:  # active constraint, no clamped entry: KKT must hold
            r_gpu, lam = kkt_norm(f, gx, gt[0], vx, vs)
            assert lam > 0
            assert r_gpu <= (1e-9 if dtype == "f64" else 2e-4) * max(1.0, np.linalg.norm(vx)), \
                "device KKT residual %g (oracle: %g)" % (r_gpu, r_orc)
            if r_orc > 1e-4:
                return "reference did not converge"
            loose = min(1e-2, max(1e-5, 100 * r_orc))  # the oracle is only this close itself
            tol = dict(rtol=max(tol["rtol"], loose), atol=max(tol["atol"], loose))
        elif not interior:
            # entries clamped at 1e-6 (proj_feasible): the reference's global line search stalls on
            # them and leaves the other entries short of their roots; its own test tolerance
            tol = dict(rtol=3e-2, atol=3e-2)  # (python/epopt/prox_test.py:258 accepts 1e-2)
    np.testing.assert_allclose(gx, wx, err_msg=name, **tol)
    np.testing.assert_allclose(gt, wt, err_msg=name, **tol)
    return "compared"


@pytest.mark.parametrize("name", sorted(EPIGRAPHS))
@pytest.mark.parametrize("n", [N, 300])
def test_vector_epigraph(solve_mod, dtype, name, n):
    """prox_test.py:222-238: projection of (v, s) onto {f(x) <= t}; infeasible points, points
    with entries outside the domain, and an already feasible point (s large)."""
    typ, kw = EPIGRAPHS[name]
    x, t = ir.variable(n, 1, "var:x"), ir.variable(1, 1, "var:t")
    e = ir.prox(typ, x, t, epigraph=True, **kw)
    outcomes = []
    for trial in range(4):
        rng = np.random.RandomState(trial)
        v, s = rng.randn(n), rng.randn() - 0.5
        if trial >= 2:
            v = np.abs(v) + 0.2  # inside the domain of the entropy / inverse
        got, want = both(solve_mod, e, 1.0, {"var:x": v, "var:t": [s]})
        outcomes.append(check_epigraph(dtype, name, got, want, v, s, "var:x", "var:t"))
    if name not in ("sum_neg_log",):  # no easy case in the reference (sum_neg_log.cc:42-90)
        v = 0.1 * np.abs(np.random.RandomState(9).randn(n)) + 0.5
        got, want = both(solve_mod, e, 1.0, {"var:x": v, "var:t": [1e4]})
        np.testing.assert_allclose(got["var:x"], want["var:x"], **tol_for(dtype, f64=1e-12, f32=1e-6))
        np.testing.assert_allclose(got["var:t"], [1e4], rtol=1e-6)


@pytest.mark.parametrize("axis", [0, 1])
@pytest.mark.parametrize("name", ["max", "log_sum_exp", "sum_exp"])
def test_epigraph_with_axis(solve_mod, dtype, name, axis):
    """prox_test.py:224-225: one projection per column (axis 0) / row (axis 1)."""
    typ, kw = EPIGRAPHS[name]
    rng = np.random.RandomState(5)
    m, n = 7, 5
    X = ir.variable(m, n, "var:X")
    t = ir.variable(1, n, "var:t") if axis == 0 else ir.variable(m, 1, "var:t")
    e = ir.prox(typ, X, t, epigraph=True, has_axis=True, axis=axis, **kw)
    k = n if axis == 0 else m
    V, s = rng.randn(m, n), rng.randn(k)
    got, want = both(solve_mod, e, 1.0, {"var:X": V.reshape(-1, order="F"), "var:t": s})
    GX, WX = (z["var:X"].reshape((m, n), order="F") for z in (got, want))
    for i in range(k):
        sl = (slice(None), i) if axis == 0 else (i, slice(None))
        check_epigraph(dtype, name, {"x": GX[sl], "t": got["var:t"][i:i + 1]},
                       {"x": WX[sl], "t": want["var:t"][i:i + 1]}, V[sl], s[i], "x", "t")


@pytest.mark.parametrize("axis", [0, 1])
@pytest.mark.parametrize("kind", ["hinge", "norm_1", "deadzone"])
def test_scaled_zone_epigraph_with_axis(solve_mod, dtype, kind, axis):
    """prox_test.py:235-236 (f_hinge_axis0/1 <= t_rvec / t_vec)."""
    rng = np.random.RandomState(8)
    m, n = 6, 9
    X = ir.variable(m, n, "var:X")
    t = ir.variable(1, n, "var:t") if axis == 0 else ir.variable(m, 1, "var:t")
    typ, kw = {"hinge": (ProxFunction.SUM_HINGE, {}), "norm_1": (ProxFunction.NORM_1, {}),
               "deadzone": (ProxFunction.SUM_DEADZONE,
                            dict(scaled_zone_params=wire.ProxScaledZoneParams(m=0.3)))}[kind]
    e = ir.prox(typ, X, t, epigraph=True, has_axis=True, axis=axis, **kw)
    k = n if axis == 0 else m
    run(solve_mod, e, 1.0, {"var:X": rng.randn(m * n), "var:t": 0.5 * rng.randn(k)}, tol_for(dtype))


@pytest.mark.parametrize("axis", [0, 1])
@pytest.mark.parametrize("shape", [(4, 6), (2000, 3), (3, 1500)])
def test_norm2_axis_group_lasso(solve_mod, dtype, axis, shape):
    """NORM_2 over rows / columns (group lasso); tall and wide shapes take the one-lane-per-row
    and the group-per-segment paths."""
    rng = np.random.RandomState(4)
    m, n = shape
    X = ir.variable(m, n, "var:X")
    e = ir.prox(ProxFunction.NORM_2, X, has_axis=True, axis=axis)
    run(solve_mod, e, 0.9, {"var:X": rng.randn(m * n)}, tol_for(dtype))


@pytest.mark.parametrize("axis", [0, 1])
def test_max_prox_axis(solve_mod, dtype, axis):
    rng = np.random.RandomState(6)
    m, n = 9, 1100
    X = ir.variable(m, n, "var:X")
    e = ir.prox(ProxFunction.MAX, X, has_axis=True, axis=axis)
    run(solve_mod, e, 0.7, {"var:X": rng.randn(m * n)}, tol_for(dtype))


def test_kl_div(solve_mod, dtype):  # prox_test.py:190,231
    rng = np.random.RandomState(1)
    n = 40
    p, q, t = ir.variable(n, 1, "var:p"), ir.variable(n, 1, "var:q"), ir.variable(1, 1, "var:t")
    for trial in range(3):
        u, v = rng.randn(n) + 0.5, rng.randn(n) + 0.5
        run(solve_mod, ir.prox(ProxFunction.SUM_KL_DIV, p, q), abs(rng.randn()) + 0.1,
            {"var:p": u, "var:q": v}, tol_for(dtype, f64=1e-9))
    p1, q1 = ir.variable(1, 1, "var:p"), ir.variable(1, 1, "var:q")
    for trial in range(3):
        u, v, s = rng.randn() + 1, rng.randn() + 1, -abs(rng.randn())
        run(solve_mod, ir.prox(ProxFunction.SUM_KL_DIV, p1, q1, t, epigraph=True), 1.0,
            {"var:p": [u], "var:q": [v], "var:t": [s]}, tol_for(dtype, f64=1e-8, f32=1e-3))


def test_exp_epigraph(solve_mod, dtype):  # prox_test.py:222
    rng = np.random.RandomState(2)
    n = 300
    x, z = ir.variable(n, 1, "var:x"), ir.variable(n, 1, "var:z")
    run(solve_mod, ir.prox(ProxFunction.EXP, x, z, epigraph=True), 1.0,
        {"var:x": rng.randn(n), "var:z": rng.randn(n)}, tol_for(dtype, f64=1e-9))


def test_second_order_cone(solve_mod, dtype):  # prox_test.py:179-183,276-288
    x, t = ir.variable(N, 1, "var:x"), ir.variable(1, 1, "var:t")
    e = ir.prox(ProxFunction.SECOND_ORDER_CONE, t, x, arg_size=[(1, 1), (1, N)])
    cases = [(np.zeros(N), 0.0), (np.arange(N), 100.0), (np.arange(N), 10.0),
             (np.arange(N), -100.0), (np.arange(N), -10.0)]
    for v, s in cases:
        run(solve_mod, e, 1.0, {"var:x": v.astype(float), "var:t": [s]}, tol_for(dtype, f64=1e-12, f32=1e-5))
    for trial in range(4):
        rng = np.random.RandomState(trial)
        ax, at, bx, bt = rng.randn(), abs(rng.randn()) + 0.1, rng.randn(), rng.randn()
        targ = ir.add(ir.linear_map(ir.scalar(at, 1), t), ir.scalar_constant(bt, (1, 1)))
        xarg = ir.add(ir.linear_map(ir.scalar(ax, N), x), ir.scalar_constant(bx, (N, 1)))
        e2 = ir.prox(ProxFunction.SECOND_ORDER_CONE, targ, xarg, arg_size=[(1, 1), (1, N)])
        run(solve_mod, e2, 1.0, {"var:x": rng.randn(N), "var:t": [rng.randn()]},
            tol_for(dtype, f64=1e-11, f32=2e-5))
    rng = np.random.RandomState(3)
    for (m, n) in [(5, 4), (3000, 3)]:
        X, tv = ir.variable(m, n, "var:X"), ir.variable(m, 1, "var:t")
        e3 = ir.prox(ProxFunction.SECOND_ORDER_CONE, tv, X, arg_size=[(m, 1), (m, n)])
        run(solve_mod, e3, 1.0, {"var:X": rng.randn(m * n), "var:t": rng.randn(m)},
            tol_for(dtype, f64=1e-12, f32=1e-5))


def sym(rng, n):
    A = rng.randn(n, n)
    return (A + A.T) / 2


MATRIX = {
    "semidefinite": ProxFunction.SEMIDEFINITE,
    "neg_log_det": ProxFunction.NEG_LOG_DET,
    "lambda_max": ProxFunction.LAMBDA_MAX,
}


@pytest.mark.parametrize("name", sorted(MATRIX))
@pytest.mark.parametrize("n", [3, 8, 33])
def test_symmetric_matrix_prox(solve_mod, dtype, name, n):
    """prox_test.py:169,172,184.  Eigenvalues come from the Jacobi SVD of the shifted matrix; a
    matrix with a +-lambda eigenvalue pair is included (the case a plain SVD cannot resolve)."""
    rng = np.random.RandomState(n)
    X = ir.variable(n, n, "var:X")
    e = ir.prox(MATRIX[name], X)
    V = rng.randn(n, n) if name == "semidefinite" else sym(rng, n)
    tol = tol_for(dtype, f64=1e-8, f32=2e-3)
    run(solve_mod, e, 0.6, {"var:X": V.reshape(-1, order="F")}, tol)
    Q, _ = np.linalg.qr(rng.randn(n, n))
    d = rng.randn(n)
    d[1] = -d[0]
    P = (Q * d) @ Q.T
    run(solve_mod, e, 0.6, {"var:X": ((P + P.T) / 2).reshape(-1, order="F")}, tol)
    run(solve_mod, e, 0.6, {"var:X": np.zeros(n * n)}, tol)  # the first ADMM iterate
    run(solve_mod, e, 0.6, {"var:X": 1e-9 * V.reshape(-1, order="F")}, tol)


@pytest.mark.parametrize("name", ["lambda_max", "neg_log_det", "norm_nuclear"])
def test_matrix_epigraphs(solve_mod, dtype, name):  # prox_test.py:221,226,229
    rng = np.random.RandomState(4)
    n = 6
    X, t = ir.variable(n, n, "var:X"), ir.variable(1, 1, "var:t")
    typ = dict(MATRIX, norm_nuclear=ProxFunction.NORM_NUCLEAR)[name]
    e = ir.prox(typ, X, t, epigraph=True)
    for trial in range(3):
        V = sym(rng, n) if name != "norm_nuclear" else rng.randn(n, n)
        run(solve_mod, e, 1.0, {"var:X": V.reshape(-1, order="F"), "var:t": [rng.randn()]},
            tol_for(dtype, f64=1e-7, f32=3e-3))


@pytest.fixture
def block_svd():
    """The block Jacobi takes over from 1536 columns; EPSILON_HIP_SVD=block forces it for the
    small shapes of these tests (the library reads the variable on every call)."""
    import os
    old = os.environ.get("EPSILON_HIP_SVD")
    os.environ["EPSILON_HIP_SVD"] = "block"
    yield
    if old is None:
        os.environ.pop("EPSILON_HIP_SVD", None)
    else:
        os.environ["EPSILON_HIP_SVD"] = old


@pytest.mark.parametrize("shape", [(300, 250), (250, 400), (200, 200), (520, 193)])
def test_nuclear_norm_block_jacobi(solve_mod, dtype, shape, block_svd):
    """The block Jacobi SVD (batched Gram on the MFMA kernel, 64 x 64 eigenproblems on chip):
    nuclear-norm prox of tall, wide and square matrices against the singular value thresholding
    of numpy's SVD."""
    rng = np.random.RandomState(shape[0])
    m, n = shape
    r = 12
    V = rng.randn(m, r) @ rng.randn(r, n) + 0.1 * rng.randn(m, n)  # low rank + noise, as in RPCA
    lam = 3.0
    X = ir.variable(m, n, "var:X")
    e = ir.prox(ProxFunction.NORM_NUCLEAR, X)
    fb = e.proto.SerializeToString()
    got = solve_mod.eval_prox(fb, lam, e.data, {"var:X": V.reshape(-1, order="F").tobytes()})
    G = np.frombuffer(got["var:X"]).reshape((m, n), order="F")
    U, sv, Vt = np.linalg.svd(V, full_matrices=False)
    want = (U * np.maximum(sv - lam, 0)) @ Vt
    tol = dict(rtol=0, atol=1e-8 * sv[0]) if dtype == "f64" else dict(rtol=0, atol=3e-4 * sv[0])
    np.testing.assert_allclose(G, want, **tol)


def test_symmetric_functions_block_jacobi(solve_mod, dtype, block_svd):
    """SEMIDEFINITE and NEG_LOG_DET at n = 256 (block path through the shifted SVD)."""
    rng = np.random.RandomState(7)
    n = 256
    A = rng.randn(n, n)
    S = (A + A.T) / 2
    X = ir.variable(n, n, "var:X")
    got = solve_mod.eval_prox(ir.prox(ProxFunction.SEMIDEFINITE, X).proto.SerializeToString(), 1.0, {},
                              {"var:X": S.reshape(-1, order="F").tobytes()})
    G = np.frombuffer(got["var:X"]).reshape((n, n), order="F")
    d, Q = np.linalg.eigh(S)
    want = (Q * np.maximum(d, 0)) @ Q.T
    scale = np.abs(d).max()
    np.testing.assert_allclose(G, want, rtol=0, atol=(1e-8 if dtype == "f64" else 2e-3) * scale)
    lam = 0.7
    got = solve_mod.eval_prox(ir.prox(ProxFunction.NEG_LOG_DET, X).proto.SerializeToString(), lam, {},
                              {"var:X": S.reshape(-1, order="F").tobytes()})
    G = np.frombuffer(got["var:X"]).reshape((n, n), order="F")
    dd = (d + np.sqrt(d * d + 4 * lam)) / 2
    want = (Q * dd) @ Q.T
    np.testing.assert_allclose(G, want, rtol=0, atol=(1e-8 if dtype == "f64" else 2e-3) * scale)


@pytest.mark.parametrize("n", [96, 300, 700])
def test_svd_factors_are_orthogonal_and_reconstruct(solve_mod, n):
    """The decomposition behind the orthogonally-invariant operators on the reference's robust-PCA
    matrix (problems/robust_pca.py:5-22: rank 10 + 10 % sparse part), through the library's
    microbenchmark entry: on-chip kernel (n = 96) and block Jacobi on the matrix cores (fp32, every
    larger size).  Y = W V^T with V orthogonal and W's columns orthogonal, cold and warm-started;
    the defects are those of fp32 rotations accumulated over the sweeps (no Newton-Schulz clean-up
    here, the prox operator adds that)."""
    import ctypes
    from epsilon_amd import _solve
    solve_mod.set_option("dtype", "f32")
    L = _solve.lib()
    ms_c, ms_w = ctypes.c_double(), ctypes.c_double()
    sw_c, sw_w = ctypes.c_int(), ctypes.c_int()
    d = (ctypes.c_double * 6)()
    _solve._check(L.eps_bench_svd(ctypes.c_int64(n), ctypes.c_int64(n), ctypes.c_int(10), ctypes.c_int(40),
                                  ctypes.c_double(1e-3), ctypes.byref(ms_c), ctypes.byref(sw_c),
                                  ctypes.byref(ms_w), ctypes.byref(sw_w), d))
    assert 2 <= sw_c.value < 40 and 1 <= sw_w.value <= sw_c.value
    for off in (0, 3):  # cold, warm
        assert d[off + 0] < 2e-4, list(d)   # ||V^T V - I||_F / sqrt(n)
        assert d[off + 1] < 2e-4, list(d)   # ||W V^T - Y||_F / ||Y||_F
        assert d[off + 2] < 2e-3, list(d)   # off-diagonal mass of W^T W


@pytest.mark.parametrize("case", ["low_rank_tall", "low_rank_wide", "many_above", "full_rank", "edge_at_threshold",
                                  "edge_just_above"])
def test_nuclear_norm_thresholded_partial_svd(solve_mod, dtype, case):
    """From 512 columns / rows the nuclear-norm prox first tries the leading singular block alone
    (randomized subspace iteration + Jacobi on the small problem, certified by the residuals of
    the pairs above the threshold and by the spectral norm of the remainder), and falls back to the
    full decomposition when the spectrum above lambda is wide or does not separate
    (reference prox/ortho_invariant.cc:76-105 thresholds every singular value the same way).
    Tall, wide, 70 values above the threshold, a full-rank matrix (fallback) and a bulk edge that
    sits right at the threshold - all against numpy's SVD."""
    from epsilon_amd import _solve
    rng = np.random.RandomState(11)
    if case == "low_rank_tall":
        m, n, r, noise, lam = 900, 600, 12, 0.01, 2.0
    elif case == "low_rank_wide":
        m, n, r, noise, lam = 560, 1100, 9, 0.01, 2.0
    elif case == "many_above":
        m, n, r, noise, lam = 800, 800, 70, 0.01, 2.0
    elif case == "full_rank":
        m, n, r, noise, lam = 640, 640, 5, 1.0, 3.0     # bulk up to ~50: everything above lambda
    elif case == "edge_at_threshold":
        m, n, r, noise, lam = 700, 640, 8, 0.05, 1.8    # bulk edge 0.05 (sqrt(700) + sqrt(640)) ~ 2.6
    else:
        # the top of the noise bulk a few per cent ABOVE the threshold: the power-iteration
        # certificate of the remainder is a lower bound, so it must not accept here on its
        # resolution alone (the fp64 tolerance below sees a single value kept out of the result)
        m, n, r, noise, lam = 700, 640, 8, 0.05, 2.5
    V = rng.randn(m, r) @ rng.randn(r, n) + noise * rng.randn(m, n)
    X = ir.variable(m, n, "var:X")
    e = ir.prox(ProxFunction.NORM_NUCLEAR, X)
    fb = e.proto.SerializeToString()
    _solve.profile_enable(True)
    _solve.profile_reset()
    got = solve_mod.eval_prox(fb, lam, e.data, {"var:X": V.reshape(-1, order="F").tobytes()})
    tags = _solve.profile_dump()
    _solve.profile_enable(False)
    G = np.frombuffer(got["var:X"]).reshape((m, n), order="F")
    U, sv, Vt = np.linalg.svd(V, full_matrices=False)
    want = (U * np.maximum(sv - lam, 0)) @ Vt
    tol = dict(rtol=0, atol=1e-8 * sv[0]) if dtype == "f64" else dict(rtol=0, atol=3e-4 * sv[0])
    np.testing.assert_allclose(G, want, **tol)
    # the partial route was entered in every case; the full decomposition ran only where it had to
    names = list(tags)
    assert any(t.startswith("partial_svd") for t in names), names
    full = any(t in names for t in ("block_jacobi_svd:%dx%d" % (m, n), "jacobi_svd:%dx%d" % (m, n),
                                    "block_jacobi_svd_no_v:%dx%d" % (m, n)))  # (fp32, m >= n: the one-sided form)
    if case != "edge_just_above":  # (either route is fine there, the result is what counts)
        assert full == (case in ("full_rank", "edge_at_threshold")), (case, names)


@pytest.mark.parametrize("case", ["rpca", "bulk_only", "tall", "near_threshold"])
def test_nuclear_norm_polar_route(solve_mod, dtype, case):
    """Round 3: a nuclear-norm prox whose spectrum is (numerically) full above the threshold runs
    on GEMMs only - polar factor Q of Y by Newton-Schulz, X = Q (H - tau I)_+ with the positive part
    from the matrix sign function, the dominant outliers taken out first by a randomized block -
    and is held to the optimality condition of the prox before it is returned
    (prox_more.cc: PolarNuclearProx; reference prox/ortho_invariant.cc:36-105).  Against numpy's SVD:
    the reference's robust-PCA matrix, a pure noise bulk, a tall matrix, and a spectrum with many
    values within a per cent of the threshold (where the sign iteration is NOT converged and must
    not matter)."""
    from epsilon_amd import _solve
    rng = np.random.RandomState(21)
    if case == "rpca":
        m = n = 1100
        V = rng.randn(m, 10) @ rng.randn(10, n) + (rng.rand(m, n) < 0.1) * (10 * rng.randn(m, n))
        lam = 1.0
    elif case == "bulk_only":
        m = n = 1056
        V = rng.randn(m, n)
        lam = 3.0
    elif case == "tall":
        m, n = 1500, 1030
        V = rng.randn(m, 6) @ rng.randn(6, n) * 3 + rng.randn(m, n)
        lam = 2.0
    else:
        m = n = 1040
        U0, _ = np.linalg.qr(rng.randn(m, n))
        V0, _ = np.linalg.qr(rng.randn(n, n))
        sv0 = np.concatenate([np.linspace(0.98, 1.02, 400), np.linspace(1.5, 40, n - 400)])
        V = (U0 * sv0) @ V0.T
        lam = 1.0
    X = ir.variable(m, n, "var:X")
    e = ir.prox(ProxFunction.NORM_NUCLEAR, X)
    _solve.profile_enable(True)
    _solve.profile_reset()
    got = solve_mod.eval_prox(e.proto.SerializeToString(), lam, e.data, {"var:X": V.reshape(-1, order="F").tobytes()})
    tags = _solve.profile_dump()
    _solve.profile_enable(False)
    G = np.frombuffer(got["var:X"]).reshape((m, n), order="F")
    U, sv, Vt = np.linalg.svd(V, full_matrices=False)
    want = (U * np.maximum(sv - lam, 0)) @ Vt
    assert any(t.startswith("polar_prox") for t in tags), sorted(tags)
    # the decomposition must NOT have run: the polar route's own certificate accepted the result
    assert not any(t.startswith(("block_jacobi_svd:%dx%d" % (m, n), "block_jacobi_svd_no_v:%dx%d" % (m, n)))
                   for t in tags), sorted(tags)
    err = np.linalg.norm(G - want, 2)
    # fp32: rounding of the products ~6e-6 ||Y_rest||_2 per direction; an unconverged sign within the
    # resolution (1e-3 lam) of the threshold costs <= 5e-4 lam in that direction
    tol = (1e-8 if dtype == "f64" else 2e-5) * sv[0] + (1e-3 * lam if case == "near_threshold" else 0.0)
    assert err <= tol, (case, err, tol)


@pytest.mark.parametrize("name", ["max", "sum_largest_9", "log_sum_exp"])
@pytest.mark.parametrize("shape", [(200001, 1), (3, 150000)])
def test_vector_prox_long_slices(solve_mod, dtype, name, shape):
    """Slices of 131072 entries and more are solved with every reduction on the whole grid (one
    launch per iteration of the scalar equation, the last-arriving workgroup advances the state)
    instead of by one workgroup: one long vector, and three long rows of a matrix through the axis
    form (strided slices).  Same oracle, same tolerances as the short slices."""
    typ, kw = VECTOR_PROX[name]
    rng = np.random.RandomState(5)
    m, n = shape
    if n == 1:
        x = ir.variable(m, 1, "var:x")
        v = rng.randn(m)
        e = ir.prox(typ, x, **kw)
    else:
        x = ir.variable(m, n, "var:x")
        v = rng.randn(m * n)
        e = ir.prox(typ, x, has_axis=True, axis=1, **kw)   # one slice per row (stride m)
    run(solve_mod, e, 0.7, {"var:x": v}, tol_for(dtype))


LONG_EPIGRAPHS = ["max", "sum_largest_4", "log_sum_exp", "sum_exp", "sum_logistic", "sum_neg_entr",
                  "sum_inv_pos", "sum_neg_log"]


@pytest.mark.parametrize("name", LONG_EPIGRAPHS)
def test_vector_epigraph_long_slice_routes_agree(solve_mod, dtype, name):
    """Epigraph projections of one long slice (140 000 entries): the grid-wide route (every
    reduction of the scalar iteration one launch of the whole chip) against the one-workgroup
    route of the same library (EPSILON_HIP_SEG_GRID=0), which the short-slice tests hold to the
    oracle.  Same algorithm, same fp64 scalars: they differ by the summation order only.  An
    infeasible point, and a feasible one (the easy case)."""
    import os
    typ, kw = EPIGRAPHS[name]
    n = 140000
    x, t = ir.variable(n, 1, "var:x"), ir.variable(1, 1, "var:t")
    e = ir.prox(typ, x, t, epigraph=True, **kw)
    fb = e.proto.SerializeToString()
    rng = np.random.RandomState(21)
    v = np.abs(rng.randn(n)) * 0.5 + 0.2
    for s_in in (-3.0, 1e9):
        vb = {"var:x": v.tobytes(), "var:t": np.array([s_in]).tobytes()}
        got = {k: np.frombuffer(b) for k, b in solve_mod.eval_prox(fb, 1.0, e.data, vb).items()}
        os.environ["EPSILON_HIP_SEG_GRID"] = "0"
        try:
            ref = {k: np.frombuffer(b) for k, b in solve_mod.eval_prox(fb, 1.0, e.data, vb).items()}
        finally:
            os.environ.pop("EPSILON_HIP_SEG_GRID", None)
        tol = dict(rtol=1e-9, atol=1e-9) if dtype == "f64" else dict(rtol=2e-5, atol=2e-5)
        if name == "sum_largest_4":  # both stop their bisection at |g| <= 1e-5: same bracket sequence
            tol = dict(rtol=1e-6, atol=1e-6) if dtype == "f64" else dict(rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(got["var:x"], ref["var:x"], err_msg=name, **tol)
        np.testing.assert_allclose(got["var:t"], ref["var:t"], err_msg=name, **tol)


def test_long_slices_norm2_soc_zone_epigraph(solve_mod, dtype):
    """The single-reduction operators and the Michelot iteration on long slices (grid-wide route):
    NORM_2 per row of a 2 x 140000 matrix, one second-order cone of 150000 entries, and the
    scaled-zone epigraph (hinge, deadzone) per column of a 140000 x 2 matrix - against the
    oracle."""
    rng = np.random.RandomState(17)
    m, n = 2, 140000
    X = ir.variable(m, n, "var:X")
    run(solve_mod, ir.prox(ProxFunction.NORM_2, X, has_axis=True, axis=1), 40.0,
        {"var:X": rng.randn(m * n)}, tol_for(dtype))
    N1 = 150000
    x, t = ir.variable(N1, 1, "var:x"), ir.variable(1, 1, "var:t")
    e = ir.prox(ProxFunction.SECOND_ORDER_CONE, t, x, arg_size=[(1, 1), (1, N1)])
    for s in (500.0, 100.0, -100.0, -500.0):
        run(solve_mod, e, 1.0, {"var:x": rng.randn(N1), "var:t": [s]}, tol_for(dtype, f64=1e-11, f32=2e-5))
    Y = ir.variable(n, m, "var:Y")
    tv = ir.variable(1, m, "var:t")
    for typ, kw in ((ProxFunction.SUM_HINGE, {}),
                    (ProxFunction.SUM_DEADZONE, dict(scaled_zone_params=wire.ProxScaledZoneParams(m=0.3)))):
        ez = ir.prox(typ, Y, tv, epigraph=True, has_axis=True, axis=0, **kw)
        run(solve_mod, ez, 1.0, {"var:Y": rng.randn(n * m), "var:t": [100.0, 1e7]}, tol_for(dtype))
