"""Caller-level shim `epsilon_amd.solver` (reference python/epopt/cvxpy_solver.py:27-104): status
mapping and column-major unpacking on CPU; the solve routes - incl. the single-prox shortcut
`eval_prox(f, lam=1e12, data, {})` - on the GPU against the oracle."""

import numpy as np
import pytest

from epsilon_amd import ir, problems, solver, wire
from epsilon_amd.wire import ProxFunction
from oracle import epsilon_oracle as orc


def test_cvxpy_status_mapping():
    S = wire.SolverStatus
    assert solver.cvxpy_status(S(state=S.OPTIMAL)) == "optimal"
    assert solver.cvxpy_status(S(state=S.MAX_ITERATIONS_REACHED)) == "optimal_inaccurate"
    assert solver.cvxpy_status(S(state=S.RUNNING)) == "solver_error"
    assert solver.cvxpy_status(S()) == "solver_error"


def test_unpack_is_column_major():
    # a 2 x 3 variable arrives as its columns one after the other (cvxpy_solver.py:27-32)
    X = np.arange(6.0).reshape(2, 3)
    out = solver.unpack_values({"var:X": (2, 3)}, {"var:X": X.tobytes(order="F")})
    np.testing.assert_array_equal(out["var:X"], X)
    with pytest.raises(solver.SolverError):
        solver.unpack_values({"var:X": (2, 3)}, {})
    with pytest.raises(solver.SolverError):
        solver.unpack_values({"var:X": (2, 2)}, {"var:X": X.tobytes()})


def test_parameter_values_ship_as_data_blobs():
    plist, data = solver.parameter_values({"param:b": np.array([1.0, 2.0, 3.0])})
    (pid, cbytes), = plist
    c = wire.Constant.FromString(cbytes)
    assert pid == "param:b" and (c.m, c.n) == (3, 1) and c.data_location in data
    np.testing.assert_array_equal(np.frombuffer(data[c.data_location]), [1.0, 2.0, 3.0])


def _single_prox_cases():
    rng = np.random.RandomState(0)
    n = 9
    c = rng.randn(n)
    x = ir.variable(n, 1, "var:x")
    norm1 = ir.prox(ProxFunction.NORM_1, ir.add(x, ir.linear_map(ir.scalar(-1, n), ir.constant(c))), alpha=1.0)
    A, b = rng.randn(14, 5), rng.randn(14)
    y = ir.variable(5, 1, "var:y")
    lsq = ir.prox(ProxFunction.SUM_SQUARE,
                  ir.add(ir.linear_map(ir.dense_matrix(A), y), ir.linear_map(ir.scalar(-1, 14), ir.constant(b))),
                  alpha=1.0)
    z = ir.variable(30, 1, "var:z")
    cz = np.cumsum(rng.randn(30))
    tv = ir.prox(ProxFunction.TOTAL_VARIATION_1D,
                 ir.add(z, ir.linear_map(ir.scalar(-1, 30), ir.constant(cz))), alpha=2.0)
    return {"norm_1": (norm1, {"var:x": c}), "least_squares": (lsq, {"var:y": np.linalg.lstsq(A, b, rcond=None)[0]}),
            # any constant shift of c minimises TV(z - c); the 1/2 ||z||^2 / lam tie-break picks mean zero
            "tv_only": (tv, {"var:z": cz - cz.mean()})}


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["norm_1", "least_squares", "tv_only"])
def test_single_prox_route(solve_mod, name):
    """One objective term, no constraints: the reference evaluates ONE prox with lam = 1e12 and an
    empty v (cvxpy_solver.py:79-88)."""
    solve_mod.set_option("dtype", "f64")
    try:
        f, want = _single_prox_cases()[name]
        prob = ir.Problem([f], [])
        status, values, info = solver.solve(prob)
        assert status == solver.OPTIMAL and info["route"] == "eval_prox"
        ref = orc.eval_prox(f.proto.SerializeToString(), 1e12, f.data, {})
        for k, v in want.items():
            assert values[k].shape == (v.size, 1)
            # lam = 1e12 scales the prox input by 1e-6 and back: ~1e-5 of absolute noise on both
            # sides (the two TV algorithms - dynamic program vs level sets - round differently)
            tol = 1e-4 if name == "tv_only" else 1e-5
            np.testing.assert_allclose(values[k].ravel(), np.frombuffer(ref[k]), rtol=tol, atol=tol)
            np.testing.assert_allclose(values[k].ravel(), v, rtol=tol, atol=tol)
    finally:
        solve_mod.set_option("dtype", "f32")


@pytest.mark.gpu
def test_single_prox_route_without_a_constant_fails_like_the_reference(solve_mod):
    """A vector prox whose argument carries no constant sees an EMPTY input on this route
    (v = {} and g empty, prox/vector_prox.cc:141-143), and reading block "arg:0" of it is a failed
    CHECK in the reference (vector/block_vector.cc operator()) - `_solve.error` here, and an error
    in the oracle too."""
    x = ir.variable(9, 1, "var:x")
    f = ir.prox(ProxFunction.NORM_1, x, alpha=1.0)
    with pytest.raises(solve_mod.error):
        solver.solve(ir.Problem([f], []))
    with pytest.raises(Exception):
        orc.eval_prox(f.proto.SerializeToString(), 1e12, f.data, {})


@pytest.mark.gpu
def test_solve_route_and_warm_start_cache(solve_mod):
    """The general route (status mapping, (n, m).T unpack of a matrix variable) and the
    warm-started handle cache with a re-bound parameter (cvxpy_solver.py:70-74,91-95)."""
    solve_mod.set_option("dtype", "f64")
    try:
        # matrix variable: robust PCA n = 8 (variables are 8 x 8)
        prob, info = problems.robust_pca(8, r=2, seed=1)
        status, values, meta = solver.solve(prob, max_iterations=300)
        st_o, x_o = orc.solve(prob.SerializeToString(), [],
                              wire.SolverParams(max_iterations=300).SerializeToString(), prob.expression_data())
        so = wire.SolverStatus.FromString(st_o)
        assert status == solver.cvxpy_status(so) and meta["num_iterations"] == so.num_iterations
        for k, v in values.items():
            assert v.shape == (8, 8)
            np.testing.assert_allclose(v, np.frombuffer(x_o[k]).reshape(8, 8, order="F"), rtol=1e-7, atol=1e-9)
        # non-convergence is not an error: MAX_ITERATIONS_REACHED -> optimal_inaccurate
        status, _, meta = solver.solve(prob, max_iterations=3)
        assert status == solver.OPTIMAL_INACCURATE and meta["num_iterations"] == 3
        # warm start with a parameter
        A, b = problems.regression_data(40, 90, seed=2)
        lam = 0.3 * np.abs(A.T.dot(b)).max()
        lp = problems.lasso_ir(ir.dense_matrix(A), ir.parameter(40, 1, "param:b"), lam, 90)
        s1, v1, m1 = solver.solve(lp, {"param:b": b}, warm_start=True)
        s2, v2, m2 = solver.solve(lp, {"param:b": 1.05 * b}, warm_start=True)
        assert m1["route"] == m2["route"] == "warm_start_handle" and s1 == s2 == solver.OPTIMAL
        assert m2["num_iterations"] <= m1["num_iterations"]
        cold = solver.solve(lp, {"param:b": 1.05 * b})
        obj = lambda v: problems.lasso_objective(A, 1.05 * b, lam, v[problems.LASSO_VAR].ravel())  # noqa: E731
        assert abs(obj(v2) - obj(cold[1])) <= 1e-2 * abs(obj(cold[1]))
    finally:
        solver.clear_warm_start_cache()
        solve_mod.set_option("dtype", "f32")
