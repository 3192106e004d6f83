"""BASELINE.json configs[3] on the reference's OWN data: multiclass hinge on the 2000-sample MNIST
subset the reference ships (python/epopt/problems/mnist_small.mat, loaded by problems/mnist.py:14-22;
the arrays are committed as data in tests/golden/mnist_small.npz by tests/golden/make_mnist_small.py),
set up as the notebook does (docs/notebooks/mnist.rst:88-105: X / 255, k = 10, lam = 1; compiled form
printed at mnist.rst:118-129).  Both drivers: stopping sweep, residuals and iterates against the
oracle; and the objective against an independent scipy solve (tests/golden/mnist_small_optimum.json,
L-BFGS on a smoothing with continuation - no code of this repository)."""

import hashlib
import json
import os

import numpy as np
import pytest

from epsilon_amd import problems, wire
from oracle import epsilon_oracle as orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load():
    d = np.load(os.path.join(GOLDEN, "mnist_small.npz"))
    X8, y = d["X"], d["y"]
    X = X8.astype(np.float64) / 255.0
    Y = np.zeros((X.shape[0], 10))
    Y[np.arange(X.shape[0]), y.astype(int)] = 1.0
    return X8, y, X, Y


def test_fixture_is_the_reference_data():
    X8, y, X, Y = load()
    assert X8.shape == (2000, 784) and X8.dtype == np.uint8 and y.shape == (2000,)
    assert hashlib.sha256(X8.tobytes()).hexdigest()[:16] == "4d1e2862007bab59"
    assert hashlib.sha256(y.tobytes()).hexdigest()[:16] == "e1be9271eae282fe"
    assert set(np.unique(y)) == set(range(10))
    opt = json.load(open(os.path.join(GOLDEN, "mnist_small_optimum.json")))
    assert opt["objective_lower_bound"] <= opt["objective_upper_bound"]
    assert opt["objective_upper_bound"] - opt["objective_lower_bound"] < 0.01 * opt["objective_upper_bound"]


@pytest.mark.gpu
@pytest.mark.parametrize("solver_id", [0, 1])
@pytest.mark.parametrize("dt", ["f64", "f32"])
def test_multiclass_hinge_on_mnist_small_matches_oracle(solve_mod, solver_id, dt):
    _, _, X, Y = load()
    prob, _ = problems.multiclass_hinge(X, Y, 1.0)
    pb, data = prob.SerializeToString(), prob.expression_data()
    sb = wire.SolverParams(solver=solver_id).SerializeToString()
    solve_mod.set_option("dtype", dt)
    try:
        st_g, x_g = solve_mod.solve(pb, [], sb, data)
    finally:
        solve_mod.set_option("dtype", "f32")
    st_o, x_o = orc.solve(pb, [], sb, data)
    g, o = wire.SolverStatus.FromString(st_g), wire.SolverStatus.FromString(st_o)
    assert o.state == wire.SolverStatus.OPTIMAL
    assert g.state == o.state and g.num_iterations == o.num_iterations, (g, o)
    # fp32 + two-block driver: the z-update projects onto ALL constraints at once through one
    # block factorisation whose Schur complement holds X X^T of raw pixel data (2000 x 2000, rank
    # <= 784, + I): kappa ~ 1e5, so the explicit-inverse solve alone carries kappa * eps ~ 1e-2 of
    # forward error per sweep in fp32 (round 2 could only hold these iterates norm-wise to 20 %).
    # The library now estimates kappa_1 of every pivot block at Init and refines the block solve
    # against the KKT blocks as given (block.cc; include/epsilon_hip.h "refine"): all four
    # combinations meet the same bounds.
    rt = 1e-7 if dt == "f64" else 2e-3
    for f in ("r_norm", "s_norm", "epsilon_primal", "epsilon_dual"):
        np.testing.assert_allclose(getattr(g.residuals, f), getattr(o.residuals, f), rtol=rt, atol=1e-6)
    # iterates: fp64 to rounding; fp32 within 1e-2 of the largest entry of each variable (the
    # tolerance of the reference's own checks, prox_test.py:250-266; measured: 6e-3 on `t` after the
    # 30 sweeps - pixel data, K = 784 contractions, no contraction of the rounding yet)
    for k in x_o:
        a, b = np.frombuffer(x_g[k]), np.frombuffer(x_o[k])
        atol = (1e-8 if dt == "f64" else 1e-2) * max(1.0, np.abs(b).max())
        np.testing.assert_allclose(a, b, rtol=0, atol=atol, err_msg=k)


@pytest.mark.gpu
def test_ill_conditioned_block_solve_is_refined_in_fp32(solve_mod):
    """The decision itself: on mnist_small the two-block driver's constraint projection has a pivot
    block with kappa_1 >> 1e3, so the fp32 mode refines it; with refinement forced off the iterates
    drift from the fp64 oracle by an order of magnitude more (the regression the 20 % bound of round
    2 would have hidden); fp64 never refines; a well-conditioned lasso never refines."""
    _, _, X, Y = load()
    prob, _ = problems.multiclass_hinge(X, Y, 1.0)
    pb, data = prob.SerializeToString(), prob.expression_data()
    sb = wire.SolverParams(solver=1).SerializeToString()
    _, x_o = orc.solve(pb, [], sb, data)

    def run(refine, dt="f32"):
        solve_mod.set_option("refine", refine)
        solve_mod.set_option("dtype", dt)
        solve_mod.block_solve_stats(reset=True)
        try:
            _, x = solve_mod.solve(pb, [], sb, data)
        finally:
            solve_mod.set_option("refine", "auto")
            solve_mod.set_option("dtype", "f32")
        cond, steps = solve_mod.block_solve_stats(reset=True)
        err = max(np.linalg.norm(np.frombuffer(x[k]) - np.frombuffer(x_o[k])) /
                  max(np.linalg.norm(np.frombuffer(x_o[k])), 1e-30) for k in x_o)
        return cond, steps, err

    cond, steps, err = run("auto")
    assert cond > 1e4 and steps >= 1, (cond, steps)
    assert err < 5e-3, err
    cond0, steps0, err0 = run("0")
    assert steps0 == 0
    assert err0 > 3 * err, (err0, err)
    _, steps64, err64 = run("auto", "f64")
    assert steps64 == 0 and err64 < 1e-7
    # config-1-shaped lasso: I + A A^T with unit columns has kappa ~ 10: no refinement, fused path kept
    lp, _ = problems.lasso(100, 300, seed=0)
    solve_mod.block_solve_stats(reset=True)
    solve_mod.solve(lp.SerializeToString(), [], wire.SolverParams().SerializeToString(), lp.expression_data())
    cond_l, steps_l = solve_mod.block_solve_stats(reset=True)
    assert steps_l == 0 and cond_l < 1e3, (cond_l, steps_l)


@pytest.mark.gpu
def test_multiclass_hinge_on_mnist_small_reaches_the_independent_optimum(solve_mod):
    """At the reference's default tolerance the solver stops after ~30 sweeps far from the optimum
    (objective ~200 against ~66: rel_tol = 1e-2 is loose for this problem, in the reference too -
    its notebook reports 8.5 % training error after 40 sweeps); run on (30000 sweeps, rel_tol = 1e-6,
    fp64 on the device) the solve has to land within 1 % of the independent optimum."""
    _, _, X, Y = load()
    opt = json.load(open(os.path.join(GOLDEN, "mnist_small_optimum.json")))
    prob, _ = problems.multiclass_hinge(X, Y, 1.0)
    sb = wire.SolverParams(rel_tol=1e-6, abs_tol=1e-8, max_iterations=30000).SerializeToString()
    solve_mod.set_option("dtype", "f64")
    try:
        st, x = solve_mod.solve(prob.SerializeToString(), [], sb, prob.expression_data())
    finally:
        solve_mod.set_option("dtype", "f32")
    S = wire.SolverStatus.FromString(st)
    Theta = np.frombuffer(x["var:Theta"]).reshape(784, 10, order="F")
    obj = problems.multiclass_hinge_objective(X, Y, 1.0, Theta)
    assert S.state in (wire.SolverStatus.OPTIMAL, wire.SolverStatus.MAX_ITERATIONS_REACHED), S
    assert obj <= opt["objective_upper_bound"] * 1.01, (obj, opt)
    assert obj >= opt["objective_lower_bound"] * 0.99, (obj, opt)
