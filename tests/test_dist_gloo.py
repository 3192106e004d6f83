"""N > 1 path on CPU: world_size-2 (and 3) gloo runs of the column-sharded lasso sweep - the
partitioning and the placement of the collectives - against the single-process oracle."""

import numpy as np
import pytest

from epsilon_amd import dist as edist
from epsilon_amd import problems, wire
from oracle import epsilon_oracle as orc
from tests import mp_util


def test_column_range_covers_everything():
    for n in (1, 7, 500, 50000, 50001):
        for world in (1, 2, 3, 4, 8):
            for align in (1, 64):
                spans = [edist.column_range(n, r, world, align) for r in range(world)]
                assert spans[0][0] == 0 and spans[-1][1] == n
                for a, b in zip(spans, spans[1:]):
                    assert a[1] == b[0] and a[0] <= a[1]
                if align > 1:
                    assert all(s[0] % align == 0 for s in spans if s[1] > s[0])


def test_lasso_sharded_keys():
    prob, _ = problems.lasso(8, 20, seed=0)
    assert edist.lasso_sharded_keys(prob) == ["separate:var:x:sum_square", "var:x", "constraint:0"]


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_sweep_equals_single_process(tmp_path, world):
    m, n = 40, 101
    x0, x1, status, _ = mp_util.run_ranks(world, "oracle", str(tmp_path), m, n, seed=3)
    prob, info = problems.lasso(m, n, seed=3)
    st, x = orc.solve(prob.SerializeToString(), [], wire.SolverParams().SerializeToString(),
                      prob.expression_data())
    S = wire.SolverStatus.FromString(st)
    for s in status:  # every rank sees the same residuals and stops at the same sweep
        assert int(s[0]) == S.num_iterations
        np.testing.assert_allclose(s[1:], [S.residuals.r_norm, S.residuals.s_norm,
                                           S.residuals.epsilon_primal, S.residuals.epsilon_dual],
                                   rtol=1e-9)
    np.testing.assert_allclose(x0, np.frombuffer(x["separate:var:x:sum_square"]), rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(x1, np.frombuffer(x["var:x"]), rtol=1e-9, atol=1e-11)


def consensus_reference(m, n, seed, world):
    A, b = problems.regression_data(m, n, seed=seed)
    lam = 0.3 * np.abs(A.T.dot(b)).max()
    prob = problems.consensus_lasso(A, b, lam, world)
    st, x = orc.solve(prob.SerializeToString(), [], wire.SolverParams().SerializeToString(),
                      prob.expression_data())
    return wire.SolverStatus.FromString(st), x


@pytest.mark.parametrize("world", [2, 3])
def test_consensus_sweep_equals_single_process(tmp_path, world):
    """Consensus form (SURVEY.md 8(e) mode E2): rows of A split over the ranks, one term f_g per
    rank, z-averaging all-reduce.  The per-rank sweeps must reproduce the single-process solve
    of the stacked problem (terms f_1..f_G, h; G consensus constraints) sweep for sweep."""
    m, n = 61, 23
    x0, x1, status, parts = mp_util.run_ranks(world, "oracle_consensus", str(tmp_path), m, n, seed=4)
    S, x = consensus_reference(m, n, 4, world)
    for s in status:
        assert int(s[0]) == S.num_iterations
        np.testing.assert_allclose(s[1:], [S.residuals.r_norm, S.residuals.s_norm,
                                           S.residuals.epsilon_primal, S.residuals.epsilon_dual],
                                   rtol=1e-9)
    for g, p in enumerate(parts):
        np.testing.assert_allclose(p["x0"], np.frombuffer(x["var:x_%d" % g]), rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(p["x1"], np.frombuffer(x[problems.CONSENSUS_Z]), rtol=1e-9, atol=1e-11)


@pytest.mark.parametrize("world", [2, 3])
def test_row_sharded_robust_pca_equals_single_process(tmp_path, world):
    """BASELINE.json configs[4] sharded by rows: the singular value thresholding runs through the
    all-reduced Gram matrix, the residual norms through all-reduced partial sums; the ranks must
    reproduce the oracle's single-process solve sweep for sweep."""
    nn = 26
    x0, x1, status, parts = mp_util.run_ranks(world, "oracle_rpca", str(tmp_path), nn, 8, seed=2,
                                              max_iter=400)
    M = problems.robust_pca_data(nn, r=3, density=0.1, seed=2)
    prob = problems.robust_pca_ir(M, 0.1)
    st, x = orc.solve(prob.SerializeToString(), [], wire.SolverParams(max_iterations=400).SerializeToString(),
                      prob.expression_data())
    S = wire.SolverStatus.FromString(st)
    for s in status:
        assert int(s[0]) == S.num_iterations
        np.testing.assert_allclose(s[1:], [S.residuals.r_norm, S.residuals.s_norm,
                                           S.residuals.epsilon_primal, S.residuals.epsilon_dual],
                                   rtol=1e-6)
    np.testing.assert_allclose(x0, np.frombuffer(x["var:L"]).reshape(nn, nn, order="F"), rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(x1, np.frombuffer(x["var:S"]).reshape(nn, nn, order="F"), rtol=1e-6, atol=1e-8)
