"""Pins the CPU oracle with the reference's own C++ test expectations, restated
(the reference binaries cannot be built here: glog / generated protobuf headers are absent).

  linear/linear_map_test.cc:67-229          all pairings of the typed maps for * and +
  linear/dense_matrix_impl_test.cc:24-29    dense Apply and transposed Apply
  linear/kronecker_product_impl_test.cc:9-20  Apply(vec X) == vec(B X A^T)
  vector/block_cholesky_test.cc:23-104      ForwardSub, BackSub, ComputeFill == 4 and 25,
                                            block LDL solve vs a dense Cholesky solve (1e-8)
  vector/block_matrix_test.cc, block_vector_test.cc   container algebra
"""

import numpy as np
import pytest
import scipy.sparse as sp

from oracle import epsilon_oracle as orc
from oracle.epsilon_oracle import LM, BlockCholesky, BlockMatrix, BlockVector


import json as _json
import os as _os

GOLDEN = _json.load(open(_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "golden",
                                       "reference_known_answers.json")))



def rnd(rng, *shape):
    return rng.uniform(-1, 1, size=shape)  # Eigen::Random is uniform on [-1, 1]


def maps(rng, m, n):
    """One map of each type with the given shape (square for diagonal / scalar)."""
    out = {"dense": LM.dense(rnd(rng, m, n)), "sparse": LM.sparse(sp.random(m, n, 0.5, random_state=rng))}
    if m == n:
        out["diag"] = LM.diagonal(rnd(rng, n))
        out["scalar"] = LM.scalar(-1.5, n)
    for a in range(1, m + 1):
        if m % a == 0:
            for b in range(1, n + 1):
                if n % b == 0 and 1 < a * b and (a, b) != (m, n):
                    out["kron"] = LM.kron(LM.dense(rnd(rng, a, b)), LM.dense(rnd(rng, m // a, n // b)))
    return out


def test_multiply_all_pairings_match_dense():  # linear_map_test.cc:67-150
    rng = np.random.RandomState(0)
    for A in maps(rng, 6, 6).values():
        for B in maps(rng, 6, 6).values():
            C = orc.lm_multiply(A, B)
            np.testing.assert_allclose(C.as_dense(), A.as_dense() @ B.as_dense(), atol=1e-8)
            C = orc.lm_multiply(A.T(), B)
            np.testing.assert_allclose(C.as_dense(), A.as_dense().T @ B.as_dense(), atol=1e-8)


def test_add_all_pairings_match_dense():  # linear_map_test.cc:152-229
    rng = np.random.RandomState(1)
    for A in maps(rng, 6, 6).values():
        for B in maps(rng, 6, 6).values():
            C = orc.lm_add(A, B)
            np.testing.assert_allclose(C.as_dense(), A.as_dense() + B.as_dense(), atol=1e-8)


def test_result_types_of_the_tables():
    """The special forms the fill model depends on (linear_map_multiply.cc:188-241,
    linear_map_add.cc:167-226)."""
    rng = np.random.RandomState(2)
    K = LM.kron(LM.dense(rnd(rng, 2, 2)), LM.scalar(0.5, 3))
    S = LM.scalar(2.0, 6)
    assert orc.lm_multiply(S, K).type == orc.KRONECKER
    assert orc.lm_multiply(K, S).type == orc.KRONECKER
    assert orc.lm_multiply(K, K).type == orc.KRONECKER
    assert orc.lm_add(S, K).type == orc.KRONECKER       # kron(A, aI) + bI folds into the factor
    assert orc.lm_add(K, K).type == orc.KRONECKER       # shared factor
    assert orc.lm_multiply(S, LM.dense(rnd(rng, 6, 6))).type == orc.DENSE
    assert orc.lm_multiply(S, LM.diagonal(rnd(rng, 6))).type == orc.DIAGONAL
    assert orc.lm_multiply(S, S).type == orc.SCALAR
    assert orc.lm_add(LM.diagonal(rnd(rng, 6)), S).type == orc.DIAGONAL
    assert orc.lm_add(LM.dense(rnd(rng, 6, 6)), S).type == orc.DENSE


def test_dense_apply_and_transpose():  # dense_matrix_impl_test.cc:24-29
    rng = np.random.RandomState(0)
    A0 = rnd(rng, 2, 3)
    A = LM.dense(A0)
    x, y = rnd(rng, 3), rnd(rng, 2)
    np.testing.assert_allclose(A.apply(x), A0 @ x, atol=1e-8)
    np.testing.assert_allclose(A.T().apply(y), A0.T @ y, atol=1e-8)


def test_kronecker_apply():  # kronecker_product_impl_test.cc:9-20
    rng = np.random.RandomState(0)
    A, B, X = rnd(rng, 2, 3), rnd(rng, 4, 5), rnd(rng, 5, 3)
    C = LM.kron(LM.dense(A), LM.dense(B))
    np.testing.assert_allclose(C.apply(X.reshape(-1, order="F")),
                               (B @ X @ A.T).reshape(-1, order="F"), atol=1e-8)
    np.testing.assert_allclose(C.as_dense(), np.kron(A, B))


def test_forward_and_back_sub():  # block_cholesky_test.cc:23-58
    rng = np.random.RandomState(0)
    L0 = rnd(rng, 5, 2)
    L = BlockMatrix()
    L.set("two", "one", LM.dense(L0))
    b1, b2 = rnd(rng, 2), rnd(rng, 5)
    b = BlockVector({"one": b1, "two": b2})
    x = orc.forward_sub(L, ["one", "two"], b)
    np.testing.assert_allclose(x("one"), b1)
    np.testing.assert_allclose(x("two"), b2 - L0 @ b1)
    x = orc.back_sub(L.T(), ["one", "two"], b)
    np.testing.assert_allclose(x("one"), b1 - L0.T @ b2)
    np.testing.assert_allclose(x("two"), b2)


def test_compute_fill_exact_values():  # block_cholesky_test.cc:60-74: 4 and 25
    rng = np.random.RandomState(0)
    A0 = rnd(rng, 5, 2)
    A = BlockMatrix()
    A.set("one", "one", LM.identity(5))
    A.set("one", "two", LM.dense(A0))
    A.set("two", "one", LM.dense(A0.T))
    A.set("two", "two", LM.identity(2))
    want = GOLDEN["compute_fill"]["fill_when_eliminating"]  # 4 and 25
    assert orc.compute_fill(A, "one") == want["one"]
    assert orc.compute_fill(A, "two") == want["two"]


def test_block_cholesky_vs_dense_solve():  # block_cholesky_test.cc:76-104
    rng = np.random.RandomState(0)
    A12 = rnd(rng, 5, 2)
    A = BlockMatrix()
    A.set("one", "one", LM.scalar(10, 5))
    A.set("one", "two", LM.dense(A12))
    A.set("two", "one", LM.dense(A12.T))
    A.set("two", "two", LM.scalar(10, 2))
    b1, b2 = rnd(rng, 5), rnd(rng, 2)
    x = BlockCholesky().compute(A).solve(BlockVector({"one": b1, "two": b2}))
    A0 = 10 * np.eye(7)
    A0[:5, 5:] = A12
    A0[5:, :5] = A12.T
    x0 = np.linalg.solve(A0, np.concatenate([b1, b2]))
    np.testing.assert_allclose(x("one"), x0[:5], atol=1e-8)
    np.testing.assert_allclose(x("two"), x0[5:], atol=1e-8)


def test_block_matrix_and_vector_algebra():  # block_matrix_test.cc:27-93, block_vector_test.cc
    rng = np.random.RandomState(3)
    A = BlockMatrix()
    A.set("r1", "c1", LM.dense(rnd(rng, 3, 2)))
    A.set("r2", "c1", LM.dense(rnd(rng, 4, 2)))
    A.set("r2", "c2", LM.scalar(2.0, 4))
    assert (A.m(), A.n()) == (7, 6)
    x = BlockVector({"c1": rnd(rng, 2), "c2": rnd(rng, 4), "unused": rnd(rng, 3)})
    y = A.apply(x)  # keys of x absent from A are skipped (block_matrix.cc:155-168)
    D = A.as_dense(["r1", "r2"], ["c1", "c2"])
    np.testing.assert_allclose(np.concatenate([y("r1"), y("r2")]),
                               D @ np.concatenate([x("c1"), x("c2")]))
    AtA = A.T() @ A
    np.testing.assert_allclose(AtA.as_dense(["c1", "c2"], ["c1", "c2"]), D.T @ D)
    v = BlockVector({"a": np.ones(2)})
    v.isub(BlockVector({"a": np.ones(2), "b": np.ones(3)}))  # InsertOrAdd(key, -value)
    np.testing.assert_allclose(v("a"), 0)
    np.testing.assert_allclose(v("b"), -1)
    assert abs(v.norm() - np.sqrt(3)) < 1e-15


def test_lasso_elimination_order_fat_and_tall():
    """the fill model eliminates constraint:0, then x (Gram A A^T, m x m) for a
    fat A, and arg:0 before x (Gram A^T A, n x n) for a tall A."""
    import math
    for (m, n, expect) in [(8, 20, ["constraint:0", "var:x", "arg:0"]),
                           (20, 8, ["constraint:0", "arg:0", "var:x"])]:
        rng = np.random.RandomState(0)
        H, A = BlockMatrix(), BlockMatrix()
        H.set("arg:0", "var:x", LM.dense(rnd(rng, m, n)))
        A.set("constraint:0", "var:x", LM.identity(n))
        alpha = math.sqrt(2)
        M = (H + H.T()).scaled(alpha) + (A + A.T()) - H.left_identity() - A.left_identity()
        chol = BlockCholesky().compute(M)
        assert chol.p == expect


_ATOMS = GOLDEN["constant_atoms"]


@pytest.mark.parametrize("case", _ATOMS["cases"], ids=["%s-%d" % (c["prox"], i) for i, c in enumerate(_ATOMS["cases"])])
def test_reference_constant_atom_known_answers(case):
    """The reference's own solver-level known answers (python/epopt/constant_atoms_test.py:
    minimise / maximise atom(x) s.t. x == constant; the objective at the returned variable must be
    the atom's value at the constant within 1e-2 relative to 1 + |value|, solved with rel_tol 1e-3,
    max_iterations 10000) for every atom that is one prox function of this path, through the
    oracle's multi-block driver on the hand-compiled problem."""
    from epsilon_amd import problems, wire
    kw = {k: case[k] for k in ("k", "alpha", "beta", "arg_scale", "linear", "axis") if k in case}
    prob, c = problems.constant_atom(case["prox"], case["arg"], **kw)
    sp_ = wire.SolverParams(rel_tol=_ATOMS["rel_tol"], max_iterations=_ATOMS["max_iterations"]).SerializeToString()
    st, x = orc.solve(prob.SerializeToString(), [], sp_, prob.expression_data())
    S = wire.SolverStatus.FromString(st)
    assert S.state == wire.SolverStatus.OPTIMAL
    X = np.frombuffer(x["var:x"]).reshape(c.shape, order="F")
    val = problems.constant_atom_value(case["prox"], X, **kw)
    if case.get("maximize"):
        val = -val
    assert abs(val - case["expected"]) / (1 + abs(case["expected"])) <= _ATOMS["tolerance"], (val, case["expected"])
