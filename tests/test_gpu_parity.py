"""GPU parity tests: every case calls the HIP path through the C-ABI (epsilon_amd._solve ->
libepsilon_hip.so) and checks it against the CPU oracle on the same seeded inputs.

Tolerances: the GPU path runs in f64 (EPSILON dtype option) for tight parity with the fp64
reference semantics, and in f32 (the production dtype, BASELINE.json north_star) within the
stated fp32 tolerance.  Projection / threshold outputs are compared exactly where the
operands are exactly representable.
"""

import os

import numpy as np
import pytest
import scipy.sparse as sp

from epsilon_amd import ir, problems, wire
from epsilon_amd.wire import ProxFunction
from oracle import epsilon_oracle as orc

pytestmark = pytest.mark.gpu

TOL = {"f64": dict(rtol=1e-10, atol=1e-11), "f32": dict(rtol=2e-4, atol=2e-5)}


@pytest.fixture(params=["f64", "f32"])
def dtype(request, solve_mod):
    solve_mod.set_option("dtype", request.param)
    yield request.param
    solve_mod.set_option("dtype", "f32")


def oracle_map(lmap):
    return orc.build_linear_map(lmap.proto, lmap.data)


def make_maps(rng):
    return {
        "dense": ir.dense_matrix(rng.randn(7, 5)),
        "dense_t": ir.transpose(ir.dense_matrix(rng.randn(5, 7))),
        "diag": ir.diagonal_matrix(rng.randn(6)),
        "scalar": ir.scalar(-3.2, 6),
        "kron_dd": ir.kronecker_product(ir.dense_matrix(rng.randn(2, 3)), ir.dense_matrix(rng.randn(4, 5))),
        "kron_sd": ir.kronecker_product(ir.identity(3), ir.dense_matrix(rng.randn(4, 5))),
        "kron_ds": ir.kronecker_product(ir.transpose(ir.dense_matrix(rng.randn(3, 2))), ir.scalar(2.5, 4)),
        "big_dense": ir.dense_matrix(rng.randn(1030, 517)),
        "odd_dense": ir.dense_matrix(rng.randn(333, 1201)),
    }


@pytest.mark.parametrize("name", ["dense", "dense_t", "diag", "scalar", "kron_dd", "kron_sd",
                                  "kron_ds", "big_dense", "odd_dense"])
def test_linear_map_apply_and_adjoint(solve_mod, dtype, name):
    """reference linear/dense_matrix_impl_test.cc:24-29, kronecker_product_impl_test.cc:9-20."""
    rng = np.random.RandomState(0)
    A = make_maps(rng)[name]
    O = oracle_map(A)
    x = rng.randn(A.n)
    y = rng.randn(A.m)
    np.testing.assert_allclose(solve_mod.linear_map_apply(A, x), O.apply(x), **TOL[dtype])
    np.testing.assert_allclose(solve_mod.linear_map_apply(A, y, transpose=True), O.T().apply(y),
                               **TOL[dtype])


def square_maps(rng, n=6):
    return {
        "dense": ir.dense_matrix(rng.randn(n, n)),
        "diag": ir.diagonal_matrix(rng.randn(n)),
        "scalar": ir.scalar(1.7, n),
        "kron": ir.kronecker_product(ir.dense_matrix(rng.randn(2, 2)), ir.dense_matrix(rng.randn(3, 3))),
        "kron_s": ir.kronecker_product(ir.dense_matrix(rng.randn(2, 2)), ir.scalar(0.5, 3)),
        "sparse": ir.sparse_matrix(sp.random(n, n, density=0.4, random_state=rng, format="csc")),
        "sparse_diag": ir.sparse_matrix(sp.diags(rng.randn(n)).tocsc()),
    }


@pytest.mark.parametrize("op", ["+", "*"])
def test_linear_map_algebra_tables(solve_mod, dtype, op):
    """All pairings of {Dense, Sparse, Diagonal, Scalar, Kronecker}: values vs dense math and result
    *type* vs the oracle's restatement of the dispatch tables
    (reference linear/linear_map_test.cc:67-229)."""
    rng = np.random.RandomState(1)
    maps = square_maps(rng)
    for an, A in maps.items():
        for bn, B in maps.items():
            OA, OB = oracle_map(A), oracle_map(B)
            OC = orc.lm_add(OA, OB) if op == "+" else orc.lm_multiply(OA, OB)
            rtype, dense = solve_mod.linear_map_binary(op, A, B)
            np.testing.assert_allclose(dense, OC.as_dense(), err_msg="%s %s %s" % (an, op, bn),
                                       **TOL[dtype])
            expect = OC.type
            assert rtype == expect, "%s %s %s -> type %d, oracle %d" % (an, op, bn, rtype, expect)


def sparse_maps(rng):
    sel = sp.coo_matrix((np.ones(40), (np.arange(40), rng.permutation(300)[:40])), shape=(40, 300))
    long_rows = sp.random(5, 9000, density=0.6, random_state=rng)
    skew = sp.vstack([sp.random(200, 700, density=0.01, random_state=rng),
                      sp.csr_matrix(rng.randn(1, 700))])
    return {
        "random": sp.random(300, 500, density=0.05, random_state=rng),
        "selection": sel,                    # reference python/epopt/linear_map.py:81-92 (index)
        "transpose_matrix": sp.coo_matrix(   # linear_map.py:125-134
            (np.ones(12), (np.arange(12), np.tile(np.arange(4) * 3, 3) + np.repeat(np.arange(3), 4))),
            shape=(12, 12)),
        "empty_rows": sp.random(64, 33, density=0.02, random_state=rng),
        "all_zero": sp.csc_matrix((7, 9)),
        "long_rows": long_rows,
        "skew": skew,
        "wide_1row": sp.csr_matrix(rng.randn(1, 5000)),
        "tall_1col": sp.csr_matrix(rng.randn(5000, 1)),
    }


@pytest.mark.parametrize("name", ["random", "selection", "transpose_matrix", "empty_rows",
                                  "all_zero", "long_rows", "skew", "wide_1row", "tall_1col"])
def test_sparse_map_apply_and_adjoint(solve_mod, dtype, name):
    """reference linear/sparse_matrix_impl.h:25,27-29 (Eigen A_*x, A_.transpose()) through the
    wire format of python/epopt/constant.py:18-28 (CSC indptr | indices | data)."""
    rng = np.random.RandomState(3)
    S = sparse_maps(rng)[name]
    A = ir.sparse_matrix(S)
    x = rng.randn(S.shape[1])
    y = rng.randn(S.shape[0])
    np.testing.assert_allclose(solve_mod.linear_map_apply(A, x), S.dot(x), **TOL[dtype])
    np.testing.assert_allclose(solve_mod.linear_map_apply(A, y, transpose=True), S.T.dot(y),
                               **TOL[dtype])
    O = oracle_map(A)
    np.testing.assert_allclose(O.apply(x), S.dot(x), rtol=1e-12, atol=1e-12)


def test_sparse_products_with_dense_and_kronecker(solve_mod, dtype):
    """Rectangular Sparse x Dense, Dense x Sparse (Dense results, linear_map_multiply.cc:39-45,
    71-77) and Sparse x Kronecker (Sparse result, :103-110)."""
    rng = np.random.RandomState(4)
    S = sp.random(30, 45, density=0.1, random_state=rng)
    D1 = rng.randn(45, 17)
    D2 = rng.randn(11, 30)
    rtype, dense = solve_mod.linear_map_binary("*", ir.sparse_matrix(S), ir.dense_matrix(D1))
    assert rtype == orc.DENSE
    np.testing.assert_allclose(dense, S.dot(D1), **TOL[dtype])
    rtype, dense = solve_mod.linear_map_binary("*", ir.dense_matrix(D2), ir.sparse_matrix(S))
    assert rtype == orc.DENSE
    np.testing.assert_allclose(dense, D2.dot(S.toarray()), **TOL[dtype])
    rtype, dense = solve_mod.linear_map_binary(
        "*", ir.transpose(ir.dense_matrix(D2.T.copy())), ir.sparse_matrix(S))
    np.testing.assert_allclose(dense, D2.dot(S.toarray()), **TOL[dtype])
    K = ir.kronecker_product(ir.dense_matrix(rng.randn(9, 5)), ir.identity(5))
    rtype, dense = solve_mod.linear_map_binary("*", ir.sparse_matrix(S), K)
    assert rtype == orc.SPARSE
    np.testing.assert_allclose(dense, S.dot(oracle_map(K).as_dense()), **TOL[dtype])


def test_sparse_inverse(solve_mod, dtype):
    """reference sparse_matrix_impl.cc:60-78: scalar-like -> scalar inverse, else dense."""
    rng = np.random.RandomState(5)
    inv = solve_mod.linear_map_inverse(ir.sparse_matrix(sp.identity(6, format="csc") * 4.0))
    np.testing.assert_allclose(inv, np.eye(6) / 4.0, **TOL[dtype])
    G = sp.random(40, 60, density=0.1, random_state=rng)
    W = (G.dot(G.T) + sp.identity(40)).tocsc()
    inv = solve_mod.linear_map_inverse(ir.sparse_matrix(W))
    tol = dict(rtol=1e-9, atol=1e-10) if dtype == "f64" else dict(rtol=2e-3, atol=2e-4)
    np.testing.assert_allclose(inv, np.linalg.inv(W.toarray()), **tol)


@pytest.mark.parametrize("solver", [0, 1])
def test_sparse_lasso_matches_oracle(solve_mod, dtype, solver):
    """Lasso with a sparse design matrix: the SUM_SQUARE prox builds A A^T through the sparse
    table entries and every sweep runs two SpMVs."""
    rng = np.random.RandomState(6)
    m, n = 60, 150
    A = sp.random(m, n, density=0.15, random_state=rng, format="csc")
    x0 = np.where(rng.rand(n) < 0.1, rng.randn(n), 0)
    b = A.dot(x0) + 0.01 * rng.randn(m)
    lam = 0.1 * np.abs(A.T.dot(b)).max()
    prob = problems.lasso_ir(ir.sparse_matrix(A), ir.constant(b), lam, n)
    params = wire.SolverParams(solver=solver, max_iterations=60, rel_tol=1e-3, abs_tol=1e-5)
    sg, xg, so, xo = solve_both(solve_mod, prob, params)
    assert sg.state == so.state and sg.num_iterations == so.num_iterations
    tol = dict(rtol=1e-7, atol=1e-9) if dtype == "f64" else dict(rtol=2e-3, atol=2e-4)
    for k in xo:
        np.testing.assert_allclose(np.frombuffer(xg[k]), np.frombuffer(xo[k]), err_msg=k, **tol)


@pytest.mark.parametrize("n", [1, 5, 64, 65, 130, 300])
@pytest.mark.parametrize("sign", [1.0, -1.0])
def test_dense_inverse(solve_mod, dtype, n, sign):
    """reference linear/dense_matrix_impl.cc:21-30 (LDLT explicit inverse); the KKT Schur
    complements are +/- definite."""
    rng = np.random.RandomState(n)
    G = rng.randn(n, 2 * n + 3)
    W = sign * (np.eye(n) + G.dot(G.T) / n)
    inv = solve_mod.linear_map_inverse(ir.dense_matrix(W))
    ref = np.linalg.inv(W)
    tol = dict(rtol=1e-9, atol=1e-10) if dtype == "f64" else dict(rtol=2e-3, atol=2e-4)
    np.testing.assert_allclose(inv, ref, **tol)


@pytest.mark.parametrize("n", [1024, 1300, 2049])
def test_cached_inverse_apply_symmetric_kernel(solve_mod, dtype, n):
    """The explicit inverse of a symmetric matrix is applied by the half-traffic symmetric
    kernel (k::Symv) from n = 1024 up: y = W^-1 x against a dense solve, sizes that end in
    partial 128 x 128 tiles included."""
    rng = np.random.RandomState(n)
    G = rng.randn(n, n + 7) / np.sqrt(n)
    W = np.eye(n) + G.dot(G.T)
    x = rng.randn(n)
    got = solve_mod.linear_map_apply(ir.dense_matrix(W), x, inverse=True)
    want = np.linalg.solve(W, x)
    tol = dict(rtol=1e-9, atol=1e-10) if dtype == "f64" else dict(rtol=2e-3, atol=2e-4)
    np.testing.assert_allclose(got, want, **tol)
    # ... and from the tile-packed copy of the lower tiles that a repeatedly applied symmetric map
    # gets (every 128 x 128 tile contiguous, zero-padded at the edge): the same arithmetic in the
    # same order, so the same bits
    import os
    saved = os.environ.get("EPSILON_HIP_SYMV_PACKED")
    try:
        os.environ["EPSILON_HIP_SYMV_PACKED"] = "2"
        packed = solve_mod.linear_map_apply(ir.dense_matrix(W), x, inverse=True)
        os.environ["EPSILON_HIP_SYMV_PACKED"] = "0"
        plain = solve_mod.linear_map_apply(ir.dense_matrix(W), x, inverse=True)
    finally:
        if saved is None:
            os.environ.pop("EPSILON_HIP_SYMV_PACKED", None)
        else:
            os.environ["EPSILON_HIP_SYMV_PACKED"] = saved
    assert np.array_equal(packed, plain) and np.array_equal(plain, got)


def test_gemm_mfma_vs_oracle(solve_mod):
    """Dense*Dense through the f32 MFMA kernel (all four transpose combinations, ragged
    edges) vs numpy."""
    solve_mod.set_option("dtype", "f32")
    rng = np.random.RandomState(2)
    for (m, k, n) in [(300, 500, 260), (128, 64, 128), (129, 33, 65)]:
        for ta in (False, True):
            for tb in (False, True):
                A = rng.randn(*((k, m) if ta else (m, k)))
                B = rng.randn(*((n, k) if tb else (k, n)))
                _, C = solve_mod.linear_map_binary("*", ir.dense_matrix(A), ir.dense_matrix(B), ta, tb)
                ref = (A.T if ta else A).dot(B.T if tb else B)
                np.testing.assert_allclose(C, ref, rtol=1e-4, atol=1e-3)
    # SYRK path: A * A^T of one buffer
    A = rng.randn(200, 700)
    Am = ir.dense_matrix(A)
    _, C = solve_mod.linear_map_binary("*", Am, Am, False, True)
    np.testing.assert_allclose(C, A.dot(A.T), rtol=1e-4, atol=1e-3)
    # ... with a long contraction and more lower-triangle tiles than the chip holds at once
    # (33 * 34 / 2 = 561 = 512 + 49): the 49 tail tiles are split over K and fixed up
    A = (rng.randn(4160, 8200) / 64).astype(np.float32).astype(np.float64)
    Am = ir.dense_matrix(A)
    _, C = solve_mod.linear_map_binary("*", Am, Am, False, True)
    ref = A.dot(A.T)
    np.testing.assert_allclose(C, ref, rtol=1e-4, atol=1e-4)
    assert np.array_equal(C, C.T)


@pytest.mark.parametrize("shape", [(2300, 8200), (5700, 8200)])
def test_gram_f16_split_vs_fp64(solve_mod, shape):
    """The Gram product of a long contraction runs on the f16 matrix cores with two-term split
    operands (kernels_gemm_f16split.hip): it has to be as accurate as an f32 product - compared
    with numpy fp64 on rows of very different scale (one global scale for the split), against the
    error the exact-f32 kernel makes on the same input.  The second shape has more tiles than CUs:
    its tail tiles take the split-K path."""
    m, k = shape
    rng = np.random.RandomState(21)
    A = rng.randn(m, k) / np.sqrt(k)
    A *= np.exp(rng.uniform(-4, 4, size=(m, 1)))  # row norms over e^-4 .. e^4
    A = A.astype(np.float32).astype(np.float64)
    Am = ir.dense_matrix(A)
    ref = A.dot(A.T)
    scale = np.sqrt(np.outer(np.diag(ref), np.diag(ref)))  # |C_ij| <= scale_ij
    solve_mod.set_option("dtype", "f32")
    _, C = solve_mod.linear_map_binary("*", Am, Am, False, True)
    err_split = np.abs(C - ref) / scale
    assert np.array_equal(C, C.T)
    solve_mod.set_option("gemm", "mfma")  # forces the exact-f32 MFMA kernel
    try:
        _, C32 = solve_mod.linear_map_binary("*", Am, Am, False, True)
    finally:
        solve_mod.set_option("gemm", "auto")
    err_f32 = np.abs(C32 - ref) / scale
    # Both accumulate K terms in f32 (one rounding per term: ~sqrt(K) ulps on a sum of squares), so
    # both sit a few 1e-6 from fp64 on the Cauchy-Schwarz scale; the split must not be worse than
    # the exact-f32 kernel by more than rounding noise.
    bound = 4 * np.sqrt(k) * 2.0 ** -24
    assert err_split.max() < bound and err_f32.max() < bound, (err_split.max(), err_f32.max(), bound)
    assert err_split.max() <= 1.5 * err_f32.max() + 2e-7, (err_split.max(), err_f32.max())
    assert np.sqrt(np.mean(err_split ** 2)) <= 1.5 * np.sqrt(np.mean(err_f32 ** 2)) + 2e-8


def test_gemm_f16_split_general_and_inverse(solve_mod):
    """The split-f16 kernel as a general product (two different operands, all transpose
    combinations, ragged edges) against fp64 and against the exact-f32 kernel's error; and the
    blocked Cholesky inverse, whose large GEMMs take this kernel, at n = 4200."""
    solve_mod.set_option("dtype", "f32")
    rng = np.random.RandomState(23)
    m, k, n = 2310, 2050, 2200
    for ta in (False, True):
        for tb in (False, True):
            A = rng.randn(*((k, m) if ta else (m, k))) * np.exp(rng.uniform(-3, 3, size=((1, m) if ta else (m, 1))))
            B = rng.randn(*((n, k) if tb else (k, n))) * np.exp(rng.uniform(-3, 3, size=((n, 1) if tb else (1, n))))
            A = A.astype(np.float32).astype(np.float64)
            B = B.astype(np.float32).astype(np.float64)
            oa, ob = (A.T if ta else A), (B.T if tb else B)
            ref = oa.dot(ob)
            scale = np.outer(np.linalg.norm(oa, axis=1), np.linalg.norm(ob, axis=0))
            _, C = solve_mod.linear_map_binary("*", ir.dense_matrix(A), ir.dense_matrix(B), ta, tb)
            solve_mod.set_option("gemm", "mfma")
            try:
                _, C32 = solve_mod.linear_map_binary("*", ir.dense_matrix(A), ir.dense_matrix(B), ta, tb)
            finally:
                solve_mod.set_option("gemm", "auto")
            e16, e32 = np.abs(C - ref) / scale, np.abs(C32 - ref) / scale
            assert e16.max() < 4 * np.sqrt(k) * 2.0 ** -24, (ta, tb, e16.max())
            assert e16.max() <= 1.5 * e32.max() + 2e-7, (ta, tb, e16.max(), e32.max())
    nn = 4200
    Q = rng.randn(nn, nn + 300) / np.sqrt(nn)
    Mx = (np.eye(nn) + 2 * Q.dot(Q.T)).astype(np.float32).astype(np.float64)
    W = solve_mod.linear_map_inverse(ir.dense_matrix(Mx))
    resid = np.abs(W.dot(Mx) - np.eye(nn)).max()
    assert resid < 2e-4, resid
    Wx = np.linalg.inv(Mx)
    assert np.abs(W - Wx).max() < 1e-5 * np.abs(Wx).max() + 1e-6
    # ... of which X^T X and the doubling level of 2048-blocks run as single launches with a k
    # range per tile (triangular operands): no less accurate than the same inverse on the
    # exact-f32 kernels
    solve_mod.set_option("gemm", "mfma")
    try:
        W32 = solve_mod.linear_map_inverse(ir.dense_matrix(Mx))
    finally:
        solve_mod.set_option("gemm", "auto")
    assert np.abs(W - Wx).max() <= 1.5 * np.abs(W32 - Wx).max() + 1e-7
    assert np.array_equal(W, W.T)


def test_gemm_f16_split_general_tail_tiles(solve_mod):
    """A general product with more tiles than CUs and a long contraction: the ragged last round of
    tiles is split over K (raw partial tiles + the fix-up kernel, two different operands and row
    scale arrays), in patch order and in plain order, with every staging variant of the kernel."""
    import os
    solve_mod.set_option("dtype", "f32")
    rng = np.random.RandomState(29)
    m, k, n = 4300, 2100, 4400  # 17 x 18 = 306 tiles: 256 + a tail of 50 split 5 ways
    A = (rng.randn(m, k) * np.exp(rng.uniform(-3, 3, size=(m, 1)))).astype(np.float32).astype(np.float64)
    B = (rng.randn(k, n) * np.exp(rng.uniform(-3, 3, size=(1, n)))).astype(np.float32).astype(np.float64)
    ref = A.dot(B)
    scale = np.outer(np.linalg.norm(A, axis=1), np.linalg.norm(B, axis=0))
    results = {}
    saved = {v: os.environ.get(v) for v in ("EPSILON_HIP_GEMM_STAGE", "EPSILON_HIP_GEMM_ORDER")}
    try:
        for stage, order in (("ring", "1"), ("ring", "0"), ("lds", "1"), ("reg", "1")):
            os.environ["EPSILON_HIP_GEMM_STAGE"] = stage
            os.environ["EPSILON_HIP_GEMM_ORDER"] = order
            solve_mod.profile_enable(True)
            solve_mod.profile_reset()
            _, C = solve_mod.linear_map_binary("*", ir.dense_matrix(A), ir.dense_matrix(B), False, False)
            tags = solve_mod.profile_dump()
            solve_mod.profile_enable(False)
            assert any(t.startswith("gemm_f16split") for t in tags), list(tags)
            err = np.abs(C - ref) / scale
            assert err.max() < 4 * np.sqrt(k) * 2.0 ** -24, (stage, order, err.max())
            results[(stage, order)] = C
    finally:
        for v, old in saved.items():
            if old is None:
                os.environ.pop(v, None)
            else:
                os.environ[v] = old
    # the same multiply-adds in the same order whatever the staging (the tile order decides WHICH
    # tiles form the ragged round and are summed in K chunks, so it may change their last bits)
    first = results[("ring", "1")]
    for key, C in results.items():
        if key[1] == "1":
            assert np.array_equal(C, first), key


@pytest.mark.parametrize("mode", ["mfma", "mfma_simple", "auto"])
def test_gemm_f64_mfma_vs_numpy(solve_mod, mode):
    """Dense*Dense in fp64 on v_mfma_f64_16x16x4_f64 (its accumulator map is NOT the f32 one): all
    four transpose combinations, ragged edges, asymmetric operands, SYRK.  "mfma" / "auto": the
    software-pipelined kernel (kernels_gemm_f64.hip) where the operands allow 16-byte loads - even
    leading dimensions: edge tiles shifted back to the matrix edge (even sizes) or bounds-checked
    (odd sizes, sizes below one tile) - and the plain MFMA kernel ("mfma") or the VALU kernel
    ("auto") elsewhere; "mfma_simple": the plain MFMA kernel everywhere."""
    solve_mod.set_option("dtype", "f64")
    solve_mod.set_option("gemm", mode)
    try:
        rng = np.random.RandomState(12)
        for (m, k, n) in [(300, 500, 260), (128, 64, 128), (129, 33, 65), (64, 16, 64), (257, 19, 130),
                          (386, 1030, 258), (130, 72, 66), (512, 40, 384)]:
            for ta in (False, True):
                for tb in (False, True):
                    A = rng.randn(*((k, m) if ta else (m, k))) + np.arange(m)[None if ta else slice(None), None if not ta else slice(None)] * 0.01
                    B = rng.randn(*((n, k) if tb else (k, n)))
                    _, C = solve_mod.linear_map_binary("*", ir.dense_matrix(A), ir.dense_matrix(B), ta, tb)
                    ref = (A.T if ta else A).dot(B.T if tb else B)
                    np.testing.assert_allclose(C, ref, rtol=1e-12, atol=1e-11)
        for shape in [(200, 700), (1100, 264)]:  # (1100: 45 lower tiles on the compact grid)
            A = rng.randn(*shape)
            Am = ir.dense_matrix(A)
            _, C = solve_mod.linear_map_binary("*", Am, Am, False, True)
            np.testing.assert_allclose(C, A.dot(A.T), rtol=1e-12, atol=1e-11)
            assert np.array_equal(C, C.T)
        # the blocked Cholesky inverse: batched products with strides
        nn = 1500
        Q = rng.randn(nn, 40)
        Mx = np.eye(nn) + 2 * Q.dot(Q.T)
        W = solve_mod.linear_map_inverse(ir.dense_matrix(Mx))
        assert np.abs(W.dot(Mx) - np.eye(nn)).max() < 1e-10
    finally:
        solve_mod.set_option("gemm", "auto")
        solve_mod.set_option("dtype", "f32")


@pytest.mark.parametrize("dt", ["f32", "f64"])
def test_gemm_long_contraction_split_k(solve_mod, dt):
    """A long contraction into a small result (the X^T R of the multiclass hinge) is split over K
    into partial products that are added in a fixed order: all transpose combinations, a K that
    is not a multiple of the chunk, vs numpy."""
    solve_mod.set_option("dtype", dt)
    rng = np.random.RandomState(4)
    for (m, k, n) in [(70, 20011, 33), (200, 9000, 1), (1, 8200, 130)]:
        for ta in (False, True):
            for tb in (False, True):
                A = rng.randn(*((k, m) if ta else (m, k)))
                B = rng.randn(*((n, k) if tb else (k, n)))
                _, C = solve_mod.linear_map_binary("*", ir.dense_matrix(A), ir.dense_matrix(B), ta, tb)
                ref = (A.T if ta else A).dot(B.T if tb else B)
                tol = dict(rtol=1e-4, atol=2e-2) if dt == "f32" else dict(rtol=1e-10, atol=1e-9)
                np.testing.assert_allclose(C, ref, **tol)


def test_gemm_few_right_hand_sides(solve_mod):
    """Dense times skinny (<= 16 columns) runs as a mat-vec with several right-hand sides
    (kernels_gemv_multi.hip: the Kronecker applies of the multiclass problems), both
    orientations of the matrix, ragged row / column counts, every width class, vs numpy."""
    solve_mod.set_option("dtype", "f32")
    rng = np.random.RandomState(6)
    for (m, k, n) in [(5000, 300, 10), (4100, 257, 3), (1028, 1000, 13), (2052, 130, 16), (60000, 20, 1)]:
        for ta in (False, True):
            A = rng.randn(*((k, m) if ta else (m, k)))
            B = rng.randn(k, n)
            _, C = solve_mod.linear_map_binary("*", ir.dense_matrix(A), ir.dense_matrix(B), ta, False)
            ref = (A.T if ta else A).dot(B)
            np.testing.assert_allclose(C, ref, rtol=1e-4, atol=5e-3)
    # long contraction, few outputs (X^T R of the hinge problem)
    A, B = rng.randn(30000, 36), rng.randn(30000, 10)
    _, C = solve_mod.linear_map_binary("*", ir.dense_matrix(A), ir.dense_matrix(B), True, False)
    np.testing.assert_allclose(C, A.T.dot(B), rtol=1e-4, atol=2e-2)


# ---- proximal operators through eval_prox -----------------------------------------------------------


def run_prox(solve_mod, expr, lam, v_map, tol):
    data = expr.data
    fb = expr.proto.SerializeToString()
    vb = {k: np.asarray(v, dtype=np.float64).tobytes(order="F") for k, v in v_map.items()}
    got = solve_mod.eval_prox(fb, lam, data, vb)
    want = orc.eval_prox(fb, lam, data, vb)
    assert set(got) == set(want)
    for k in want:
        np.testing.assert_allclose(np.frombuffer(got[k]), np.frombuffer(want[k]), err_msg=k, **tol)
    return got


def prox_cases(rng, n=10):
    x = ir.variable(n, 1, "var:x")
    X = ir.variable(n, 3, "var:X")
    w = rng.randn(n)
    w[0] = 0
    A20, b20 = rng.randn(20, n), rng.randn(20)
    A5, b5 = rng.randn(5, n), rng.randn(5)
    q_a = ir.constant(np.abs(rng.randn(n)))
    q_b = ir.constant(np.abs(rng.randn(n)))
    sz_q = wire.ProxScaledZoneParams(alpha_expr=q_a.proto, beta_expr=q_b.proto)
    qdata = dict(q_a.data)
    qdata.update(q_b.data)
    cases = {
        "norm_1": ir.prox(ProxFunction.NORM_1, x),
        "norm_1_weighted": ir.prox(ProxFunction.NORM_1, ir.linear_map(ir.diagonal_matrix(w), x)),
        "norm_1_scaled": ir.prox(ProxFunction.NORM_1, ir.linear_map(ir.scalar(-2.5, n), x), alpha=0.7),
        "deadzone": ir.prox(ProxFunction.SUM_DEADZONE, x,
                            scaled_zone_params=wire.ProxScaledZoneParams(m=0.4)),
        "hinge": ir.prox(ProxFunction.SUM_HINGE, x),
        "hinge_1mx": ir.prox(ProxFunction.SUM_HINGE,
                             ir.add(ir.linear_map(ir.scalar(-1, n), x), ir.scalar_constant(1.0, (n, 1)))),
        "quantile": ir.prox(ProxFunction.SUM_QUANTILE, x, scaled_zone_params=sz_q, data=qdata),
        "norm_2": ir.prox(ProxFunction.NORM_2, x),
        "norm_2_fro": ir.prox(ProxFunction.NORM_2, ir.reshape(X, 3 * n, 1), arg_size=[(3 * n, 1)]),
        "non_negative": ir.prox(ProxFunction.NON_NEGATIVE, x),
        "non_negative_scaled": ir.prox(ProxFunction.NON_NEGATIVE, ir.linear_map(ir.scalar(-1.3, n), x)),
        "non_negative_elemwise": ir.prox(ProxFunction.NON_NEGATIVE,
                                         ir.linear_map(ir.diagonal_matrix(rng.randn(n)), x)),
        "sum_square_20": ir.prox(ProxFunction.SUM_SQUARE,
                                 ir.add(ir.linear_map(ir.dense_matrix(A20), x),
                                        ir.linear_map(ir.scalar(-1, 20), ir.constant(b20)))),
        "sum_square_5": ir.prox(ProxFunction.SUM_SQUARE,
                                ir.add(ir.linear_map(ir.dense_matrix(A5), x),
                                       ir.linear_map(ir.scalar(-1, 5), ir.constant(b5)))),
        "sum_square_matrix": ir.prox(ProxFunction.SUM_SQUARE,
                                     ir.add(ir.linear_map(ir.left_matrix_product(ir.dense_matrix(A20), 3),
                                                          ir.reshape(X, 3 * n, 1)),
                                            ir.linear_map(ir.scalar(-1, 60), ir.constant(rng.randn(60))))),
        "zero": ir.prox(ProxFunction.ZERO,
                        ir.add(ir.linear_map(ir.dense_matrix(A5), x),
                               ir.linear_map(ir.scalar(-1, 5), ir.constant(A5.dot(rng.randn(n)))))),
        "affine": ir.prox(ProxFunction.AFFINE, ir.linear_map(ir.dense_matrix(rng.randn(1, n)), x)),
        "tv_1d": ir.prox(ProxFunction.TOTAL_VARIATION_1D, x),
    }
    return cases


PROX_NAMES = ["norm_1", "norm_1_weighted", "norm_1_scaled", "deadzone", "hinge", "hinge_1mx",
              "quantile", "norm_2", "norm_2_fro", "non_negative", "non_negative_scaled",
              "non_negative_elemwise", "sum_square_20", "sum_square_5", "sum_square_matrix",
              "zero", "affine", "tv_1d"]


@pytest.mark.parametrize("name", PROX_NAMES)
def test_eval_prox_vs_oracle(solve_mod, dtype, name):
    """The prox cases of reference python/epopt/prox_test.py:168-221 that this build covers,
    3 seeded trials each, random lambda as prox_test.py:279-283."""
    for trial in range(3):
        rng = np.random.RandomState(trial)
        expr = prox_cases(rng)[name]
        lam = abs(rng.randn()) + 0.05
        v_map = {vid: rng.randn(sz[0] * sz[1]) for vid, sz in ir.get_variables(expr.proto).items()}
        tol = TOL[dtype] if dtype == "f64" else dict(rtol=5e-4, atol=5e-5)
        run_prox(solve_mod, expr, lam, v_map, tol)


def test_projections_are_bit_exact(solve_mod):
    """max(v,0) and the dead-zone / threshold branches on f32-representable inputs must match
    the oracle bit for bit (north_star: bit-exact for indexing/projection ops)."""
    solve_mod.set_option("dtype", "f32")
    rng = np.random.RandomState(3)
    n = 4099
    v = rng.randn(n).astype(np.float32).astype(np.float64)
    x = ir.variable(n, 1, "var:x")
    got = run_prox(solve_mod, ir.prox(ProxFunction.NON_NEGATIVE, x), 1.0, {"var:x": v},
                   dict(rtol=0, atol=0))
    assert np.array_equal(np.frombuffer(got["var:x"]), np.maximum(v, 0))
    # soft threshold with lam = 1 (all scalings are exactly 1): the f32 result is exact
    got = run_prox(solve_mod, ir.prox(ProxFunction.NORM_1, x), 1.0, {"var:x": v}, dict(rtol=0, atol=0))
    v32 = v.astype(np.float32)
    ref = (np.sign(v32) * np.maximum(np.abs(v32) - np.float32(1), np.float32(0))).astype(np.float64)
    assert np.array_equal(np.frombuffer(got["var:x"]), ref)


# ---- ADMM drivers -------------------------------------------------------------------------------------


def solve_both(solve_mod, prob, params):
    pb, sb, data = prob.SerializeToString(), params.SerializeToString(), prob.expression_data()
    st_g, x_g = solve_mod.solve(pb, [], sb, data)
    st_o, x_o = orc.solve(pb, [], sb, data)
    return wire.SolverStatus.FromString(st_g), x_g, wire.SolverStatus.FromString(st_o), x_o


@pytest.mark.parametrize("solver", [0, 1])
@pytest.mark.parametrize("shape", [(200, 500), (40, 15)])
def test_lasso_iterates_match_oracle(solve_mod, dtype, solver, shape):
    """BASELINE.json configs[0]: lasso 200x500 (fat: eliminates to the m x m Gram A A^T) and a
    tall 40x15 instance (the other elimination order, n x n Gram), both drivers."""
    prob, info = problems.lasso(shape[0], shape[1], seed=0)
    params = wire.SolverParams(solver=solver)
    sg, xg, so, xo = solve_both(solve_mod, prob, params)
    assert sg.state == so.state == wire.SolverStatus.OPTIMAL
    assert sg.num_iterations == so.num_iterations
    rt = 1e-8 if dtype == "f64" else 2e-3
    for f in ("r_norm", "s_norm", "epsilon_primal", "epsilon_dual"):
        np.testing.assert_allclose(getattr(sg.residuals, f), getattr(so.residuals, f), rtol=rt, atol=1e-9 if dtype == "f64" else 1e-5)
    tol = dict(rtol=1e-8, atol=1e-10) if dtype == "f64" else dict(rtol=1e-3, atol=1e-4)
    for k in xo:
        np.testing.assert_allclose(np.frombuffer(xg[k]), np.frombuffer(xo[k]), err_msg=k, **tol)


def test_lasso_fixed_sweeps_trace(solve_mod, dtype):
    """Iterate parity sweep by sweep: run exactly k sweeps on the GPU (solver handle) and in
    the oracle and compare x after sweeps 1, 2, 3, 10."""
    prob, info = problems.lasso(60, 150, seed=1)
    pb, data = prob.SerializeToString(), prob.expression_data()
    for k in (1, 2, 3, 10):
        params = wire.SolverParams(max_iterations=k)
        sb = params.SerializeToString()
        sg, xg, so, xo = solve_both(solve_mod, prob, params)
        assert sg.num_iterations == so.num_iterations == k
        tol = dict(rtol=1e-9, atol=1e-11) if dtype == "f64" else dict(rtol=1e-3, atol=1e-4)
        for key in xo:
            np.testing.assert_allclose(np.frombuffer(xg[key]), np.frombuffer(xo[key]), **tol)


def test_solver_handle_staged_run_equals_one_shot(solve_mod, dtype):
    prob, info = problems.lasso(50, 120, seed=2)
    pb, data = prob.SerializeToString(), prob.expression_data()
    sb = wire.SolverParams().SerializeToString()
    st1, x1 = solve_mod.solve(pb, [], sb, data)
    s = solve_mod.Solver(pb, sb, data)
    s.init()
    total = 0
    while True:
        done = s.run(7)
        total += done
        if done < 7:
            break
    st2, x2 = s.result()
    s.close()
    a, b = wire.SolverStatus.FromString(st1), wire.SolverStatus.FromString(st2)
    assert a.state == b.state and a.num_iterations == b.num_iterations
    for k in x1:
        assert np.array_equal(np.frombuffer(x1[k]), np.frombuffer(x2[k]))


def test_tv_1d_problem(solve_mod, dtype):
    prob, info = problems.tv_1d(200, seed=0)
    sg, xg, so, xo = solve_both(solve_mod, prob, wire.SolverParams())
    assert sg.state == so.state
    assert sg.num_iterations == so.num_iterations
    tol = dict(rtol=1e-7, atol=1e-9) if dtype == "f64" else dict(rtol=2e-3, atol=2e-3)
    for k in xo:
        np.testing.assert_allclose(np.frombuffer(xg[k]), np.frombuffer(xo[k]), **tol)


def test_multiclass_hinge_problem(solve_mod, dtype):
    """Kronecker constraint maps + AFFINE / NON_NEGATIVE / SUM_SQUARE terms (config 4 shape)."""
    X, Y = problems.multiclass_hinge_data(30, 8, 3, seed=0)
    prob, info = problems.multiclass_hinge(X, Y, lam=0.1)
    params = wire.SolverParams(max_iterations=40)
    sg, xg, so, xo = solve_both(solve_mod, prob, params)
    assert sg.num_iterations == so.num_iterations
    tol = dict(rtol=1e-7, atol=1e-9) if dtype == "f64" else dict(rtol=5e-3, atol=5e-3)
    for k in xo:
        np.testing.assert_allclose(np.frombuffer(xg[k]), np.frombuffer(xo[k]), err_msg=k, **tol)


@pytest.mark.parametrize("name", ["basis_pursuit", "least_abs_dev", "hinge_l1", "quantile", "lp"])
def test_lp_type_problems(solve_mod, dtype, name):
    """Graph-form problems over ZERO / AFFINE / NON_NEGATIVE / NORM_1 / SUM_HINGE / SUM_QUANTILE
    (reference solve_test.py:26-48): same stopping iteration and iterates as the oracle."""
    prob, info = {"basis_pursuit": lambda: problems.basis_pursuit(10, 30),
                  "least_abs_dev": lambda: problems.least_abs_dev(30, 5),
                  "hinge_l1": lambda: problems.hinge_l1(40, 10),
                  "quantile": lambda: problems.quantile(40, 3),
                  "lp": lambda: problems.lp(20, 8)}[name]()
    params = wire.SolverParams(max_iterations=80)
    sg, xg, so, xo = solve_both(solve_mod, prob, params)
    assert sg.num_iterations == so.num_iterations and sg.state == so.state
    tol = dict(rtol=1e-6, atol=1e-8) if dtype == "f64" else dict(rtol=5e-3, atol=5e-3)
    for k in xo:
        np.testing.assert_allclose(np.frombuffer(xg[k]), np.frombuffer(xo[k]), err_msg=k, **tol)


@pytest.mark.parametrize("name", ["group_lasso", "logreg_l1", "covsel", "mv_lasso", "fused_lasso"])
def test_more_benchmark_problems(solve_mod, dtype, name):
    """Drivers over the batched NORM_2 (group lasso), SUM_LOGISTIC + ZERO graph form (l1 logistic
    regression), NEG_LOG_DET (sparse inverse covariance), the Kronecker data map of a matrix
    variable (multivariate lasso) and three terms on one variable (fused lasso: least squares +
    l1 + total variation): same stopping iteration and iterates as the oracle."""
    prob, info = {"group_lasso": lambda: problems.group_lasso(30, 20, 3),
                  "mv_lasso": lambda: problems.mv_lasso(30, 40, 3, rho=0.1),
                  "fused_lasso": lambda: problems.fused_lasso(30, 4, 12, rho=0.3),
                  "logreg_l1": lambda: problems.logreg_l1(40, 15),
                  "covsel": lambda: problems.covsel(6)}[name]()
    params = wire.SolverParams(max_iterations=60)
    sg, xg, so, xo = solve_both(solve_mod, prob, params)
    assert sg.num_iterations == so.num_iterations and sg.state == so.state
    tol = dict(rtol=1e-6, atol=1e-8) if dtype == "f64" else dict(rtol=5e-3, atol=5e-3)
    for k in xo:
        np.testing.assert_allclose(np.frombuffer(xg[k]), np.frombuffer(xo[k]), err_msg=k, **tol)


@pytest.mark.parametrize("name,solver_id", [("mv_lasso", 0), ("hinge_l1", 0), ("group_lasso", 0), ("lp", 0),
                                            ("least_abs_dev", 0), ("mv_lasso", 1), ("hinge_l1", 1)])
def test_generic_path_graph_replay_is_bit_identical(solve_mod, dtype, name, solver_id):
    """The sweeps of the generic operator path between two residual checks CAN be replayed from one
    hipGraph (admm.cc: canonical state buffers, copies back at the end of the captured batch, a
    buffer hold around everything the capture touches; off by default - it measured no faster than
    eager launches): the same kernels with the same arguments, so the iterates, the residuals and
    the stopping sweep are those of the eager launches, bit for bit; the replay must really have
    happened."""
    import os
    prob, info = {"mv_lasso": lambda: problems.mv_lasso(30, 40, 3, rho=0.1),
                  "hinge_l1": lambda: problems.hinge_l1(40, 10),
                  "group_lasso": lambda: problems.group_lasso(30, 20, 3),
                  "lp": lambda: problems.lp(20, 8),
                  "least_abs_dev": lambda: problems.least_abs_dev(30, 5)}[name]()
    pb, data = prob.SerializeToString(), prob.expression_data()
    # (the graph is built once a run has lasted 50 sweeps: batches of 10 and a ragged last one follow)
    sb = wire.SolverParams(max_iterations=135, solver=solver_id, ignore_stopping_criteria=True).SerializeToString()
    solve_mod.set_option("dtype", dtype)
    out = {}
    try:
        for mode in ("1", "0"):
            solve_mod.set_option("graph_generic", mode)
            solve_mod.graph_stats(reset=True)
            st, x = solve_mod.solve(pb, [], sb, data)
            out[mode] = (wire.SolverStatus.FromString(st), x, solve_mod.graph_stats())
    finally:
        solve_mod.set_option("graph_generic", "0")
        solve_mod.set_option("dtype", "f32")
    (sg, xg, (replayed, captures)), (se, xe, (replayed_e, _)) = out["1"], out["0"]
    assert replayed_e == 0
    assert replayed >= 20 and captures >= 1, (replayed, captures)
    assert sg.num_iterations == se.num_iterations and sg.state == se.state
    for f in ("r_norm", "s_norm", "epsilon_primal", "epsilon_dual"):
        assert getattr(sg.residuals, f) == getattr(se.residuals, f), f
    for k in xe:
        assert xg[k] == xe[k], k


def _atom_cases():
    import json
    g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden",
                                    "reference_known_answers.json")))["constant_atoms"]
    return g, g["cases"]


@pytest.mark.parametrize("solver_id", [0, 1])
def test_reference_constant_atom_known_answers(solve_mod, dtype, solver_id):
    """The reference's own solver-level known answers (python/epopt/constant_atoms_test.py: the
    objective of `minimise atom(x) s.t. x == constant` at the returned variable must be the atom's
    value at the constant within 1e-2 relative to 1 + |value|; rel_tol 1e-3, max_iterations 10000)
    for every atom that is one prox function of this path - 67 cases, through the C ABI, both
    drivers, both arithmetic types - and the same stopping sweep as the oracle."""
    g, cases = _atom_cases()
    sp = wire.SolverParams(rel_tol=g["rel_tol"], max_iterations=g["max_iterations"], solver=solver_id)
    solve_mod.set_option("dtype", dtype)
    try:
        for case in cases:
            kw = {k: case[k] for k in ("k", "alpha", "beta", "arg_scale", "linear", "axis") if k in case}
            prob, c = problems.constant_atom(case["prox"], case["arg"], **kw)
            sg, xg, so, xo = solve_both(solve_mod, prob, sp)
            assert sg.state == wire.SolverStatus.OPTIMAL, case
            X = np.frombuffer(xg["var:x"]).reshape(c.shape, order="F")
            val = problems.constant_atom_value(case["prox"], X, **kw)
            if case.get("maximize"):
                val = -val
            assert abs(val - case["expected"]) / (1 + abs(case["expected"])) <= g["tolerance"], (case, val)
            if dtype == "f64":
                assert sg.num_iterations == so.num_iterations, (case, sg.num_iterations, so.num_iterations)
    finally:
        solve_mod.set_option("dtype", "f32")


def test_error_reporting(solve_mod):
    """A failed CHECK surfaces as _solve.error with a message (reference: longjmp ->
    _solve.error("CHECK failed"), solvemodule.cc:245-248)."""
    with pytest.raises(solve_mod.error):
        solve_mod.solve(b"\x0a\x05\x08", [], b"", {})  # truncated length-delimited field
    x = ir.variable(5, 1, "var:x")
    bad = ir.prox(ProxFunction.NORM_NUCLEAR + 2, x)  # SIGMA_MAX: no operator registered
    with pytest.raises(solve_mod.error):
        solve_mod.eval_prox(bad.proto.SerializeToString(), 1.0, {}, {"var:x": np.zeros(5).tobytes()})


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_hip_solve_matches_single_gpu(solve_mod, tmp_path, world):
    """The real HIP sharded path (column slabs, all-reduce of the A u partials and of the
    Gram / residual partial sums) with `world` processes sharing this GPU, collectives through
    the host-callback backend over gloo; must reproduce the single-process oracle."""
    from tests import mp_util
    m, n = 40, 101
    x0, x1, status, parts = mp_util.run_ranks(world, "hip", str(tmp_path), m, n, seed=3,
                                              env_extra={"EPS_TEST_DTYPE": "f64"})
    prob, info = problems.lasso(m, n, seed=3)
    st, x = orc.solve(prob.SerializeToString(), [], wire.SolverParams().SerializeToString(),
                      prob.expression_data())
    S = wire.SolverStatus.FromString(st)
    for s, p in zip(status, parts):
        assert int(p["state"]) == wire.SolverStatus.OPTIMAL
        assert int(s[0]) == S.num_iterations
        np.testing.assert_allclose(s[1:], [S.residuals.r_norm, S.residuals.s_norm,
                                           S.residuals.epsilon_primal, S.residuals.epsilon_dual],
                                   rtol=1e-8)
    np.testing.assert_allclose(x0, np.frombuffer(x["separate:var:x:sum_square"]), rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(x1, np.frombuffer(x["var:x"]), rtol=1e-8, atol=1e-10)


@pytest.mark.parametrize("dt", ["f64", "f32"])
def test_sharded_solve_distributed_inverse(solve_mod, tmp_path, dt):
    """From 1024 rows the explicit inverse of the replicated Schur complement is split over the
    ranks (Cholesky + triangular solves for a slab of columns each, all-gather): 3 ranks sharing
    this GPU against the single-process oracle."""
    from tests import mp_util
    m, n = 1100, 2300
    x0, x1, status, parts = mp_util.run_ranks(3, "hip", str(tmp_path), m, n, seed=5,
                                              env_extra={"EPS_TEST_DTYPE": dt,
                                                         "EPSILON_HIP_DIST_INVERSE": "2"})
    prob, info = problems.lasso(m, n, seed=5)
    st, x = orc.solve(prob.SerializeToString(), [], wire.SolverParams().SerializeToString(),
                      prob.expression_data())
    S = wire.SolverStatus.FromString(st)
    for s, p in zip(status, parts):
        assert int(p["state"]) == wire.SolverStatus.OPTIMAL and int(s[0]) == S.num_iterations
    tol = dict(rtol=1e-7, atol=1e-9) if dt == "f64" else dict(rtol=2e-3, atol=2e-4)
    np.testing.assert_allclose(x0, np.frombuffer(x[problems.LASSO_COPY]), **tol)
    np.testing.assert_allclose(x1, np.frombuffer(x[problems.LASSO_VAR]), **tol)


@pytest.mark.parametrize("world,dt", [(2, "f64"), (3, "f64"), (3, "f32")])
def test_consensus_form_solve(solve_mod, tmp_path, world, dt):
    """Consensus form (mode E2): every rank holds its own term f_g (a row block of A), x_g and
    the consensus row sharded, z replicated and averaged by one all-reduce of n entries per
    sweep.  `world` processes sharing this GPU against the oracle's single-process solve of the
    stacked problem: same stopping sweep, residuals and iterates."""
    from tests import mp_util
    m, n = 61, 23
    x0, x1, status, parts = mp_util.run_ranks(world, "hip_consensus", str(tmp_path), m, n, seed=4,
                                              env_extra={"EPS_TEST_DTYPE": dt})
    A, b = problems.regression_data(m, n, seed=4)
    lam = 0.3 * np.abs(A.T.dot(b)).max()
    prob = problems.consensus_lasso(A, b, lam, world)
    st, x = orc.solve(prob.SerializeToString(), [], wire.SolverParams().SerializeToString(),
                      prob.expression_data())
    S = wire.SolverStatus.FromString(st)
    tol = dict(rtol=1e-8, atol=1e-10) if dt == "f64" else dict(rtol=2e-3, atol=2e-4)
    for g, (s, p) in enumerate(zip(status, parts)):
        assert int(p["state"]) == wire.SolverStatus.OPTIMAL
        assert int(s[0]) == S.num_iterations
        np.testing.assert_allclose(s[1:], [S.residuals.r_norm, S.residuals.s_norm,
                                           S.residuals.epsilon_primal, S.residuals.epsilon_dual],
                                   rtol=1e-8 if dt == "f64" else 5e-3)
        np.testing.assert_allclose(p["x0"], np.frombuffer(x["var:x_%d" % g]), **tol)
        np.testing.assert_allclose(p["x1"], np.frombuffer(x[problems.CONSENSUS_Z]), **tol)


@pytest.mark.parametrize("world,dt,nn", [(2, "f64", 24), (3, "f64", 41), (3, "f32", 70)])
def test_robust_pca_row_sharded(solve_mod, tmp_path, world, dt, nn):
    """configs[4] in its sharded form: the matrix split by rows over the ranks, the nuclear-norm
    prox running the row-sharded block Jacobi SVD (panel Grams and column norms all-reduced, V
    and the singular values replicated).  `world` processes sharing this GPU against the oracle's
    single-process solve: same stopping sweep, residuals and iterates."""
    from tests import mp_util
    x0, x1, status, parts = mp_util.run_ranks(world, "hip_rpca", str(tmp_path), nn, 8, seed=2,
                                              max_iter=400, env_extra={"EPS_TEST_DTYPE": dt})
    M = problems.robust_pca_data(nn, r=3, density=0.1, seed=2)
    prob = problems.robust_pca_ir(M, 0.1)
    st, x = orc.solve(prob.SerializeToString(), [], wire.SolverParams(max_iterations=400).SerializeToString(),
                      prob.expression_data())
    S = wire.SolverStatus.FromString(st)
    Lref = np.frombuffer(x["var:L"]).reshape(nn, nn, order="F")
    Sref = np.frombuffer(x["var:S"]).reshape(nn, nn, order="F")
    tol = dict(rtol=1e-6, atol=1e-7) if dt == "f64" else dict(rtol=5e-3, atol=5e-3)
    for s, p in zip(status, parts):
        assert int(p["state"]) == S.state
        assert int(s[0]) == S.num_iterations
        np.testing.assert_allclose(s[1:], [S.residuals.r_norm, S.residuals.s_norm,
                                           S.residuals.epsilon_primal, S.residuals.epsilon_dual],
                                   rtol=1e-5 if dt == "f64" else 2e-2)
    np.testing.assert_allclose(x0, Lref, **tol)   # run_ranks stacks the row blocks
    np.testing.assert_allclose(x1, Sref, **tol)


def test_rccl_backend_single_rank(solve_mod):
    """RCCL backend end to end (dlopen, unique id, ncclCommInitRank, ncclAllReduce on the solver
    stream) on a 1-rank communicator with the sharded code path forced on."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, "rccl_single_rank.py")],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


@pytest.mark.parametrize("shape", [(200, 500), (700, 1501), (4100, 4301), (8, 21), (1500, 701),
                                   (10244, 10260)])   # past the 10240 rows that 256 threads hold
@pytest.mark.parametrize("kind", ["lasso", "deadzone", "quantile"])
def test_fused_sweep_matches_generic_and_oracle(solve_mod, shape, kind):
    """The one-pass fused sweep (kernels_fused.hip) against the unfused operator path (same
    library, fused=0) and the oracle: same stopping sweep, same residuals, iterates within fp32
    rounding of each other."""
    solve_mod.set_option("dtype", "f32")
    m, n = shape
    if kind == "lasso":
        prob, info = problems.lasso(m, n, seed=5)
    else:
        A, b = problems.regression_data(m, n, seed=5)
        lam = 0.3 * np.abs(A.T.dot(b)).max()
        x = ir.variable(n, 1, problems.LASSO_COPY)
        y = ir.variable(n, 1, problems.LASSO_VAR)
        f0 = ir.prox(ProxFunction.SUM_SQUARE, ir.add(ir.linear_map(ir.dense_matrix(A), x),
                                                     ir.linear_map(ir.scalar(-1, m), ir.constant(b))))
        if kind == "quantile":  # per-element alpha / beta from data vectors (scaled_zone.cc:34-76)
            rq = np.random.RandomState(9)
            qa, qb = ir.constant(0.2 + rq.rand(n)), ir.constant(0.2 + rq.rand(n))
            qd = dict(qa.data)
            qd.update(qb.data)
            f1 = ir.prox(ProxFunction.SUM_QUANTILE, y, alpha=lam, data=qd,
                         scaled_zone_params=wire.ProxScaledZoneParams(alpha_expr=qa.proto, beta_expr=qb.proto))
        else:
            f1 = ir.prox(ProxFunction.SUM_DEADZONE, ir.linear_map(ir.scalar(2.0, n), y), alpha=lam,
                         scaled_zone_params=wire.ProxScaledZoneParams(m=0.05))
        prob = ir.Problem([f0, f1], [ir.zero(ir.add(x, ir.linear_map(ir.scalar(-1, n), y)))])
    pb, data = prob.SerializeToString(), prob.expression_data()
    sb = wire.SolverParams(max_iterations=200).SerializeToString()
    try:
        solve_mod.set_option("fused", "1")
        solve_mod.profile_reset()
        solve_mod.profile_enable(True)
        st_f, x_f = solve_mod.solve(pb, [], sb, data)
        tags = solve_mod.profile_dump()
        solve_mod.profile_enable(False)
        solve_mod.set_option("fused", "0")
        st_g, x_g = solve_mod.solve(pb, [], sb, data)
    finally:
        solve_mod.set_option("fused", "1")
    took_fused = any(t.startswith("lasso_fused") for t in tags)
    # fat problems eliminate to the m x m Gram and match the fused pattern; a tall one (m > n)
    # eliminates the argument row first (n x n Gram) and must stay on the generic path
    assert took_fused == (m < n), "fused=%s for %dx%d: %s" % (took_fused, m, n, list(tags))
    st_o, x_o = orc.solve(pb, [], sb, data)
    sf, sg, so = (wire.SolverStatus.FromString(s) for s in (st_f, st_g, st_o))
    assert sf.state == sg.state == so.state
    assert sf.num_iterations == sg.num_iterations == so.num_iterations
    for f in ("r_norm", "s_norm", "epsilon_primal", "epsilon_dual"):
        np.testing.assert_allclose(getattr(sf.residuals, f), getattr(so.residuals, f), rtol=3e-3, atol=1e-5)
    for k in x_o:
        np.testing.assert_allclose(np.frombuffer(x_f[k]), np.frombuffer(x_g[k]), rtol=1e-4, atol=2e-5, err_msg=k)
        np.testing.assert_allclose(np.frombuffer(x_f[k]), np.frombuffer(x_o[k]), rtol=1e-3, atol=1e-4, err_msg=k)


@pytest.mark.parametrize("shape", [(200, 500), (700, 1501), (8, 21), (4100, 4301)])
@pytest.mark.parametrize("kind", ["lasso", "deadzone", "quantile"])
def test_fused_sweep_two_block_driver(solve_mod, shape, kind):
    """The same one-pass sweep for the TWO_BLOCK driver (prox_admm_two_block.cc:96-133): Jacobi
    x-updates, closed-form projection onto the consensus constraint, dual ascent - against the
    unfused operator path and the oracle."""
    solve_mod.set_option("dtype", "f32")
    m, n = shape
    if kind == "lasso":
        prob, info = problems.lasso(m, n, seed=7)
    else:
        A, b = problems.regression_data(m, n, seed=7)
        lam = 0.3 * np.abs(A.T.dot(b)).max()
        x = ir.variable(n, 1, problems.LASSO_COPY)
        y = ir.variable(n, 1, problems.LASSO_VAR)
        f0 = ir.prox(ProxFunction.SUM_SQUARE, ir.add(ir.linear_map(ir.dense_matrix(A), x),
                                                     ir.linear_map(ir.scalar(-1, m), ir.constant(b))))
        if kind == "quantile":
            rq = np.random.RandomState(9)
            qa, qb = ir.constant(0.2 + rq.rand(n)), ir.constant(0.2 + rq.rand(n))
            qd = dict(qa.data)
            qd.update(qb.data)
            f1 = ir.prox(ProxFunction.SUM_QUANTILE, y, alpha=lam, data=qd,
                         scaled_zone_params=wire.ProxScaledZoneParams(alpha_expr=qa.proto, beta_expr=qb.proto))
        else:
            f1 = ir.prox(ProxFunction.SUM_DEADZONE, ir.linear_map(ir.scalar(2.0, n), y), alpha=lam,
                         scaled_zone_params=wire.ProxScaledZoneParams(m=0.05))
        prob = ir.Problem([f0, f1], [ir.zero(ir.add(ir.linear_map(ir.scalar(2.0, n), x),
                                                   ir.linear_map(ir.scalar(-2.0, n), y)))])
    pb, data = prob.SerializeToString(), prob.expression_data()
    sb = wire.SolverParams(max_iterations=300, solver=1).SerializeToString()
    try:
        solve_mod.set_option("fused", "1")
        solve_mod.profile_reset()
        solve_mod.profile_enable(True)
        st_f, x_f = solve_mod.solve(pb, [], sb, data)
        tags = solve_mod.profile_dump()
        solve_mod.profile_enable(False)
        solve_mod.set_option("fused", "0")
        st_g, x_g = solve_mod.solve(pb, [], sb, data)
    finally:
        solve_mod.set_option("fused", "1")
    assert any(t.startswith("lasso_fused") for t in tags), sorted(tags)
    st_o, x_o = orc.solve(pb, [], sb, data)
    sf, sg, so = (wire.SolverStatus.FromString(s) for s in (st_f, st_g, st_o))
    assert sf.state == sg.state == so.state
    assert sf.num_iterations == sg.num_iterations == so.num_iterations
    for f in ("r_norm", "s_norm", "epsilon_primal", "epsilon_dual"):
        np.testing.assert_allclose(getattr(sf.residuals, f), getattr(so.residuals, f), rtol=3e-3, atol=1e-5)
    for k in x_o:
        np.testing.assert_allclose(np.frombuffer(x_f[k]), np.frombuffer(x_g[k]), rtol=1e-4, atol=2e-5, err_msg=k)
        np.testing.assert_allclose(np.frombuffer(x_f[k]), np.frombuffer(x_o[k]), rtol=1e-3, atol=1e-4, err_msg=k)


@pytest.mark.parametrize("shape", [(200, 500), (700, 1501), (2, 7), (1030, 2049), (5200, 5301)])
def test_fused_sweep_fp64(solve_mod, shape):
    """The fused pass in fp64 (the reference's arithmetic type; two rows per 16-byte load, 512-thread
    workgroups above 5120 rows): same stopping sweep and residuals as the oracle, iterates to fp64
    rounding of both the unfused operator path and the oracle."""
    m, n = shape
    prob, info = problems.lasso(m, n, seed=6)
    pb, data = prob.SerializeToString(), prob.expression_data()
    sb = wire.SolverParams(max_iterations=200).SerializeToString()
    solve_mod.set_option("dtype", "f64")
    try:
        solve_mod.set_option("fused", "1")
        solve_mod.profile_reset()
        solve_mod.profile_enable(True)
        st_f, x_f = solve_mod.solve(pb, [], sb, data)
        tags = solve_mod.profile_dump()
        solve_mod.profile_enable(False)
        solve_mod.set_option("fused", "0")
        st_g, x_g = solve_mod.solve(pb, [], sb, data)
    finally:
        solve_mod.set_option("fused", "1")
        solve_mod.set_option("dtype", "f32")
    assert any(t.startswith("lasso_fused:%dx%d" % (m, n)) for t in tags), sorted(tags)
    st_o, x_o = orc.solve(pb, [], sb, data)
    sf, sg, so = (wire.SolverStatus.FromString(s) for s in (st_f, st_g, st_o))
    assert sf.state == sg.state == so.state == wire.SolverStatus.OPTIMAL
    assert sf.num_iterations == sg.num_iterations == so.num_iterations
    for f in ("r_norm", "s_norm", "epsilon_primal", "epsilon_dual"):
        np.testing.assert_allclose(getattr(sf.residuals, f), getattr(so.residuals, f), rtol=1e-8, atol=1e-11)
    for k in x_o:
        np.testing.assert_allclose(np.frombuffer(x_f[k]), np.frombuffer(x_g[k]), rtol=1e-9, atol=1e-11, err_msg=k)
        np.testing.assert_allclose(np.frombuffer(x_f[k]), np.frombuffer(x_o[k]), rtol=1e-8, atol=1e-10, err_msg=k)


@pytest.mark.parametrize("solver_id", [0, 1])
@pytest.mark.parametrize("kind", ["lasso", "quantile"])
@pytest.mark.parametrize("shape", [(200, 500), (700, 1501), (5200, 5301)])
def test_fused_sweep_fp64_two_block_and_quantile(solve_mod, shape, kind, solver_id):
    """Round 3: the fp64 fused pass also carries the TWO_BLOCK driver's chain
    (prox_admm_two_block.cc:96-133) and per-column alpha / beta of the scaled zone (SUM_QUANTILE with
    data vectors, scaled_zone.cc:34-76) - until now fp64 fell back to the operator path for both.
    Fused against unfused against the oracle, to fp64 rounding."""
    m, n = shape
    if kind == "lasso":
        if solver_id == 0:
            pytest.skip("covered by test_fused_sweep_fp64")
        prob, info = problems.lasso(m, n, seed=8)
    else:
        A, b = problems.regression_data(m, n, seed=8)
        lam = 0.3 * np.abs(A.T.dot(b)).max()
        x = ir.variable(n, 1, problems.LASSO_COPY)
        y = ir.variable(n, 1, problems.LASSO_VAR)
        f0 = ir.prox(ProxFunction.SUM_SQUARE, ir.add(ir.linear_map(ir.dense_matrix(A), x),
                                                     ir.linear_map(ir.scalar(-1, m), ir.constant(b))))
        rq = np.random.RandomState(9)
        qa, qb = ir.constant(0.2 + rq.rand(n)), ir.constant(0.2 + rq.rand(n))
        qd = dict(qa.data)
        qd.update(qb.data)
        f1 = ir.prox(ProxFunction.SUM_QUANTILE, y, alpha=lam, data=qd,
                     scaled_zone_params=wire.ProxScaledZoneParams(alpha_expr=qa.proto, beta_expr=qb.proto))
        prob = ir.Problem([f0, f1], [ir.zero(ir.add(x, ir.linear_map(ir.scalar(-1, n), y)))])
    pb, data = prob.SerializeToString(), prob.expression_data()
    sb = wire.SolverParams(max_iterations=200, solver=solver_id).SerializeToString()
    solve_mod.set_option("dtype", "f64")
    try:
        solve_mod.set_option("fused", "1")
        solve_mod.profile_reset()
        solve_mod.profile_enable(True)
        st_f, x_f = solve_mod.solve(pb, [], sb, data)
        tags = solve_mod.profile_dump()
        solve_mod.profile_enable(False)
        solve_mod.set_option("fused", "0")
        st_g, x_g = solve_mod.solve(pb, [], sb, data)
    finally:
        solve_mod.set_option("fused", "1")
        solve_mod.set_option("dtype", "f32")
    assert any(t.startswith("lasso_fused:%dx%d" % (m, n)) for t in tags), sorted(tags)
    st_o, x_o = orc.solve(pb, [], sb, data)
    sf, sg, so = (wire.SolverStatus.FromString(s) for s in (st_f, st_g, st_o))
    assert sf.state == sg.state == so.state
    assert sf.num_iterations == sg.num_iterations == so.num_iterations
    for f in ("r_norm", "s_norm", "epsilon_primal", "epsilon_dual"):
        np.testing.assert_allclose(getattr(sf.residuals, f), getattr(so.residuals, f), rtol=1e-8, atol=1e-11)
    for k in x_o:
        np.testing.assert_allclose(np.frombuffer(x_f[k]), np.frombuffer(x_g[k]), rtol=1e-9, atol=1e-11, err_msg=k)
        np.testing.assert_allclose(np.frombuffer(x_f[k]), np.frombuffer(x_o[k]), rtol=1e-8, atol=1e-10, err_msg=k)


@pytest.mark.parametrize("apply_mode", ["slab", "replicated"])
def test_fused_sweep_sharded(solve_mod, tmp_path, apply_mode):
    """Fused pass on column slabs with the all-reduce between the pass and the cached-inverse
    apply (3 ranks sharing this GPU, fp32); the inverse applied by row slabs + all-gather
    (default) or whole on every rank."""
    from tests import mp_util
    m, n = 40, 101
    x0, x1, status, parts = mp_util.run_ranks(3, "hip", str(tmp_path), m, n, seed=3,
                                              env_extra={"EPS_TEST_DTYPE": "f32",
                                                         "EPSILON_HIP_SHARDED_APPLY": apply_mode})
    prob, info = problems.lasso(m, n, seed=3)
    st, x = orc.solve(prob.SerializeToString(), [], wire.SolverParams().SerializeToString(),
                      prob.expression_data())
    S = wire.SolverStatus.FromString(st)
    for s, p in zip(status, parts):
        assert int(p["state"]) == wire.SolverStatus.OPTIMAL and int(s[0]) == S.num_iterations
    np.testing.assert_allclose(x0, np.frombuffer(x[problems.LASSO_COPY]), rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(x1, np.frombuffer(x[problems.LASSO_VAR]), rtol=1e-3, atol=1e-4)


@pytest.mark.parametrize("world,apply_mode,graph,dt", [(2, "slab", "1", "f32"), (3, "slab", "0", "f32"),
                                                       (3, "slab", "1", "f32"), (3, "replicated", "1", "f32"),
                                                       (3, "slab", "1", "f64"), (2, "replicated", "0", "f64")])
def test_fused_sweep_sharded_peer_window(solve_mod, tmp_path, world, apply_mode, graph, dt):
    """The same solve with the per-sweep exchanges on the one-shot peer-write window
    (csrc/kernels_peer.hip): ranks are separate processes sharing this GPU, their windows are
    mapped into each other through HIP IPC, every rank writes its partial forward product / its
    slab of w straight into the peers' windows from inside the sweep's kernels; sweeps between
    residual checks eager (graph 0) or replayed from a hipGraph (graph 1).  fp64: a value travels
    as two granules (high word, low word), each its own flag; iterates to 1e-8 of the oracle's."""
    from tests import mp_util
    m, n = 40, 101
    x0, x1, status, parts = mp_util.run_ranks(world, "hip", str(tmp_path), m, n, seed=3,
                                              env_extra={"EPS_TEST_DTYPE": dt, "EPS_TEST_PEER": "1",
                                                         "EPSILON_HIP_GRAPH": graph,
                                                         "EPSILON_HIP_SHARDED_APPLY": apply_mode})
    prob, info = problems.lasso(m, n, seed=3)
    st, x = orc.solve(prob.SerializeToString(), [], wire.SolverParams().SerializeToString(),
                      prob.expression_data())
    S = wire.SolverStatus.FromString(st)
    for s, p in zip(status, parts):
        assert int(p["state"]) == wire.SolverStatus.OPTIMAL and int(s[0]) == S.num_iterations
        tags = set(str(t) for t in p["tags"])
        # the exchanges really rode on the window: no collective call inside the sweeps
        assert "peer_reduce_exchange" in tags, sorted(tags)
        assert ("peer_slab_apply_exchange" in tags) == (apply_mode == "slab"), sorted(tags)
    rtol, atol = (1e-3, 1e-4) if dt == "f32" else (1e-8, 1e-10)
    np.testing.assert_allclose(x0, np.frombuffer(x[problems.LASSO_COPY]), rtol=rtol, atol=atol)
    np.testing.assert_allclose(x1, np.frombuffer(x[problems.LASSO_VAR]), rtol=rtol, atol=atol)


def test_peer_window_graph_replay_is_bit_identical(solve_mod, tmp_path):
    """45 sweeps through the exchange kernels, once launched eagerly and once replayed from
    hipGraphs of the sweeps between residual checks: the iterates must agree bit for bit; plus
    the oracle comparison and the rank-of-8 rehearsal window (tests/peer_single_rank.py)."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    fe, fg = str(tmp_path / "eager.npz"), str(tmp_path / "graph.npz")
    env = dict(os.environ, MASTER_PORT="29641")
    r1 = subprocess.run([sys.executable, os.path.join(here, "peer_single_rank.py"), "eager", fe],
                        capture_output=True, text=True, timeout=600, env=env)
    assert r1.returncode == 0 and "PEER_EAGER_DONE" in r1.stdout, r1.stdout[-2000:] + r1.stderr[-3000:]
    env = dict(os.environ, MASTER_PORT="29642")
    r2 = subprocess.run([sys.executable, os.path.join(here, "peer_single_rank.py"), "graph", fg],
                        capture_output=True, text=True, timeout=600, env=env)
    assert r2.returncode == 0 and "PEER_OK" in r2.stdout, r2.stdout[-2000:] + r2.stderr[-3000:]
    a, b = np.load(fe), np.load(fg)
    assert sorted(a.files) == sorted(b.files) and a.files
    for k in a.files:
        assert np.array_equal(a[k], b[k]), k


@pytest.mark.parametrize("shape", [(3, 3), (12, 7), (7, 12), (40, 25), (1, 5)])
def test_nuclear_norm_prox(solve_mod, dtype, shape):
    """NORM_NUCLEAR (reference prox_test.py:190; prox/ortho_invariant.cc): one-sided Jacobi SVD
    on the device vs the oracle's eig(Y^T Y) restatement and vs numpy's SVD."""
    m, n = shape
    rng = np.random.RandomState(m * 100 + n)
    V = rng.randn(m, n)
    lam = 0.4
    X = ir.variable(m, n, "var:X")
    expr = ir.prox(ProxFunction.NORM_NUCLEAR, X)
    tol = dict(rtol=1e-7, atol=1e-8) if dtype == "f64" else dict(rtol=2e-4, atol=2e-5)
    got = run_prox(solve_mod, expr, lam, {"var:X": V.reshape(-1, order="F")}, tol)
    U, s, Vt = np.linalg.svd(V, full_matrices=False)
    want = (U * np.maximum(s - lam, 0)) @ Vt
    np.testing.assert_allclose(np.frombuffer(got["var:X"]).reshape((m, n), order="F"), want, **tol)


@pytest.mark.parametrize("n", [10, 24])
def test_robust_pca_problem(solve_mod, dtype, n):
    """BASELINE.json configs[4] shape (robust PCA: NORM_NUCLEAR + NORM_1, constraint with a
    constant) at test size, iterate parity with the oracle."""
    prob, info = problems.robust_pca(n, r=2, seed=0)
    sg, xg, so, xo = solve_both(solve_mod, prob, wire.SolverParams(max_iterations=60))
    assert sg.num_iterations == so.num_iterations and sg.state == so.state
    tol = dict(rtol=1e-6, atol=1e-7) if dtype == "f64" else dict(rtol=5e-3, atol=5e-3)
    for k in xo:
        np.testing.assert_allclose(np.frombuffer(xg[k]), np.frombuffer(xo[k]), err_msg=k, **tol)


@pytest.mark.parametrize("case", ["noise", "walk", "ties", "steps", "big", "tiny"])
def test_tv1d_parallel_kernel(solve_mod, dtype, case):
    """Exact parallel TV-1D prox (level-set divide and conquer, kernels_tv.hip) vs the DP
    oracle, plus the KKT certificate of SURVEY.md 8(c) on the device result."""
    from oracle import c_oracle
    rng = np.random.RandomState(11)
    if case == "noise":
        v, lam = rng.randn(5000), 0.7
    elif case == "walk":
        v, lam = np.cumsum(rng.randn(20000)) + rng.randn(20000), 25.0
    elif case == "ties":
        v, lam = np.round(rng.randn(3000) * 2), 1.0
    elif case == "steps":
        v, lam = np.repeat(rng.randn(300), 50) + 0.1 * rng.randn(15000), 3.0
    elif case == "big":
        b, lam = problems.tv_1d_data(300000, seed=1)
        v = b
    else:
        v, lam = np.array([0.0, 10.0]), 1.0
    if dtype == "f32":
        v = v.astype(np.float32).astype(np.float64)
    got = solve_mod.tv1d(v, lam)
    want = c_oracle.tv1d(v, lam)
    tol = dict(rtol=1e-9, atol=1e-9) if dtype == "f64" else dict(rtol=1e-4, atol=1e-4 * max(1.0, np.abs(v).max()))
    np.testing.assert_allclose(got, want, **tol)
    if dtype == "f64":
        bound, jump, end = orc.tv1d_kkt_violation(got, v, lam)
        assert bound < 1e-7 and jump < 1e-7 and end < 1e-7
    if case == "tiny":
        np.testing.assert_allclose(got, [1.0, 9.0])


def lasso_with_parameter(m, n, seed):
    """Compiled lasso whose right-hand side b is a CVXPY Parameter (a CONSTANT carrying a
    parameter_id, reference python/epopt/expression.py:227-236)."""
    A, b = problems.regression_data(m, n, seed=seed)
    lam = 0.3 * np.abs(A.T.dot(b)).max()
    prob = problems.lasso_ir(ir.dense_matrix(A), ir.parameter(m, 1, "param:b"), lam, n)
    return prob, A, b, lam


def bind(b):
    data = {}
    c = ir.store(np.asarray(b, dtype=np.float64).reshape(-1, 1), data)
    return ("param:b", c.SerializeToString()), data


def test_parameter_binding_one_shot(solve_mod, dtype):
    """`parameters` argument of _solve.solve (reference solvemodule.cc:89-106)."""
    prob, A, b, lam = lasso_with_parameter(60, 150, 4)
    p, pdata = bind(b)
    data = dict(prob.expression_data())
    data.update(pdata)
    sb = wire.SolverParams().SerializeToString()
    st_g, x_g = solve_mod.solve(prob.SerializeToString(), [p], sb, data)
    st_o, x_o = orc.solve(prob.SerializeToString(), [p], sb, data)
    a, o = wire.SolverStatus.FromString(st_g), wire.SolverStatus.FromString(st_o)
    assert a.state == o.state and a.num_iterations == o.num_iterations
    tol = dict(rtol=1e-8, atol=1e-10) if dtype == "f64" else dict(rtol=1e-3, atol=1e-4)
    for k in x_o:
        np.testing.assert_allclose(np.frombuffer(x_g[k]), np.frombuffer(x_o[k]), **tol)


def test_warm_start_reuses_factorisation(solve_mod, dtype):
    """a live solver handle keeps A and the cached inverse in HBM; re-binding
    the parameter b and re-running Init() with warm_start redoes no GEMM / inverse and starts
    from the previous x/y/u (reference solvemodule.cc:142-156, prox_admm.cc:115-120)."""
    prob, A, b, lam = lasso_with_parameter(96, 240, 5)
    rng = np.random.RandomState(0)
    b2 = b + 0.05 * rng.randn(b.size)
    pb = prob.SerializeToString()
    sp = wire.SolverParams(warm_start=True)
    sb = sp.SerializeToString()
    p1, d1 = bind(b)
    p2, d2 = bind(b2)
    data = dict(prob.expression_data())
    data.update(d1)
    data.update(d2)
    s = solve_mod.Solver(pb, sb, data)
    s.set_parameter(*p1)
    s.init()
    s.run(-1)
    st1, x1 = s.result()
    solve_mod.profile_reset()
    solve_mod.profile_enable(True)
    s.set_parameter(*p2)
    s.init()
    tags = solve_mod.profile_dump()
    solve_mod.profile_enable(False)
    assert not any(t.startswith(("syrk", "gemm", "spd_inverse")) for t in tags), \
        "second Init redid dense setup work: %s" % sorted(tags)
    s.run(-1)
    st2, x2 = s.result()
    s.close()
    # oracle: same two solves on one solver object with warm_start
    problem = wire.Problem.FromString(pb)
    odata = dict(data)
    odata[orc.PARAMS_KEY] = {"param:b": wire.Constant.FromString(p1[1])}
    osolver = orc.create_solver(problem, odata, sp)
    osolver.solve()
    it1 = osolver.status.num_iterations
    odata[orc.PARAMS_KEY] = {"param:b": wire.Constant.FromString(p2[1])}
    xo = osolver.solve()
    it2 = osolver.status.num_iterations
    a1, a2 = wire.SolverStatus.FromString(st1), wire.SolverStatus.FromString(st2)
    assert a1.num_iterations == it1 and a2.num_iterations == it2
    assert it2 <= it1  # the warm start pays off
    tol = dict(rtol=1e-7, atol=1e-9) if dtype == "f64" else dict(rtol=2e-3, atol=2e-4)
    for k in x2:
        np.testing.assert_allclose(np.frombuffer(x2[k]), xo(k), err_msg=k, **tol)


@pytest.mark.parametrize("kind", ["norm_1", "deadzone", "hinge", "quantile", "sum_square", "inside"])
def test_epigraph_projections(solve_mod, dtype, kind):
    """Epigraph operators (reference prox_test.py:232-246: NORM_1 / SUM_DEADZONE / SUM_HINGE /
    SUM_QUANTILE / SUM_SQUARE epigraphs): projection of (v, s) onto {f(x) <= t}."""
    n = 50
    for trial in range(3):
        rng = np.random.RandomState(trial + 17)
        x = ir.variable(n, 1, "var:x")
        t = ir.variable(1, 1, "var:t")
        v, s = rng.randn(n), 0.5 * rng.randn()
        kw = {}
        if kind in ("norm_1", "inside"):
            typ = ProxFunction.NORM_1
            if kind == "inside":
                s = float(np.abs(v).sum()) + 1.0
        elif kind == "deadzone":
            typ, kw = ProxFunction.SUM_DEADZONE, dict(scaled_zone_params=wire.ProxScaledZoneParams(m=0.3))
        elif kind == "hinge":
            typ = ProxFunction.SUM_HINGE
        elif kind == "quantile":
            qa, qb = ir.constant(np.full(n, 0.7)), ir.constant(np.full(n, 1.6))
            data = dict(qa.data)
            data.update(qb.data)
            typ, kw = ProxFunction.SUM_QUANTILE, dict(
                data=data, scaled_zone_params=wire.ProxScaledZoneParams(alpha_expr=qa.proto, beta_expr=qb.proto))
        else:
            typ = ProxFunction.SUM_SQUARE
        expr = ir.prox(typ, x, t, epigraph=True, **kw)
        tol = dict(rtol=1e-9, atol=1e-10) if dtype == "f64" else dict(rtol=3e-4, atol=3e-5)
        run_prox(solve_mod, expr, 1.0, {"var:x": v, "var:t": np.array([s])}, tol)


@pytest.mark.parametrize("kind", ["norm_1", "hinge"])
def test_epigraph_projection_long_vector(solve_mod, dtype, kind):
    """Above 65536 entries the scaled-zone epigraph runs device-wide reductions with the active-set
    iteration on the host (shorter vectors are solved on chip by one launch)."""
    n = 200000
    rng = np.random.RandomState(5)
    x, t = ir.variable(n, 1, "var:x"), ir.variable(1, 1, "var:t")
    typ = ProxFunction.NORM_1 if kind == "norm_1" else ProxFunction.SUM_HINGE
    v, s = rng.randn(n), 0.05 * n
    expr = ir.prox(typ, x, t, epigraph=True)
    tol = dict(rtol=1e-9, atol=1e-10) if dtype == "f64" else dict(rtol=3e-4, atol=3e-4)
    run_prox(solve_mod, expr, 1.0, {"var:x": v, "var:t": np.array([s])}, tol)


def test_sample_sharded_multiclass_hinge(solve_mod, tmp_path):
    """BASELINE.json configs[3] shape: multiclass hinge with the SAMPLES sharded over 2 ranks
    (t, y and the big constraint row are sharded; Theta is replicated).  The contraction
    kron(I_k, X_g^T X_g) is summed over ranks as a Kronecker factor, the X^T(.) products of the
    forward substitution as vectors; iterates must equal the single-process oracle."""
    from tests import mp_util
    m, nf = 40, 8
    t_all, theta_parts, status, parts = mp_util.run_ranks(2, "hip_mnist", str(tmp_path), m, nf, seed=0,
                                                          max_iter=40, env_extra={"EPS_TEST_DTYPE": "f64"})
    X, Y = problems.multiclass_hinge_data(m, nf, 3, seed=0)
    prob, info = problems.multiclass_hinge(X, Y, lam=0.1)
    st, x = orc.solve(prob.SerializeToString(), [], wire.SolverParams(max_iterations=40).SerializeToString(),
                      prob.expression_data())
    S = wire.SolverStatus.FromString(st)
    for s_, p in zip(status, parts):
        assert int(s_[0]) == S.num_iterations and int(p["state"]) == S.state
        np.testing.assert_allclose(s_[1:], [S.residuals.r_norm, S.residuals.s_norm,
                                            S.residuals.epsilon_primal, S.residuals.epsilon_dual], rtol=1e-7)
        np.testing.assert_allclose(p["x1"], np.frombuffer(x["var:Theta"]), rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(t_all, np.frombuffer(x["max_entries:t"]), rtol=1e-7, atol=1e-9)


# ---- edge cases: degenerate shapes, limits, malformed inputs -----------------------------------


@pytest.mark.parametrize("shape", [(1, 1), (1, 7), (7, 1), (3, 3)])
def test_lasso_degenerate_shapes(solve_mod, dtype, shape):
    """One row, one column, square: the elimination order, the kernels' tails and the stopping
    rule on the smallest problems."""
    prob, info = problems.lasso(shape[0], shape[1], seed=2)
    sg, xg, so, xo = solve_both(solve_mod, prob, wire.SolverParams(max_iterations=200))
    assert sg.state == so.state and sg.num_iterations == so.num_iterations
    tol = dict(rtol=1e-7, atol=1e-9) if dtype == "f64" else dict(rtol=2e-3, atol=2e-4)
    for k in xo:
        np.testing.assert_allclose(np.frombuffer(xg[k]), np.frombuffer(xo[k]), err_msg=k, **tol)


def test_iteration_limits(solve_mod, dtype):
    """max_iterations = 1 stops after one sweep with MAX_ITERATIONS_REACHED; zero tolerances never
    report OPTIMAL (reference prox_admm.cc:149-168)."""
    prob, _ = problems.lasso(20, 50, seed=1)
    sg, xg, so, xo = solve_both(solve_mod, prob, wire.SolverParams(max_iterations=1))
    assert sg.state == so.state == wire.SolverStatus.MAX_ITERATIONS_REACHED
    assert sg.num_iterations == so.num_iterations
    sg, xg, so, xo = solve_both(solve_mod, prob, wire.SolverParams(max_iterations=35, abs_tol=0, rel_tol=0))
    assert sg.state == so.state == wire.SolverStatus.MAX_ITERATIONS_REACHED
    assert sg.num_iterations == so.num_iterations


def test_malformed_inputs_raise(solve_mod):
    """Every failed check is an exception with a message, never a crash: missing data blob, blob
    of the wrong size, argument / variable size mismatch, non-ZERO cone, rho != 1."""
    prob, info = problems.lasso(8, 12, seed=0)
    pb, data = prob.SerializeToString(), prob.expression_data()
    sb = wire.SolverParams().SerializeToString()
    some_key = sorted(data)[0]
    missing = {k: v for k, v in data.items() if k != some_key}
    with pytest.raises(solve_mod.error, match="not in data map"):
        solve_mod.solve(pb, [], sb, missing)
    short = dict(data)
    short[some_key] = data[some_key][:-8]
    with pytest.raises(solve_mod.error):
        solve_mod.solve(pb, [], sb, short)
    with pytest.raises(solve_mod.error):
        solve_mod.solve(pb, [], wire.SolverParams(rho=2.0).SerializeToString(), data)
    # a linear map whose inner dimension does not match its argument
    x = ir.variable(5, 1, "var:x")
    bad = ir.prox(ProxFunction.SUM_SQUARE, ir.linear_map(ir.dense_matrix(np.ones((3, 5))), x))
    bad.proto.arg[0].arg[0].size.dim[0] = 4  # the variable now has 4 entries, the map 5 columns
    with pytest.raises(solve_mod.error):
        solve_mod.eval_prox(bad.proto.SerializeToString(), 1.0, bad.data, {"var:x": np.zeros(4).tobytes()})
    # v of the wrong length
    ok = ir.prox(ProxFunction.NORM_1, ir.variable(5, 1, "var:x"))  # (a fresh, unmutated variable)
    with pytest.raises(solve_mod.error):
        solve_mod.eval_prox(ok.proto.SerializeToString(), 1.0, {}, {"var:x": np.zeros(4).tobytes()})
    # the library is still usable afterwards
    st, _ = solve_mod.solve(pb, [], sb, data)
    assert wire.SolverStatus.FromString(st).state == wire.SolverStatus.OPTIMAL


def test_large_host_blob_upload(solve_mod, dtype):
    """Host blobs of 64 MB and more take the pinned, multi-threaded upload (fp32 mode converts on
    the host; two 32 MB buffers, so this one needs three chunks); smaller ones the plain copy.
    y = A x for a 3000 x 3000 matrix (72 MB of float64) whose entries make a dropped, repeated or
    misplaced stripe visible, against numpy."""
    rng = np.random.RandomState(12)
    m, n = 3000, 3000
    A = rng.randn(m, n) + np.arange(m)[:, None] * 1e-3 + np.arange(n)[None, :] * 1e-2
    x = rng.randn(n)
    got = solve_mod.linear_map_apply(ir.dense_matrix(A), x)
    want = A.dot(x)
    tol = dict(rtol=1e-11, atol=1e-9) if dtype == "f64" else dict(rtol=2e-4, atol=2e-2)
    np.testing.assert_allclose(got, want, **tol)
