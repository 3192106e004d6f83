"""C-ABI ownership contract of the solver handles (include/epsilon_hip.h): host blobs are copied by
eps_solver_create / eps_solver_set_parameter, exactly as the reference copies every blob into its
DataMap (python/epopt/solvemodule.cc:58-72), so a C caller may overwrite and free its buffers as
soon as the call returns; re-binding a location replaces its contents.

The calls below go through ctypes directly (not through epsilon_amd._solve.Solver) so that the
test owns the buffers and can destroy them at the points a C caller could.
"""

import ctypes

import numpy as np
import pytest

from epsilon_amd import _solve, ir, problems, wire
from oracle import epsilon_oracle as orc

pytestmark = pytest.mark.gpu


class Handle(object):
    """eps_solver_* through raw ctypes with caller-owned, destroyable host buffers."""

    def __init__(self, pb, sb, data):
        self.L = _solve.lib()
        self.h = ctypes.c_void_p()
        arr, bufs = self._blobs(data)
        _solve._check(self.L.eps_solver_create(pb, ctypes.c_size_t(len(pb)), sb, ctypes.c_size_t(len(sb)),
                                               arr, ctypes.c_size_t(len(data)), ctypes.byref(self.h)))
        self._trash(bufs)

    @staticmethod
    def _blobs(data):
        items = list(data.items())
        arr = (_solve._Blob * max(len(items), 1))()
        bufs = []
        for i, (k, v) in enumerate(items):
            kb = ctypes.create_string_buffer(k.encode("utf-8"))
            vb = ctypes.create_string_buffer(bytes(v), len(v))
            bufs.extend([kb, vb])
            arr[i].key = ctypes.cast(kb, ctypes.c_char_p)
            arr[i].ptr = ctypes.cast(vb, ctypes.c_void_p)
            arr[i].len = len(v)
            arr[i].kind = 0
        return arr, bufs

    @staticmethod
    def _trash(bufs):
        # what a C caller's free() + reuse does: the bytes the library was shown are gone
        for b in bufs:
            ctypes.memset(b, 0xA5, len(b))
        del bufs[:]

    def set_parameter(self, pid, cbytes, data):
        arr, bufs = self._blobs(data)
        pidb = ctypes.create_string_buffer(pid.encode())
        cb = ctypes.create_string_buffer(cbytes, len(cbytes))
        _solve._check(self.L.eps_solver_set_parameter(self.h, pidb, cb, ctypes.c_size_t(len(cbytes)), arr,
                                                      ctypes.c_size_t(len(data))))
        self._trash(bufs + [pidb, cb])

    def init(self):
        _solve._check(self.L.eps_solver_init(self.h))

    def run(self):
        done = ctypes.c_int()
        _solve._check(self.L.eps_solver_run(self.h, ctypes.c_int(-1), ctypes.byref(done)))

    def result(self):
        res = ctypes.c_void_p()
        _solve._check(self.L.eps_solver_result(self.h, ctypes.byref(res)))
        return _solve._take_result(res)

    def close(self):
        self.L.eps_solver_destroy(self.h)


def _lasso_param(m, n, seed):
    A, b = problems.regression_data(m, n, seed=seed)
    lam = 0.3 * np.abs(A.T.dot(b)).max()
    prob = problems.lasso_ir(ir.dense_matrix(A), ir.parameter(m, 1, "param:b"), lam, n)
    return prob, A, b


def _bind(b, key):
    """Constant proto for b stored under a FIXED location, so that a second binding re-uses it."""
    data = {}
    c = ir.store(np.asarray(b, dtype=np.float64).reshape(-1, 1), data)
    (old,) = list(data.keys())
    c.data_location = key
    return c.SerializeToString(), {key: data[old]}


@pytest.mark.parametrize("dt", ["f64", "f32"])
def test_host_blobs_may_be_freed_after_create_and_rebinding_replaces(dt):
    _solve.set_option("dtype", dt)
    try:
        prob, A, b = _lasso_param(80, 200, 11)
        rng = np.random.RandomState(3)
        b2 = b + 0.3 * rng.randn(b.size)
        pb = prob.SerializeToString()
        sp = wire.SolverParams(warm_start=True)
        sb = sp.SerializeToString()
        key = "/mem/data/rebound"
        c1, d1 = _bind(b, key)
        c2, d2 = _bind(b2, key)  # SAME location, new contents

        h = Handle(pb, sb, dict(prob.expression_data()))  # A's bytes are trashed after create
        h.set_parameter("param:b", c1, d1)                # b's bytes are trashed after the call
        h.init()                                          # reads the library's own copies
        h.run()
        st1, x1 = h.result()
        h.set_parameter("param:b", c2, d2)                # re-bind the same location
        h.init()                                          # warm re-Init: must see b2, not b
        h.run()
        st2, x2 = h.result()
        h.close()

        tol = dict(rtol=1e-7, atol=1e-9) if dt == "f64" else dict(rtol=2e-3, atol=2e-4)
        # one-shot solves through the binding (buffers alive throughout) are the comparison
        data1 = dict(prob.expression_data())
        data1.update(d1)
        st_a, x_a = _solve.solve(pb, [("param:b", c1)], wire.SolverParams().SerializeToString(), data1)
        a, g = wire.SolverStatus.FromString(st_a), wire.SolverStatus.FromString(st1)
        assert a.state == g.state and a.num_iterations == g.num_iterations
        for k in x_a:
            np.testing.assert_allclose(np.frombuffer(x1[k]), np.frombuffer(x_a[k]), err_msg=k, **tol)
        # second solve: the oracle's warm-started solve on b2
        problem = wire.Problem.FromString(pb)
        odata = dict(prob.expression_data())
        odata.update(d1)
        odata[orc.PARAMS_KEY] = {"param:b": wire.Constant.FromString(c1)}
        osolver = orc.create_solver(problem, odata, sp)
        osolver.solve()
        odata.update(d2)
        odata[orc.PARAMS_KEY] = {"param:b": wire.Constant.FromString(c2)}
        xo = osolver.solve()
        g2 = wire.SolverStatus.FromString(st2)
        assert g2.num_iterations == osolver.status.num_iterations
        for k in x2:
            np.testing.assert_allclose(np.frombuffer(x2[k]), xo(k), err_msg=k, **tol)
        # and the two right-hand sides really give different answers (the test can fail)
        assert max(np.abs(np.frombuffer(x2[k]) - np.frombuffer(x1[k])).max() for k in x2) > 1e-3
    finally:
        _solve.set_option("dtype", "f32")


def test_rebinding_a_data_matrix_drops_the_cached_factorisation():
    """A new matrix under an already-uploaded location: the device copy and every cached operator
    built from the old one must go (DataMap::Insert bumps the location's generation)."""
    _solve.set_option("dtype", "f64")
    try:
        m, n = 50, 120
        A1, b = problems.regression_data(m, n, seed=5)
        A2, _ = problems.regression_data(m, n, seed=6)
        lam = 0.3 * np.abs(A1.T.dot(b)).max()
        key = "/mem/data/A"

        def problem_for(A):
            data = {}
            c = ir.store(A, data)
            (old,) = list(data.keys())
            c.data_location = key
            prob = problems.lasso_ir(ir.dense_matrix(constant=c, data={key: data[old]}), ir.constant(b), lam, n)
            return prob

        p1, p2 = problem_for(A1), problem_for(A2)
        pb = p1.SerializeToString()
        assert pb == p2.SerializeToString()  # same IR, only the bytes under `key` differ
        sb = wire.SolverParams().SerializeToString()
        h = Handle(pb, sb, dict(p1.expression_data()))
        h.init()
        h.run()
        _, xa = h.result()
        # re-bind A through set_parameter's data argument (the parameter itself is a dummy)
        dummy = wire.Constant(constant_type=wire.Constant.SCALAR, scalar=0.0).SerializeToString()
        h.set_parameter("param:unused", dummy, {key: p2.expression_data()[key]})
        h.init()
        h.run()
        _, xb = h.result()
        h.close()
        _, x2 = _solve.solve(pb, [], sb, dict(p2.expression_data()))
        for k in x2:
            np.testing.assert_allclose(np.frombuffer(xb[k]), np.frombuffer(x2[k]), rtol=1e-8, atol=1e-10)
        assert max(np.abs(np.frombuffer(xa[k]) - np.frombuffer(xb[k])).max() for k in xa) > 1e-3
    finally:
        _solve.set_option("dtype", "f32")
