"""Caller-level shim: what `epopt.solve` does AFTER the compiler has produced the prox-affine
problem (reference python/epopt/cvxpy_solver.py:27-104), on top of `epsilon_amd._solve`.

The CVXPY conversion and the compiler stay the reference's (they need Python 2 and cvxpy 0.3.6,
SURVEY.md 8(c)); everything from the compiled `Problem` on is restated here without CVXPY:

  * `cvxpy_status`       SolverStatus.state -> CVXPY status string        (cvxpy_solver.py:34-39)
  * `unpack_values`      float64 bytes -> (m, n) arrays via reshape(n, m).T, i.e. column-major,
                         exactly `set_solution`                            (cvxpy_solver.py:27-32)
  * `parameter_values`   [(parameter_id, Constant bytes)] + data blobs     (cvxpy_solver.py:41-44)
  * `solve`              the single-prox shortcut `eval_prox(f, lam=1e12, data, {})` for a problem
                         with one objective term and no constraints        (cvxpy_solver.py:79-88),
                         else `_solve.solve` / a warm-started solver handle (cvxpy_solver.py:70-74,
                         solvemodule.cc:142-156)

A maintainer of the reference swaps `from epopt import _solve` for `from epsilon_amd import _solve`
in cvxpy_solver.py and keeps its own `solve`; this module is the same logic for callers that hold a
compiled problem (tests, benchmarks, services that cache compiled forms).
"""

import time

import numpy as np

from . import _solve, ir, wire

# cvxpy.settings values (the reference imports them, cvxpy_solver.py:7)
OPTIMAL = "optimal"
OPTIMAL_INACCURATE = "optimal_inaccurate"
SOLVER_ERROR = "solver_error"

SINGLE_PROX_LAMBDA = 1e12  # cvxpy_solver.py:82


class SolverError(Exception):
    """reference cvxpy_solver.py:24-25"""


def cvxpy_status(solver_status):
    """reference cvxpy_solver.py:34-39"""
    if solver_status.state == wire.SolverStatus.OPTIMAL:
        return OPTIMAL
    if solver_status.state == wire.SolverStatus.MAX_ITERATIONS_REACHED:
        return OPTIMAL_INACCURATE
    return SOLVER_ERROR


def unpack_values(sizes, values):
    """{var_id: float64 bytes} -> {var_id: ndarray of shape (m, n)}; `sizes` = {var_id: (m, n)}.
    The bytes are column-major, so reshape to (n, m) and transpose (cvxpy_solver.py:27-32)."""
    out = {}
    for var_id, (m, n) in sizes.items():
        if var_id not in values:
            raise SolverError("no value returned for variable %s" % var_id)
        x = np.frombuffer(values[var_id], dtype=np.float64)
        if x.size != m * n:
            raise SolverError("variable %s: %d values for size %dx%d" % (var_id, x.size, m, n))
        out[var_id] = x.reshape(n, m).transpose().copy()
    return out


def parameter_values(params):
    """{parameter_id: ndarray} -> ([(parameter_id, Constant bytes)], {location: bytes}); the value
    travels as a data blob exactly as `constant.store` ships it (cvxpy_solver.py:41-44)."""
    data, out = {}, []
    for pid, value in params.items():
        value = np.asarray(value, dtype=np.float64)
        if value.ndim == 1:
            value = value.reshape(-1, 1)
        out.append((pid, ir.store(value, data).SerializeToString()))
    return out, data


_handles = {}  # warm start: serialized problem -> live solver handle (solvemodule.cc:142-151)


def clear_warm_start_cache():
    for h in _handles.values():
        h.close()
    _handles.clear()


def solve(problem, params=None, **kwargs):
    """problem: `ir.Problem` (compiled prox-affine form).  kwargs: SolverParams fields, as the
    reference passes `**kwargs` into the proto (cvxpy_solver.py:69).
    Returns (status string, {var_id: (m, n) array}, info dict)."""
    solver_params = wire.SolverParams(**kwargs)
    proto = problem.proto()
    sizes = ir.get_variables(proto)
    if not sizes:  # "nothing to do in this case" (cvxpy_solver.py:65-67)
        return OPTIMAL, {}, {"solve_time": 0.0}
    data = dict(problem.expression_data())
    t0 = time.time()
    info = {}
    if len(proto.objective.arg) == 1 and not proto.constraint:
        # one prox function, nothing to split: a single prox evaluation with a huge lambda
        values = _solve.eval_prox(proto.objective.arg[0].SerializeToString(), SINGLE_PROX_LAMBDA, data, {})
        status = OPTIMAL
        info["route"] = "eval_prox"
    else:
        plist, pdata = parameter_values(params or {})
        data.update(pdata)
        pbytes = proto.SerializeToString()
        sbytes = solver_params.SerializeToString()
        if solver_params.warm_start:
            h = _handles.get(pbytes)
            if h is None:
                h = _solve.Solver(pbytes, sbytes, data)
                _handles[pbytes] = h
            for pid, cbytes in plist:
                h.set_parameter(pid, cbytes, pdata)
            h.init()
            h.run(-1)
            status_bytes, values = h.result()
            info["route"] = "warm_start_handle"
        else:
            status_bytes, values = _solve.solve(pbytes, plist, sbytes, data)
            info["route"] = "solve"
        st = wire.SolverStatus.FromString(status_bytes)
        status = cvxpy_status(st)
        info["num_iterations"] = st.num_iterations
        info["solver_status"] = st
    info["solve_time"] = time.time() - t0
    return status, unpack_values(sizes, values), info
