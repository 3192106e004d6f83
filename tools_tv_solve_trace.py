#!/usr/bin/env python3
"""eps_solve of the reference's tv_1d problem at n = 10^8 for a kernel trace
(rocprofv3 --kernel-trace --stats -- python3 tools_tv_solve_trace.py)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from epsilon_amd import _solve, problems, wire  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10 ** 8
_solve.set_option("dtype", "f32")
prob, info = problems.tv_1d(n, seed=0)
pb, data = prob.SerializeToString(), prob.expression_data()
t0 = time.time()
st, x = _solve.solve(pb, [], wire.SolverParams(max_iterations=50).SerializeToString(), data)
S = wire.SolverStatus.FromString(st)
print("solve %.3f s, loop %.3f s, %d sweeps" % (time.time() - t0, S.timing.total_time - S.timing.init_time,
                                                 S.num_iterations + 1), flush=True)
