import sys, os, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import bench_suite
from epsilon_amd import _solve, wire, problems
_solve.set_option("dtype", "f32")
wp, _ = problems.lasso(256, 1024, seed=1)
_solve.solve(wp.SerializeToString(), [], wire.SolverParams(max_iterations=20).SerializeToString(), wp.expression_data())
prob, obj, ref = bench_suite.robust_pca()
pb, data = prob.SerializeToString(), prob.expression_data()
params = wire.SolverParams(max_iterations=50000)
_solve.profile_enable(True); _solve.profile_reset()
t0 = time.time(); st, x = _solve.solve(pb, [], params.SerializeToString(), data); dt = time.time() - t0
tags = _solve.profile_dump()
print("solve_s", dt)
for t, (c, ms) in sorted(tags.items(), key=lambda kv: -kv[1][1])[:14]:
    print("%-40s calls %6d total_ms %9.3f avg_us %8.2f" % (t, c, ms, 1e3 * ms / c))
