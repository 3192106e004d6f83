import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import bench
from epsilon_amd import _solve, wire, problems
dev = torch.device("cuda", 0)
_solve.set_option("dtype", "f32")
wp, _ = problems.lasso(2048, 8192, seed=1)
_solve.solve(wp.SerializeToString(), [], wire.SolverParams(max_iterations=50).SerializeToString(), wp.expression_data())
At, b, lam = bench.make_instance(10000, 50000, dev)
prob = bench.build_problem(At, b, lam)
pb, data = prob.SerializeToString(), prob.expression_data()
for i in range(3):
    s = _solve.Solver(pb, wire.SolverParams(max_iterations=50000).SerializeToString(), data)
    torch.cuda.synchronize()
    _solve.profile_enable(True); _solve.profile_reset()
    t0 = time.time(); s.init(); torch.cuda.synchronize(); t1 = time.time()
    prof = _solve.profile_dump(); _solve.profile_enable(False)
    s.run(-1); torch.cuda.synchronize(); t2 = time.time()
    print("init %d: %.4f s, loop %.4f s" % (i, t1 - t0, t2 - t1), flush=True)
    for k, (c, t) in sorted(prof.items(), key=lambda kv: -kv[1][1]):
        if t > 0.05:
            print("    %-44s x%-4d total %8.3f ms" % (k, c, t), flush=True)
    s.close(); del s
