#!/bin/bash
# round-2 GPU check O: two-block fused sweep
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests/test_gpu_parity.py tests/test_mnist_small.py tests/test_solver_shim.py -x -q -m gpu -k "fused or lasso or two_block or warm or staged or limits or mnist or shim or route or sparse_lasso or lp_type" > gpurun_out/o_tests.log 2>&1
echo "tests rc=$?"; tail -6 gpurun_out/o_tests.log
python - > gpurun_out/o_twoblock.txt 2>&1 <<'PY'
import sys, time
sys.path.insert(0, ".")
import torch
import bench
from epsilon_amd import _solve, wire
dev = torch.device("cuda", 0)
At, b, lam = bench.make_instance(10000, 50000, dev)
prob = bench.build_problem(At, b, lam)
pb, data = prob.SerializeToString(), prob.expression_data()
for fused in ("1", "0"):
    _solve.set_option("fused", fused)
    s = _solve.Solver(pb, wire.SolverParams(max_iterations=10**9, ignore_stopping_criteria=True, solver=1).SerializeToString(), data)
    s.init(); s.run(20); torch.cuda.synchronize(); t0 = time.time(); s.run(100); torch.cuda.synchronize(); dt = time.time() - t0
    print("two-block fused=%s: %.1f us/sweep, %.0f iter/s" % (fused, 1e6 * dt / 100, 100 / dt), flush=True)
    s.close()
_solve.set_option("fused", "1")
s = _solve.Solver(pb, wire.SolverParams(max_iterations=50000, solver=1).SerializeToString(), data)
t0 = time.time(); s.init(); s.run(-1); torch.cuda.synchronize(); dt = time.time() - t0
st = wire.SolverStatus.FromString(s.result()[0])
print("two-block to eps: %.3f s, %d sweeps, state %d" % (dt, st.num_iterations + 1, st.state))
PY
cat gpurun_out/o_twoblock.txt
