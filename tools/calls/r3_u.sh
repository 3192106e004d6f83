#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_full_size.py tests/test_gpu_abi_ownership.py -q -x 2>&1 | tail -3
timeout -k 10 300 python3 tools_bench_nuclear_prox.py 10000 > $O/r3u_nuclear.jsonl 2> $O/r3u_nuclear.err; cat $O/r3u_nuclear.jsonl
timeout -k 10 300 python3 - <<'PY'
import sys, time, numpy as np
sys.path.insert(0, ".")
from epsilon_amd import _solve, problems, wire
_solve.set_option("dtype", "f32")
# the whole boundary of a large solve: tv_1d at n = 1e8 through eps_solve (upload, 11 sweeps, result copy)
prob, info = problems.tv_1d(10 ** 8, seed=0)
pb, data = prob.SerializeToString(), prob.expression_data()
for rep in range(2):
    t0 = time.time()
    st, x = _solve.solve(pb, [], wire.SolverParams(max_iterations=50).SerializeToString(), data)
    t1 = time.time()
    S = wire.SolverStatus.FromString(st)
    print("tv_1d n=1e8 solve() %.3f s (init %.3f, loop %.3f) iters %d state %d; result bytes %d" % (
        t1 - t0, S.timing.init_time, S.timing.total_time - S.timing.init_time, S.num_iterations + 1, S.state,
        sum(len(v) for v in x.values())), flush=True)
PY
