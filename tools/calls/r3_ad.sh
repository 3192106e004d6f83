#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out
mkdir -p $O
EPSILON_HIP_GRAPH_TRACE=1 timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -q -x -k "graph_replay_is_bit_identical or more_benchmark or lp_type" > $O/r3ad_t0.log 2>&1; rc=$?; grep "abandoned" $O/r3ad_t0.log | head -3; tail -3 $O/r3ad_t0.log; [ $rc -ne 0 ] && { tail -60 $O/r3ad_t0.log; exit 1; }
timeout -k 10 600 python3 bench_suite.py 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('  %-22s solve %.4f s (init %.4f, loop %.4f) iters %d %s' % (d['problem'], d['solve_s'], d['init_s'], d['loop_s'], d['iterations'], d['state']))"
python3 - <<'PY'
# a long run: the reference's lasso_sparse problem for 1000 sweeps, graph replay against eager launches
import sys, time, os, subprocess, json
code = r'''
import sys, time
sys.path.insert(0, ".")
import numpy as np
import bench_suite
from epsilon_amd import _solve, wire, problems
_solve.set_option("dtype", "f32")
wp, _ = problems.lasso(256, 1024, seed=1)
_solve.solve(wp.SerializeToString(), [], wire.SolverParams(max_iterations=20).SerializeToString(), wp.expression_data())
for name in ("lasso_sparse", "mnist", "mv_lasso"):
    prob, obj, ref = dict(bench_suite.SUITE)[name]()
    pb, data = prob.SerializeToString(), prob.expression_data()
    sp = wire.SolverParams(max_iterations=1000, ignore_stopping_criteria=True).SerializeToString()
    st, x = _solve.solve(pb, [], sp, data)
    S = wire.SolverStatus.FromString(st)
    print("%s: 1000 sweeps loop %.4f s; graph stats %s" % (name, S.timing.total_time - S.timing.init_time, _solve.graph_stats(reset=True)), flush=True)
'''
for g in ("1", "0"):
    env = dict(os.environ, EPSILON_HIP_GRAPH_GENERIC=g)
    print("EPSILON_HIP_GRAPH_GENERIC=" + g, flush=True)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    print("\n".join(l for l in r.stdout.splitlines() if "sweeps" in l) or r.stderr[-800:], flush=True)
PY
