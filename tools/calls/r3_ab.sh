#!/bin/bash
# Round 3, GPU call AB: hipGraph replay of the generic operator path
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out
mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -q -x -k "graph_replay_is_bit_identical" > $O/r3ab_t0.log 2>&1; rc=$?; tail -3 $O/r3ab_t0.log; [ $rc -ne 0 ] && { tail -60 $O/r3ab_t0.log; exit 1; }
timeout -k 10 1000 python3 -m pytest tests -q -x -m gpu > $O/r3ab_suite.log 2>&1; rc=$?; tail -3 $O/r3ab_suite.log; [ $rc -ne 0 ] && { tail -60 $O/r3ab_suite.log; exit 1; }
for g in 1 0; do echo "EPSILON_HIP_GRAPH_GENERIC=$g"; EPSILON_HIP_GRAPH_GENERIC=$g timeout -k 10 600 python3 bench_suite.py 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('  %-22s solve %.4f s (init %.4f, loop %.4f) iters %d %s' % (d['problem'], d['solve_s'], d['init_s'], d['loop_s'], d['iterations'], d['state']))"; done
