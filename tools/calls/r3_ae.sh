#!/bin/bash
# Round 3, GPU call AE: small-product kernel
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out
mkdir -p $O
timeout -k 10 1000 python3 -m pytest tests -q -x -m gpu > $O/r3ae_suite.log 2>&1; rc=$?; tail -3 $O/r3ae_suite.log; [ $rc -ne 0 ] && { tail -60 $O/r3ae_suite.log; exit 1; }
for i in 1 2; do timeout -k 10 600 python3 bench_suite.py 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('  %-22s solve %.4f s (init %.4f, loop %.4f) iters %d %s obj %.6g' % (d['problem'], d['solve_s'], d['init_s'], d['loop_s'], d['iterations'], d['state'], d['objective']))"; done
