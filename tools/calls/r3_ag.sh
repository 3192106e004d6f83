#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -q -x -k "sparse" 2>&1 | tail -3
EPSILON_HIP_INIT_TRACE=2 timeout -k 10 300 python3 bench_suite.py lasso_sparse 2>&1 | grep "host\]\|problem" | cut -c1-200
timeout -k 10 300 python3 bench_suite.py lasso_sparse 2>/dev/null | cut -c1-200
