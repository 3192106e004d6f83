#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
EPSILON_HIP_INIT_TRACE=2 timeout -k 10 300 python3 bench_suite.py lasso_sparse 2>&1 | grep -v "^\[W\|amdgpu.ids" | cut -c1-220 | tail -60
