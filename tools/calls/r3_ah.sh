#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
EPSILON_HIP_SVD_VERBOSE=1 timeout -k 10 300 python3 bench_suite.py robust_pca 2>&1 | grep "on-chip" | awk '{print $(NF-1)}' | tr '\n' ' '
echo
