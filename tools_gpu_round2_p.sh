#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fused or peer or eval_prox or lp_type" > gpurun_out/p_tests.log 2>&1
echo "tests rc=$?"; tail -6 gpurun_out/p_tests.log
