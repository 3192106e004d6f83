#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -q -x -k "symmetric_kernel or fused_sweep or two_block" 2>&1 | tail -3
