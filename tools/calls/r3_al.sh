#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -q -x -k "constant_atom" 2>&1 | tail -25
