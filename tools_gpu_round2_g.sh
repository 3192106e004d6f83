#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
bash tools_profile.sh; echo "profile rc=$?"
bash tools_profile_slab.sh; echo "slab rc=$?"
timeout -k 10 300 python -m pytest tests/test_mnist_small.py -q -m gpu 2>&1 | tail -3
