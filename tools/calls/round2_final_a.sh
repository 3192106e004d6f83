#!/bin/bash
# round-2 final evidence, part 1: full GPU suite, bench + rocprof + PMC passes
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests -q -m gpu -x > gpurun_out/final_gpu_suite.log 2>&1; echo "suite rc=$?"; tail -3 gpurun_out/final_gpu_suite.log
bash tools_profile.sh > gpurun_out/final_profile.log 2>&1; echo "profile rc=$?"; tail -5 gpurun_out/final_profile.log
