#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
EPSILON_HIP_GRAPH_TRACE=1 timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -q -x -s -k "graph_replay_is_bit_identical" > gpurun_out/r3ac.log 2>&1
grep "\[graph\]" gpurun_out/r3ac.log | head -20; tail -3 gpurun_out/r3ac.log
