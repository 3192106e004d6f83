#!/bin/bash
# round-2 GPU check D: caller shim, MNIST fixture, oracle/_ref on the device, elementwise prox GB/s, TV-1D traffic
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_solver_shim.py tests/test_oracle_ref.py tests/test_mnist_small.py -x -q -m gpu > gpurun_out/d_tests.log 2>&1
echo "tests rc=$?" | tee -a gpurun_out/d_tests.log
tail -15 gpurun_out/d_tests.log
timeout -k 10 300 python tools_microbench.py prox > gpurun_out/d_prox_microbench.jsonl 2> gpurun_out/d_prox_microbench.err
echo "prox microbench rc=$?"; cat gpurun_out/d_prox_microbench.jsonl
timeout -k 10 900 bash tools_profile_tv1d.sh
echo "tv profile rc=$?"; head -c 1500 gpurun_out/tv1d_profile.txt
