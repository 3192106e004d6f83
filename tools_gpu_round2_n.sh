#!/bin/bash
# round-2 GPU check N: fp64 fused pass
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests/test_gpu_parity.py tests/test_gpu_abi_ownership.py tests/test_mnist_small.py -x -q -m gpu -k "fused or lasso or warm or staged or limits or sharded or ownership or rebinding or parameter or degenerate" > gpurun_out/n_tests.log 2>&1
echo "tests rc=$?"; tail -6 gpurun_out/n_tests.log
timeout -k 10 300 python bench.py --dtype f64 --steps 100 --no-cpu-baseline > gpurun_out/n_bench_f64.json 2> gpurun_out/n_bench_f64.err; echo "bench f64 rc=$?"
