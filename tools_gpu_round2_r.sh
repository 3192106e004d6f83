#!/bin/bash
# round-2 GPU check R: general split-f16 GEMM, inverse timing
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_oracle_ref.py -x -q -m gpu -k "f16 or gemm or inverse or lasso_iterates or fused_sweep_matches or warm or nuclear or robust" > gpurun_out/r_tests.log 2>&1
echo "tests rc=$?"; tail -12 gpurun_out/r_tests.log
python tools_microbench.py inverse > gpurun_out/r_inverse.txt 2>&1; head -3 gpurun_out/r_inverse.txt
EPSILON_HIP_GRAM_F16SPLIT=0 python tools_microbench.py inverse > gpurun_out/r_inverse_off.txt 2>&1; head -3 gpurun_out/r_inverse_off.txt
timeout -k 10 300 python bench.py --steps 200 --no-cpu-baseline > gpurun_out/r_n1.json 2> gpurun_out/r_n1.err; echo "n1 rc=$?"
