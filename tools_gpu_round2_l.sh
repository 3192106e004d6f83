#!/bin/bash
# round-2 GPU check L: fp64 MFMA GEMM - parity in fp64 everywhere it is used + timing
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests/test_gpu_parity.py tests/test_oracle_ref.py tests/test_gpu_abi_ownership.py -x -q -m gpu -k "gemm or algebra or inverse or lasso or nuclear or robust or kron or hinge or f64 or ownership or rebinding or map" > gpurun_out/l_tests.log 2>&1
echo "tests rc=$?"; tail -4 gpurun_out/l_tests.log
python - > gpurun_out/l_gemm_f64.txt 2>&1 <<'PY'
import sys
sys.path.insert(0, ".")
import tools_microbench as t
t.gemm(0, 1, 4096, 4096, 4096, 0, 3, "f64")
t.gemm(0, 1, 4096, 4096, 4096, 2, 3, "f64")
t.gemm(0, 1, 10000, 10000, 50000, 2, 1, "f64")
PY
cat gpurun_out/l_gemm_f64.txt
EPSILON_HIP_GEMM=generic python - > gpurun_out/l_gemm_f64_generic.txt 2>&1 <<'PY'
import sys
sys.path.insert(0, ".")
import tools_microbench as t
t.gemm(0, 1, 4096, 4096, 4096, 0, 2, "f64")
PY
cat gpurun_out/l_gemm_f64_generic.txt
timeout -k 10 300 python bench.py --dtype f64 --steps 50 --no-cpu-baseline > gpurun_out/l_bench_f64.json 2> gpurun_out/l_bench_f64.err; echo "bench f64 rc=$?"
