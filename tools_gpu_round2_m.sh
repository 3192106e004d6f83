#!/bin/bash
# round-2 GPU check M: SYRK tail split
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gemm or inverse or lasso_iterates or fused_sweep_matches" > gpurun_out/m_tests.log 2>&1
echo "tests rc=$?"; tail -4 gpurun_out/m_tests.log
python - > gpurun_out/m_gemm.txt 2>&1 <<'PY'
import sys
sys.path.insert(0, ".")
import tools_microbench as t
t.gemm(0, 1, 10000, 10000, 50000, 2, 3, "f32")
PY
cat gpurun_out/m_gemm.txt
EPSILON_HIP_SYRK_TAIL=0 python - > gpurun_out/m_gemm_off.txt 2>&1 <<'PY'
import sys
sys.path.insert(0, ".")
import tools_microbench as t
t.gemm(0, 1, 10000, 10000, 50000, 2, 3, "f32")
PY
cat gpurun_out/m_gemm_off.txt
timeout -k 10 300 python bench.py --steps 200 --no-cpu-baseline > gpurun_out/m_n1.json 2> gpurun_out/m_n1.err; echo "n1 rc=$?"
