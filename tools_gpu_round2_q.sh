#!/bin/bash
# round-2 GPU check Q: f16-split Gram
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gram_f16 or gemm or lasso or fused_sweep_matches or inverse or warm" > gpurun_out/q_tests.log 2>&1
echo "tests rc=$?"; tail -12 gpurun_out/q_tests.log
python - > gpurun_out/q_gemm.txt 2>&1 <<'PY'
import sys
sys.path.insert(0, ".")
import tools_microbench as t
t.gemm(0, 1, 10000, 10000, 50000, 2, 3, "f32")
PY
cat gpurun_out/q_gemm.txt
timeout -k 10 300 python bench.py --steps 200 --no-cpu-baseline > gpurun_out/q_n1.json 2> gpurun_out/q_n1.err; echo "n1 rc=$?"
