cd $GRAFT_REPO_ROOT
for g in auto mfma; do
echo "=== EPSILON_HIP_GEMM=$g"
EPSILON_HIP_GEMM=$g EPSILON_HIP_SVD_TRACE=1 timeout -k 10 300 python3 -m pytest tests/test_gpu_prox_more.py -k "polar and f32" -q 2>&1 | grep -E "polar route|passed|failed|AssertionError" | head -30
done
