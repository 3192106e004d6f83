#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_prox_more.py tests/test_oracle_ref.py -q -x -k "nuclear or svd or ortho or lambda_max or semidefinite or log_det or eigensolver" > $O/r3aa_t1.log 2>&1; rc=$?; tail -3 $O/r3aa_t1.log; [ $rc -ne 0 ] && { tail -60 $O/r3aa_t1.log; exit 1; }
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -q -x -k "robust_pca or nuclear or covsel or more_benchmark" > $O/r3aa_t2.log 2>&1; rc=$?; tail -3 $O/r3aa_t2.log; [ $rc -ne 0 ] && { tail -60 $O/r3aa_t2.log; exit 1; }
for i in 1 2; do timeout -k 10 300 python3 bench_suite.py robust_pca 2>/dev/null | cut -c1-230; done
EPSILON_HIP_SVD_NO_V=0 timeout -k 10 300 python3 bench_suite.py robust_pca 2>/dev/null | cut -c1-230
