#!/bin/bash
# Round 3, GPU call J: one-sided Jacobi (no accumulated factor) in the nuclear-norm prox, fp64 fused
# two-block / per-column sweeps, Gram microbenchmark on pattern vs random operands.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -k "fp64_two_block or nuclear or robust_pca or fused_sweep_two_block" -x -q > $O/r3j_t1.log 2>&1; rc=$?
tail -3 $O/r3j_t1.log; [ $rc -ne 0 ] && { tail -40 $O/r3j_t1.log; exit 1; }
timeout -k 10 600 python3 -m pytest tests/test_gpu_prox_more.py -k "nuclear or svd or symmetric or jacobi" -x -q > $O/r3j_t2.log 2>&1; rc=$?
tail -3 $O/r3j_t2.log; [ $rc -ne 0 ] && { tail -40 $O/r3j_t2.log; exit 2; }
timeout -k 10 600 python3 -m pytest tests/test_gpu_full_size.py -k nuclear -x -q > $O/r3j_t3.log 2>&1; rc=$?
tail -3 $O/r3j_t3.log; [ $rc -ne 0 ] && { tail -40 $O/r3j_t3.log; exit 3; }
timeout -k 10 300 python3 tools_bench_nuclear_prox.py 10000 > $O/r3j_nuclear_prox.jsonl 2> $O/r3j_nuclear_prox.err; cat $O/r3j_nuclear_prox.jsonl
EPSILON_HIP_SVD_NO_V=0 timeout -k 10 300 python3 tools_bench_nuclear_prox.py 10000 > $O/r3j_nuclear_prox_twosided.jsonl 2> /dev/null; echo "two-sided:"; cat $O/r3j_nuclear_prox_twosided.jsonl
timeout -k 10 400 python3 bench_rpca.py > $O/r3j_rpca_default.json 2> $O/r3j_rpca.err; python3 -c "
import json; d=json.load(open('$O/r3j_rpca_default.json')); print({k:d[k] for k in ('solve_s','sweeps','state','first_sweep_s','median_sweep_s','constraint_rel_err')})"
timeout -k 10 400 python3 bench_rpca.py --sweeps 5 > $O/r3j_rpca_5sweeps.json 2>> $O/r3j_rpca.err; python3 -c "
import json; d=json.load(open('$O/r3j_rpca_5sweeps.json')); print({k:d[k] for k in ('solve_s','sweeps','sweep_s','constraint_rel_err')})"
echo "--- gemm microbenchmark: repeating pattern"; timeout -k 10 200 python3 tools_microbench.py gemm 2>&1 | grep -E "10000x10000x50000" | tee $O/r3j_gemm_pattern.txt
echo "--- gemm microbenchmark: pseudo-random operands"; EPSILON_HIP_BENCH_RANDOM=1 timeout -k 10 200 python3 tools_microbench.py gemm 2>&1 | grep -E "10000x10000x50000" | tee $O/r3j_gemm_random.txt
