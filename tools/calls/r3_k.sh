#!/bin/bash
# Round 3, GPU call K: TV timings after the two-launch aggregate scans; column-sorted one-sided Jacobi
# (sweep counts with and without); the Gram microbenchmark (same-buffer SYRK form) on pattern vs random data.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
bash tools/calls/r3_tv.sh r3k 2>&1 | grep -E "n=|per level|passed|failed"
for srt in 1 0; do
  echo "--- nuclear prox n=1e4, EPSILON_HIP_SVD_SORT=$srt"
  EPSILON_HIP_SVD_SORT=$srt EPSILON_HIP_SVD_VERBOSE=1 timeout -k 10 300 python3 tools_bench_nuclear_prox.py 10000 > $O/r3k_nuclear_sort$srt.jsonl 2> $O/r3k_nuclear_sort$srt.err
  grep rpca $O/r3k_nuclear_sort$srt.jsonl | head -1
  grep -c "block jacobi sweep" $O/r3k_nuclear_sort$srt.err
  grep "block jacobi sweep" $O/r3k_nuclear_sort$srt.err | tail -22 | awk '{printf "%s ", $NF} END {print ""}'
done
timeout -k 10 300 python3 -m pytest tests/test_gpu_prox_more.py -k "nuclear or svd" -x -q 2>&1 | tail -2
echo "--- syrk microbenchmark (same buffer), pattern"; timeout -k 10 200 python3 tools_microbench.py syrk 2>&1 | grep -E "10000x10000x50000" | tee $O/r3k_syrk_pattern.txt
echo "--- syrk microbenchmark (same buffer), pseudo-random"; EPSILON_HIP_BENCH_RANDOM=1 timeout -k 10 200 python3 tools_microbench.py syrk 2>&1 | grep -E "10000x10000x50000" | tee $O/r3k_syrk_random.txt
