#!/bin/bash
# round-2 final evidence, part 2: slab rehearsal gaps, TV profile, microbenchmarks, config-5 evidence, suite
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
bash tools_profile_slab.sh > gpurun_out/final_slab.log 2>&1; echo "slab rc=$?"
bash tools_profile_tv1d.sh > gpurun_out/final_tv.log 2>&1; echo "tv rc=$?"
python tools_microbench.py prox > gpurun_out/final_prox.jsonl 2> gpurun_out/final_prox.err; echo "prox rc=$?"
python tools_microbench.py syrk gemm inverse > gpurun_out/final_micro.txt 2>&1; tail -12 gpurun_out/final_micro.txt
bash tools/calls/r2_svd_final.sh > gpurun_out/final_svd.log 2>&1; echo "svd rc=$?"; tail -14 gpurun_out/final_svd.log
python tools_bench_nuclear_prox.py 10000 > gpurun_out/nuclear_prox.jsonl 2>/dev/null; echo "nuclear rc=$?"; cat gpurun_out/nuclear_prox.jsonl
python tools_bench_segprox.py 10000000 > gpurun_out/segprox_long.txt 2>&1; echo "segprox rc=$?"
timeout -k 10 300 python bench_suite.py > gpurun_out/bench_suite.jsonl 2>/dev/null; echo "suite-bench rc=$?"; cut -c1-140 gpurun_out/bench_suite.jsonl
