#!/bin/bash
# round-2 GPU check H: pipelined residual checks + one-launch norms + float4 ReducePartials
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests/test_gpu_parity.py tests/test_gpu_abi_ownership.py -x -q -m gpu -k "fused or peer or lasso or warm or staged or limits or sharded or ownership or rebinding or consensus" > gpurun_out/h_tests.log 2>&1
echo "tests rc=$?"; tail -4 gpurun_out/h_tests.log
timeout -k 10 300 python bench.py --steps 200 --no-cpu-baseline > gpurun_out/h_n1.json 2> gpurun_out/h_n1.err; echo "n1 rc=$?"
EPSILON_HIP_PIPELINE_CHECKS=0 timeout -k 10 300 python bench.py --steps 200 --no-cpu-baseline > gpurun_out/h_n1_nopipe.json 2> gpurun_out/h_n1_nopipe.err; echo "n1 nopipe rc=$?"
timeout -k 10 200 python bench.py --force-sharded --rehearse-ranks 8 --n 6272 --steps 400 --no-cpu-baseline > gpurun_out/h_peer8.json 2> gpurun_out/h_peer8.err; echo "peer8 rc=$?"
EPSILON_HIP_PIPELINE_CHECKS=0 timeout -k 10 200 python bench.py --force-sharded --rehearse-ranks 8 --n 6272 --steps 400 --no-cpu-baseline > gpurun_out/h_peer8_nopipe.json 2> gpurun_out/h_peer8_nopipe.err; echo "peer8 nopipe rc=$?"
python tools_microbench.py gemv > gpurun_out/h_gemv.txt 2>&1; tail -8 gpurun_out/h_gemv.txt
