#!/bin/bash
# round-2 GPU check F: exchange spread over the workgroup; mnist / shim tests; rehearsals 8 / 4 / 2
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_solver_shim.py tests/test_mnist_small.py -q -m gpu > gpurun_out/f_tests.log 2>&1
echo "tests rc=$?"; tail -4 gpurun_out/f_tests.log
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "peer or fused_sweep_sharded" > gpurun_out/f_tests_peer.log 2>&1
echo "peer tests rc=$?"; tail -3 gpurun_out/f_tests_peer.log
timeout -k 10 200 python bench.py --force-sharded --rehearse-ranks 8 --n 6272 --steps 400 --no-cpu-baseline > gpurun_out/f_peer8.json 2> gpurun_out/f_peer8.err
echo "peer8 rc=$?"
timeout -k 10 200 python bench.py --force-sharded --rehearse-ranks 4 --n 12544 --steps 400 --no-cpu-baseline > gpurun_out/f_peer4.json 2> gpurun_out/f_peer4.err
echo "peer4 rc=$?"
timeout -k 10 200 python bench.py --force-sharded --rehearse-ranks 2 --n 25024 --steps 400 --no-cpu-baseline > gpurun_out/f_peer2.json 2> gpurun_out/f_peer2.err
echo "peer2 rc=$?"
