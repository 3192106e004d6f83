#!/bin/bash
# round-2 GPU check C: rewritten TV-1D kernels (parity + n = 1e8), rank-of-8 rehearsal after K1 grid / K3 changes
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "tv" > gpurun_out/c_tests_tv.log 2>&1
echo "tv tests rc=$?" | tee -a gpurun_out/c_tests_tv.log
tail -5 gpurun_out/c_tests_tv.log
timeout -k 10 300 python bench_tv1d.py --iters 3 > gpurun_out/c_tv1d.json 2> gpurun_out/c_tv1d.err
echo "bench_tv1d rc=$?"; tail -c 1500 gpurun_out/c_tv1d.json
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "peer or fused_sweep" > gpurun_out/c_tests_peer.log 2>&1
echo "peer tests rc=$?"; tail -3 gpurun_out/c_tests_peer.log
timeout -k 10 200 python bench.py --force-sharded --rehearse-ranks 8 --n 6272 --steps 400 --no-cpu-baseline > gpurun_out/c_slab_peer8.json 2> gpurun_out/c_slab_peer8.err
echo "slab peer8 rc=$?"
timeout -k 10 200 python bench.py --force-sharded --rehearse-ranks 4 --n 12544 --steps 400 --no-cpu-baseline > gpurun_out/c_slab_peer4.json 2> gpurun_out/c_slab_peer4.err
echo "slab peer4 rc=$?"
