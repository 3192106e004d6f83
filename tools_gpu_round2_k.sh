#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29711 bench.py --gpus 2 --comm host --rows 2048 --cols 8192 --steps 100 --warmup 10 > gpurun_out/k_bench2_host.json 2> gpurun_out/k_bench2_host.err
echo "2-rank host bench rc=$?"; tail -c 1500 gpurun_out/k_bench2_host.json; tail -c 600 gpurun_out/k_bench2_host.err
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 3 --master-addr 127.0.0.1 --master-port 29712 bench.py --gpus 3 --comm host --rows 2048 --cols 8192 --steps 100 --warmup 10 > gpurun_out/k_bench3_host.json 2> gpurun_out/k_bench3_host.err
echo "3-rank host bench rc=$?"; tail -c 900 gpurun_out/k_bench3_host.json; tail -c 400 gpurun_out/k_bench3_host.err
