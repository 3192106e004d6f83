#!/bin/bash
# round-2 GPU check A: new ownership + peer-window tests, N=1 bench, rank-of-8 rehearsal before / after
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_abi_ownership.py tests/test_gpu_parity.py -x -q -m gpu \
  -k "ownership or rebinding or peer or fused_sweep_sharded or rccl_backend or warm_start" > gpurun_out/a_tests.log 2>&1
echo "tests rc=$?" | tee -a gpurun_out/a_tests.log
tail -5 gpurun_out/a_tests.log
timeout -k 10 300 python bench.py --steps 200 --warmup 20 > gpurun_out/a_bench_n1.json 2> gpurun_out/a_bench_n1.err
echo "bench n1 rc=$?"
timeout -k 10 200 python bench.py --force-sharded --peer off --n 6272 --steps 400 --no-time-to-eps --no-cpu-baseline > gpurun_out/a_slab_rccl.json 2> gpurun_out/a_slab_rccl.err
echo "slab rccl rc=$?"
timeout -k 10 200 python bench.py --force-sharded --rehearse-ranks 8 --n 6272 --steps 400 --no-cpu-baseline > gpurun_out/a_slab_peer8.json 2> gpurun_out/a_slab_peer8.err
echo "slab peer8 rc=$?"
EPSILON_HIP_GRAPH=0 timeout -k 10 200 python bench.py --force-sharded --rehearse-ranks 8 --n 6272 --steps 400 --no-cpu-baseline > gpurun_out/a_slab_peer8_eager.json 2> gpurun_out/a_slab_peer8_eager.err
echo "slab peer8 eager rc=$?"
EPSILON_HIP_FUSED_GRID=256 timeout -k 10 200 python bench.py --force-sharded --rehearse-ranks 8 --n 6272 --steps 400 --no-cpu-baseline > gpurun_out/a_slab_peer8_g256.json 2> gpurun_out/a_slab_peer8_g256.err
echo "slab peer8 grid256 rc=$?"
tail -c 600 gpurun_out/a_slab_peer8.err
