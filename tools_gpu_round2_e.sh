#!/bin/bash
# round-2 GPU check E: shim / ref / mnist tests, fused sweep past 10240 rows + 512-thread form, rehearsals
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_solver_shim.py tests/test_oracle_ref.py tests/test_mnist_small.py -q -m gpu > gpurun_out/e_tests.log 2>&1
echo "tests rc=$?" | tee -a gpurun_out/e_tests.log
tail -15 gpurun_out/e_tests.log
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fused_sweep or peer" > gpurun_out/e_tests_fused.log 2>&1
echo "fused tests rc=$?"; tail -5 gpurun_out/e_tests_fused.log
for cfg in "256 512" "512 256" "512 512"; do set -- $cfg
EPSILON_HIP_FUSED_BLOCK=$1 EPSILON_HIP_FUSED_GRID=$2 timeout -k 10 200 python bench.py --force-sharded --rehearse-ranks 8 --n 6272 --steps 400 --no-cpu-baseline > gpurun_out/e_peer8_b$1_g$2.json 2> gpurun_out/e_peer8_b$1_g$2.err
echo "peer8 block $1 grid $2 rc=$?"
done
for cfg in "256 512" "512 256" "512 512"; do set -- $cfg
EPSILON_HIP_FUSED_BLOCK=$1 EPSILON_HIP_FUSED_GRID=$2 timeout -k 10 200 python bench.py --steps 200 --no-cpu-baseline --no-time-to-eps > gpurun_out/e_n1_b$1_g$2.json 2> gpurun_out/e_n1_b$1_g$2.err
echo "n1 block $1 grid $2 rc=$?"
done
timeout -k 10 300 python bench.py --m 20000 --n 50000 --steps 100 --no-cpu-baseline > gpurun_out/e_m20000.json 2> gpurun_out/e_m20000.err
echo "m20000 rc=$?"
