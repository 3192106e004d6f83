#!/bin/bash
# round-2 GPU check B: peer tests again, rank-of-8 rehearsal after the K2 / K3 rewrite, CPU quota probe
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
cat /sys/fs/cgroup/cpu.max > gpurun_out/b_cpu.txt 2>&1; nproc >> gpurun_out/b_cpu.txt; echo "OMP=$OMP_NUM_THREADS" >> gpurun_out/b_cpu.txt
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu \
  -k "peer or fused_sweep_sharded or warm_start" > gpurun_out/b_tests.log 2>&1
echo "tests rc=$?" | tee -a gpurun_out/b_tests.log
tail -5 gpurun_out/b_tests.log
for g in 512 384 768; do
EPSILON_HIP_FUSED_GRID=$g timeout -k 10 200 python bench.py --force-sharded --rehearse-ranks 8 --n 6272 --steps 400 --no-cpu-baseline > gpurun_out/b_slab_peer8_g$g.json 2> gpurun_out/b_slab_peer8_g$g.err
echo "slab peer8 grid $g rc=$?"
done
timeout -k 10 200 python bench.py --force-sharded --rehearse-ranks 4 --n 12544 --steps 400 --no-cpu-baseline > gpurun_out/b_slab_peer4.json 2> gpurun_out/b_slab_peer4.err
echo "slab peer4 rc=$?"
timeout -k 10 200 python bench.py --force-sharded --rehearse-ranks 2 --n 25024 --steps 400 --no-cpu-baseline > gpurun_out/b_slab_peer2.json 2> gpurun_out/b_slab_peer2.err
echo "slab peer2 rc=$?"
