#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out
mkdir -p $O
EPSILON_HIP_INIT_TRACE=1 timeout -k 10 300 python3 bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/r3x_bench.json 2> $O/r3x_bench.err
grep -v "^\[W\|amdgpu.ids" $O/r3x_bench.err | tail -60
