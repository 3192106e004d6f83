#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out
mkdir -p $O
for g in 0 1 0 1; do
EPSILON_HIP_GRAPH=$g timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-time-to-eps --steps 400 --warmup 40 > $O/r3y_bench.json 2> $O/r3y_bench.err; python3 -c "
import json; d=json.loads(open('$O/r3y_bench.json').read().strip().splitlines()[-1]); print('graph=$g', {k:d.get(k) for k in ('value','ms_per_step')}, d['roofline']['avg_launch_ms'], d.get('kernels'))"
done
