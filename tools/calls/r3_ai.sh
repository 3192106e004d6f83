#!/bin/bash
# Round 3, GPU call AI: cached inverse applied from a tile-packed copy
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py -q -x -k "fused or lasso" > $O/r3ai_t.log 2>&1; rc=$?; tail -3 $O/r3ai_t.log; [ $rc -ne 0 ] && { tail -60 $O/r3ai_t.log; exit 1; }
for g in 1 0 1 0; do
EPSILON_HIP_SYMV_PACKED=$g timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-time-to-eps --steps 400 --warmup 40 > $O/r3ai_bench.json 2> $O/r3ai_bench.err; python3 -c "
import json; d=json.loads(open('$O/r3ai_bench.json').read().strip().splitlines()[-1]); print('packed=$g', {k:d.get(k) for k in ('value','ms_per_step','init_s')}, 'pass ms', d['roofline']['avg_launch_ms'], 'tail us %.1f' % (1e3*(d['ms_per_step']-d['roofline']['avg_launch_ms'])))"
done
