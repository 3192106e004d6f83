#!/bin/bash
# Round 3, GPU call AJ: LDS-staged epilogue of the ring GEMM
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_under_load.py -q -x -k "gram_f16 or gemm_f16 or dense_inverse or gemm_long or gemm_mfma or inverse" > $O/r3aj_t.log 2>&1; rc=$?; tail -3 $O/r3aj_t.log; [ $rc -ne 0 ] && { tail -60 $O/r3aj_t.log; exit 1; }
EPSILON_HIP_BENCH_RANDOM=1 timeout -k 10 300 python3 tools_microbench.py syrk gemm inverse 2>&1 | tail -12
timeout -k 10 300 python3 bench.py --no-cpu-baseline > $O/r3aj_bench.json 2> $O/r3aj_bench.err; python3 -c "
import json; d=json.loads(open('$O/r3aj_bench.json').read().strip().splitlines()[-1]); print({k:d.get(k) for k in ('value','ms_per_step','init_s','time_to_eps_s')}, {k:v for k,v in d['init_breakdown'].items() if k.endswith('_ms')})"
