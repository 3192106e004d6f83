#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out
mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_abi_ownership.py tests/test_mnist_small.py -q -x 2>&1 | tail -3
timeout -k 10 300 python3 bench_mnist.py > $O/r3t_mnist.json 2> $O/r3t_mnist.err; python3 -c "
import json; d=json.loads(open('$O/r3t_mnist.json').read().strip().splitlines()[-1]); print({k:v for k,v in d.items() if k in ('value','create_s','init_s','solve_s','time_to_eps_s','iterations','state','sweep_s','wall_s')}); print(list(d.keys()))"
