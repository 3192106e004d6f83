#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out
mkdir -p $O
for i in 1 2; do
timeout -k 10 300 python3 bench.py --no-cpu-baseline > $O/r3w_bench_$i.json 2> $O/r3w_bench.err; python3 -c "
import json; d=json.loads(open('$O/r3w_bench_$i.json').read().strip().splitlines()[-1]); print({k:d.get(k) for k in ('value','ms_per_step','init_s','time_to_eps_s')}, d['roofline']['frac'], d['init_breakdown'].get('gram_syrk_ms'), {k:v for k,v in d['init_breakdown'].items() if k.endswith('_ms')})"
done
