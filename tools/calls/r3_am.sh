#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout -k 10 1000 python3 -m pytest tests -q -x -m gpu > gpurun_out/r3am_suite.log 2>&1; rc=$?; tail -3 gpurun_out/r3am_suite.log; [ $rc -ne 0 ] && tail -60 gpurun_out/r3am_suite.log
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout -k 10 300 python3 bench.py > gpurun_out/r3am_bench.json 2> gpurun_out/r3am_bench.err; python3 -c "
import json; d=json.loads(open('gpurun_out/r3am_bench.json').read().strip().splitlines()[-1]); print({k:d.get(k) for k in ('value','ms_per_step','init_s','time_to_eps_s')}, d['roofline']['frac'], d['cpu_baseline']['value'], d['cpu_baseline']['kind'])"
