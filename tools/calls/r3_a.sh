#!/bin/bash
# Round 3, GPU call A: the three-threshold TV-1D (parity, timing against the round-2 binary form),
# then the whole -m gpu suite and the default bench line on the same box.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -k tv1d -x -q > $O/r3a_tv_tests.log 2>&1 || { tail -30 $O/r3a_tv_tests.log; exit 1; }
tail -3 $O/r3a_tv_tests.log
for n in 100000 1000000 10000000 100000000; do
  it=3; [ $n -le 1000000 ] && it=30
  timeout -k 10 200 python3 bench_tv1d.py --n $n --iters $it --cpu-n 1000 > $O/r3a_tv_n$n.json 2> $O/r3a_tv_n$n.err || { tail -5 $O/r3a_tv_n$n.err; exit 2; }
  EPSILON_HIP_TV=binary timeout -k 10 200 python3 bench_tv1d.py --n $n --iters $it --cpu-n 1000 > $O/r3a_tvbin_n$n.json 2> $O/r3a_tvbin_n$n.err || exit 3
  python3 - <<PY
import json
a=json.load(open("$O/r3a_tv_n$n.json")); b=json.load(open("$O/r3a_tvbin_n$n.json"))
print("n=$n  new %.3f ms (%d levels, kkt %s)   binary %.3f ms (%d levels)" % (1e3*a["seconds"], a["levels"], a["kkt"], 1e3*b["seconds"], b["levels"]))
PY
done
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/r3a_gpu_suite.log 2>&1; rc=$?
tail -15 $O/r3a_gpu_suite.log
[ $rc -ne 0 ] && exit 4
timeout -k 10 600 python3 bench.py > $O/r3a_bench.json 2> $O/r3a_bench.err || { tail -20 $O/r3a_bench.err; exit 5; }
python3 - <<PY
import json
d=json.loads(open("$O/r3a_bench.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","init_s","time_to_eps_s","iters_to_eps")}, d["roofline"]["frac"], d["cpu_baseline"]["value"], d["cpu_baseline"]["kind"], d["cpu_baseline"]["init_s_estimate"])
PY
