#!/bin/bash
# Round 3, GPU call C: TV-1D after the arithmetic rewrite (parity, timings), new tests, whole suite, bench.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -k tv1d -x -q > $O/r3c_tv_tests.log 2>&1 || { tail -30 $O/r3c_tv_tests.log; exit 1; }
tail -2 $O/r3c_tv_tests.log
for n in 100000 1000000 10000000 100000000; do
  it=3; [ $n -le 1000000 ] && it=30
  timeout -k 10 200 python3 bench_tv1d.py --n $n --iters $it --cpu-n 1000 > $O/r3c_tv_n$n.json 2> $O/r3c_tv_n$n.err || { tail -5 $O/r3c_tv_n$n.err; exit 2; }
  python3 - <<PY
import json
a=json.load(open("$O/r3c_tv_n$n.json"))
print("n=$n  %.3f ms (%d levels, %d pieces)" % (1e3*a["seconds"], a["levels"], a["constant_pieces"]))
PY
done
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/prof_tv8 -o tv8 -- python3 $R/bench_tv1d.py --iters 1 --cpu-n 1000 > /dev/null 2> $O/r3c_tv8.err )
python3 - <<PY
import sqlite3, re, os
for tag, db in (("n=1e8", "$O/prof_tv8/tv8_results.db"),):
    if not os.path.exists(db): continue
    con = sqlite3.connect(db)
    rows = [(re.sub(r"\(anonymous namespace\)::", "", r[0]).split("(")[0].replace("eps::k::","").replace("void ","")[:60], r[1], r[2]) for r in con.execute("select name,start,end from kernels order by start")]
    ours = [r for r in rows if r[0].startswith(("Tv","Agg","Prefix"))]
    half = ours[len(ours)//2:]
    agg = {}
    for n_, s, e in half:
        a = agg.setdefault(n_, [0, 0.0]); a[0] += 1; a[1] += (e - s) / 1e3
    print(tag, "last prox: span %.2f ms busy %.2f ms" % ((half[-1][2]-half[0][1])/1e6, sum(v[1] for v in agg.values())/1e3))
    for n_, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:10]:
        print("  %-40s x%-3d %8.1f us  (%.1f us each)" % (n_, c, t, t / c))
PY
rm -rf $O/prof_tv8
timeout -k 10 600 python3 -m pytest tests/test_mnist_small.py tests/test_gpu_under_load.py tests/test_gpu_bench_ranks.py -x -q -rP > $O/r3c_new_tests.log 2>&1; rc=$?
grep -E "passed|failed|error" $O/r3c_new_tests.log | tail -5
[ $rc -ne 0 ] && { tail -60 $O/r3c_new_tests.log; exit 3; }
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q --deselect tests/test_gpu_under_load.py --deselect tests/test_gpu_bench_ranks.py --deselect tests/test_mnist_small.py > $O/r3c_gpu_suite.log 2>&1; rc=$?
tail -8 $O/r3c_gpu_suite.log
[ $rc -ne 0 ] && { tail -60 $O/r3c_gpu_suite.log; exit 4; }
timeout -k 10 600 python3 bench.py > $O/r3c_bench.json 2> $O/r3c_bench.err || { tail -20 $O/r3c_bench.err; exit 5; }
python3 - <<PY
import json
d=json.loads(open("$O/r3c_bench.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","init_s","time_to_eps_s","iters_to_eps")}, d["roofline"]["frac"], d["cpu_baseline"]["value"], d["cpu_baseline"]["kind"], d["cpu_baseline"]["init_s_estimate"], d.get("init_breakdown"))
PY
