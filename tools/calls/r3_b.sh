#!/bin/bash
# Round 3, GPU call B: TV-1D kernel profile (n = 1e8 with HBM counters, n = 1e5 kernel trace), the
# new tests (refinement decision, under-load race coverage, self-launching bench), the whole
# -m gpu suite, the default bench line.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
bash tools_profile_tv1d.sh; echo "tv profile rc=$?"
cp $O/tv1d_profile.json $O/r3b_tv1d_n1e8_profile.json 2>/dev/null
cp $O/tv1d.json $O/r3b_tv1d_n1e8.json 2>/dev/null
cat $O/tv1d_profile.txt 2>/dev/null | cut -c1-220
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/prof_tv5 -o tv5 -- python3 $R/bench_tv1d.py --n 100000 --iters 5 --cpu-n 1000 > $O/r3b_tv_n1e5_under_rocprof.json 2> $O/r3b_tv5.err )
python3 - <<PY
import sqlite3, re, os
db = "$O/prof_tv5/tv5_results.db"
if os.path.exists(db):
    con = sqlite3.connect(db)
    rows = [(re.sub(r"\(anonymous namespace\)::", "", r[0]).split("(")[0].replace("eps::k::","")[:70], r[1], r[2]) for r in con.execute("select name,start,end from kernels order by start")]
    ours = [r for r in rows if not r[0].startswith("at::") and "void at" not in r[0]]
    last = ours[-(len(ours)//6):]   # the last of the 6 prox calls
    print("n=1e5 last prox: %d kernels, span %.1f us, busy %.1f us" % (len(last), (last[-1][2]-last[0][1])/1e3, sum(e-s for _,s,e in last)/1e3))
    agg = {}
    for n_, s, e in last:
        a = agg.setdefault(n_, [0, 0.0]); a[0] += 1; a[1] += (e - s) / 1e3
    for n_, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
        print("  %-70s x%-3d %8.1f us  (%.1f us each)" % (n_, c, t, t / c))
    gaps = [(last[i+1][1]-last[i][2])/1e3 for i in range(len(last)-1)]
    print("  gaps between kernels: mean %.1f us, max %.1f us" % (sum(gaps)/len(gaps), max(gaps)))
PY
rm -rf $O/prof_tv5
timeout -k 10 600 python3 -m pytest tests/test_mnist_small.py tests/test_gpu_under_load.py tests/test_gpu_bench_ranks.py -x -q -rP > $O/r3b_new_tests.log 2>&1; rc=$?
grep -E "passed|failed|error|spd inverse|cond" $O/r3b_new_tests.log | tail -12
[ $rc -ne 0 ] && { tail -40 $O/r3b_new_tests.log; exit 3; }
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/r3b_gpu_suite.log 2>&1; rc=$?
tail -8 $O/r3b_gpu_suite.log
[ $rc -ne 0 ] && { tail -60 $O/r3b_gpu_suite.log; exit 4; }
timeout -k 10 600 python3 bench.py > $O/r3b_bench.json 2> $O/r3b_bench.err || { tail -20 $O/r3b_bench.err; exit 5; }
python3 - <<PY
import json
d=json.loads(open("$O/r3b_bench.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","init_s","time_to_eps_s","iters_to_eps")}, d["roofline"]["frac"], d["cpu_baseline"]["value"], d["cpu_baseline"]["kind"], d["cpu_baseline"]["init_s_estimate"], d.get("init_breakdown"))
PY
