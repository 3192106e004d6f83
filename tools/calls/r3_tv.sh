#!/bin/bash
# TV-1D iteration loop on the GPU box: parity tests, timings at four sizes, kernel trace at n = 1e8 and 1e5.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
T=${1:-r3tv}
mkdir -p $O
cd $R
timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -k tv1d -x -q > $O/${T}_tests.log 2>&1 || { tail -30 $O/${T}_tests.log; exit 1; }
tail -2 $O/${T}_tests.log
for n in 100000 1000000 10000000 100000000; do
  it=3; [ $n -le 1000000 ] && it=30
  timeout -k 10 200 python3 bench_tv1d.py --n $n --iters $it --cpu-n 1000 > $O/${T}_n$n.json 2> $O/${T}_n$n.err || { tail -5 $O/${T}_n$n.err; exit 2; }
  python3 - <<PY
import json
a=json.load(open("$O/${T}_n$n.json"))
print("n=$n  %.3f ms (%d levels, %d pieces)" % (1e3*a["seconds"], a["levels"], a["constant_pieces"]))
PY
done
for n in 100000000 100000; do
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/prof_tvx -o tvx -- python3 $R/bench_tv1d.py --n $n --iters 1 --cpu-n 1000 > /dev/null 2> $O/${T}_prof.err )
python3 - <<PY
import sqlite3, re, os
db = "$O/prof_tvx/tvx_results.db"
if os.path.exists(db):
    con = sqlite3.connect(db)
    rows = [(re.sub(r"\(anonymous namespace\)::", "", r[0]).split("(")[0].replace("eps::k::","").replace("void ","")[:60], r[1], r[2]) for r in con.execute("select name,start,end from kernels order by start")]
    ours = [r for r in rows if r[0].startswith(("Tv","Agg","Prefix"))]
    half = ours[len(ours)//2:]
    agg = {}
    for n_, s, e in half:
        a = agg.setdefault(n_, [0, 0.0]); a[0] += 1; a[1] += (e - s) / 1e3
    print("n=$n last prox: span %.3f ms busy %.3f ms, %d kernels" % ((half[-1][2]-half[0][1])/1e6, sum(v[1] for v in agg.values())/1e3, len(half)))
    for n_, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:9]:
        print("  %-40s x%-3d %8.1f us  (%.1f us each)" % (n_, c, t, t / c))
    for key in ("TvClipKernel<float, 1>", "TvClipKernel<float, 0>", "TvBoundKernel<float>", "TvDecodeKernel<float>"):
        print("  per level %-24s" % key, " ".join("%.0f" % ((e - s) / 1e3) for n_, s, e in half if n_ == key))
PY
rm -rf $O/prof_tvx
done
