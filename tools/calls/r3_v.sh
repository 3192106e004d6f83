#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/r3v_tv -o tv --output-format csv -- python3 $R/tools_tv_solve_trace.py > $O/r3v.log 2>&1 || { tail -20 $O/r3v.log; exit 1; }
grep "solve " $O/r3v.log
cd $R
python3 - <<'PY'
import csv, glob, collections, re
f = glob.glob("gpurun_out/r3v_tv/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
agg = collections.OrderedDict(); busy = 0
def nm(r):
    k = r["Kernel_Name"].replace("eps::k::(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(.*", "", k)[:60]
for r in rows:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a = agg.setdefault(nm(r), [0, 0.0]); a[0] += 1; a[1] += d; busy += d
print("kernel busy %.1f ms over %d launches" % (busy / 1e3, len(rows)))
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
    print("  %-62s x%-5d %9.1f us  avg %.1f" % (k, c, t, t / c))
PY
rm -rf $O/r3v_tv
