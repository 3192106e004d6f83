#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
EPSILON_HIP_SVD_TRACE=1 rocprofv3 --kernel-trace --stats -d $O/r3z -o rp --output-format csv -- python3 $R/bench_suite.py robust_pca > $O/r3z.log 2> $O/r3z.err || { tail -20 $O/r3z.err; exit 1; }
cut -c1-200 $O/r3z.log
grep -c "on-chip" $O/r3z.err; grep "on-chip" $O/r3z.err | awk '{print $(NF-1)}' | sort -n | uniq -c
cd $R
python3 - <<'PY'
import csv, glob, collections, re
f = glob.glob("gpurun_out/r3z/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
agg = collections.OrderedDict(); busy = 0
def nm(r):
    k = r["Kernel_Name"].replace("eps::k::(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(.*", "", k)[:60]
for r in rows:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a = agg.setdefault(nm(r), [0, 0.0]); a[0] += 1; a[1] += d; busy += d
print("kernel busy %.1f ms over %d launches" % (busy / 1e3, len(rows)))
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
    print("  %-62s x%-5d %9.1f us  avg %.1f" % (k, c, t, t / c))
PY
rm -rf $O/r3z
