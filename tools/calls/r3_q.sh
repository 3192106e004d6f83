#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/r3q_inv -o inv --output-format csv -- python3 $R/tools_inverse_trace.py > $O/r3q.log 2>&1 || { tail -20 $O/r3q.log; exit 1; }
grep spd_inverse $O/r3q.log
cd $R
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/r3q_inv/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
# last third of the dispatches = the last of the 3 inverses
names = [r["Kernel_Name"] for r in rows]
starts = [i for i, r in enumerate(rows) if "PotrfFusedStep" in r["Kernel_Name"] or "Potrf" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# find the last copy kernel preceding the final inverse: take dispatches after 2/3 of the time span
t0 = int(rows[0]["Start_Timestamp"]); t1 = int(rows[-1]["End_Timestamp"])
agg = collections.OrderedDict()
# split into inverses by the CopyKernel (k::Copy(W, W0)) markers
marks = [i for i, r in enumerate(rows) if "Copy" in r["Kernel_Name"] and int(r["Grid_Size"]) > 10**7]
lo = marks[-1] if marks else 0
sel = rows[lo:]
span = (int(sel[-1]["End_Timestamp"]) - int(sel[0]["Start_Timestamp"])) / 1e3
busy = 0
for r in sel:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    k = r["Kernel_Name"].split("(")[0][-60:]
    a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += d; busy += d
print("last inverse: span %.0f us, kernel busy %.0f us, %d launches" % (span, busy, len(sel)))
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    print("  %-62s x%-5d %9.1f us  avg %.1f" % (k, c, t, t / c))
PY
