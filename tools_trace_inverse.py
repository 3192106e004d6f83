#!/usr/bin/env python3
"""Per-kernel breakdown of ONE explicit inverse (first PotrfFusedStep launch .. last launch before
the sweeps start) out of the rocprofv3 kernel trace of bench.py (gpurun_out/prof_stats)."""
import re
import sqlite3
import sys

db = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_stats/stats_results.db"
con = sqlite3.connect(db)
rows = list(con.execute("select name, start, end, grid_x, grid_y, workgroup_x from kernels order by start"))
names = [r[0] for r in rows]
# the first FULL-SIZE inverse (the warm-up solve has a small one)
idx = [i for i, n in enumerate(names) if "PotrfFusedStep" in n and rows[i][3] // max(1, rows[i][5]) > 100]
start = idx[0]
end = start
while end < len(rows) and "LassoFused" not in names[end]:
    end += 1
t0 = rows[start][1]
agg = {}
prev_end, gaps = None, 0.0
for r in rows[start:end]:
    n = re.sub(r"\(anonymous namespace\)::", "", r[0]).split("(")[0][:50]
    a = agg.setdefault(n, [0, 0.0])
    a[0] += 1
    a[1] += (r[2] - r[1]) / 1e3
    if prev_end is not None:
        gaps += max(0, r[1] - prev_end) / 1e3
    prev_end = max(prev_end or 0, r[2])
print("span %.2f ms, idle between kernels %.2f ms" % ((rows[end - 1][2] - t0) / 1e6, gaps / 1e3))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-52s %4d %9.1f us" % (k, v[0], v[1]))
for r in rows[start:end]:
    if "GemmSplitF16Kernel" in r[0] or "GemmMfmaF32PipeKernel" in r[0]:
        kind = "split" if "Split" in r[0] else "f32 " + r[0].split("<")[1].split(">")[0]
        z = con.execute("select grid_z from kernels where start=?", (r[1],)).fetchone()[0]
        print("%-16s workgroups %5d x %d x %d  %7.1f us  at %.2f ms" % (kind, r[3] // max(1, r[5]), r[4], z, (r[2] - r[1]) / 1e3, (r[1] - t0) / 1e6))
