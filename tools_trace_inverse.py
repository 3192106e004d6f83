#!/usr/bin/env python3
"""Reads gpurun_out/prof_inv/inv_results.db (tools_trace_inverse.sh): kernels of the LAST inverse
in launch order, aggregated per phase (potrf | trtri | lauum) by kernel name and grid."""
import os
import re
import sqlite3

ROOT = os.path.dirname(os.path.abspath(__file__))
con = sqlite3.connect(os.path.join(ROOT, "gpurun_out", "prof_inv", "inv_results.db"))
views = [r[0] for r in con.execute("select name from sqlite_master where type in ('view','table')")]
cand = [v for v in views if v == "kernels"] or [v for v in views if "kernel_dispatch" in v]
v = cand[0]
cols = [r[1] for r in con.execute("pragma table_info(%s)" % v)]
print("view", v, cols)
name = "name" if "name" in cols else "kernel_name"
gx = [c for c in cols if c in ("grid_x", "grid_size_x", "grid_size")]
gx = gx[0] if gx else "0"
wx = [c for c in cols if c in ("workgroup_x", "workgroup_size_x", "workgroup_size")]
wx = wx[0] if wx else "1"
rows = list(con.execute("select %s, start, end, %s, %s from %s order by start" % (name, gx, wx, v)))


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return n.split("(")[0][:70]


# last inverse = from the last PotrfDiagStep with the largest trailing run
idx = [i for i, r in enumerate(rows) if "PotrfDiagStep" in r[0] or "PotrfDiagKernel" in r[0]]
# runs of potrf steps: a new inverse starts where the gap in index is large after TrtriDiagBlocks
starts = [idx[0]]
for a, b in zip(idx, idx[1:]):
    if any("SymmetrizeFromLower" in rows[j][0] for j in range(a, b)):
        starts.append(b)
lo = starts[-1]
seg = rows[lo:]
hi = max(i for i, r in enumerate(seg) if "SymmetrizeFromLower" in r[0])
seg = seg[:hi + 1]
phase, agg, wall = "potrf", {}, {}
t_phase = seg[0][1]
for n, s, e, g, w in seg:
    sn = short(n)
    if "TrtriDiagBlocks" in sn or ("Copy2D" in sn and phase == "potrf"):
        pass
    key = (phase, sn, int(g) // max(int(w), 1) if w else g)
    a = agg.setdefault(key, [0, 0.0])
    a[0] += 1
    a[1] += (e - s) / 1e3
    wall.setdefault(phase, [s, e])[1] = e
    if "TrtriDiagBlocks" in sn:
        phase = "trtri"
    elif phase == "trtri" and "FillKernel" in sn:
        phase = "lauum"
for ph in ("potrf", "trtri", "lauum"):
    if ph not in wall:
        continue
    print("== %s: wall %.2f ms" % (ph, (wall[ph][1] - wall[ph][0]) / 1e6))
    items = [(k, v) for k, v in agg.items() if k[0] == ph]
    byname = {}
    for (p, n, g), (c, t) in items:
        b = byname.setdefault(n, [0, 0.0])
        b[0] += c
        b[1] += t
    for n, (c, t) in sorted(byname.items(), key=lambda x: -x[1][1]):
        print("  %-70s calls %5d  total %9.1f us  avg %8.1f us" % (n, c, t, t / c))
    big = sorted(items, key=lambda x: -x[1][1])[:12]
    for (p, n, g), (c, t) in big:
        print("     grid %6s  %-60s calls %4d total %9.1f us avg %8.1f" % (g, n[:60], c, t, t / c))
