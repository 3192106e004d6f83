#!/usr/bin/env python3
"""Summarises the kernel trace of tools_profile_slab.sh (run ON the GPU box) into
gpurun_out/slab_gaps.json: for the sweeps replayed from hipGraphs in the timed region, the average
duration of each of the three sweep kernels and the average gap between consecutive launches
(end of one kernel to start of the next), i.e. where a rank's sweep time goes."""
import json
import os
import re
import sqlite3

ROOT = os.path.dirname(os.path.abspath(__file__))
O = os.path.join(ROOT, "gpurun_out")


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return n.split("(")[0].replace("eps::k::", "")[:80]


def main():
    con = sqlite3.connect(os.path.join(O, "prof_slab", "slab_results.db"))
    rows = [(short(r[0]), r[1], r[2]) for r in con.execute("select name,start,end from kernels order by start")]
    names = ("LassoFusedStreamKernel", "PeerReduceExchangeKernel", "PeerSlabApplyExchangeKernel")
    seq = [r for r in rows if r[0].startswith(names)]
    # sweeps = consecutive triples fused -> reduce -> slab apply on the full-size slab; keep the last
    # 4000 launches (the timed region and the instrumented-free tail; the warm-up solve is small)
    triples = []
    i = 0
    while i + 2 < len(seq):
        a, b, c = seq[i], seq[i + 1], seq[i + 2]
        if a[0].startswith(names[0]) and b[0].startswith(names[1]) and c[0].startswith(names[2]):
            triples.append((a, b, c))
            i += 3
        else:
            i += 1
    big = [t for t in triples if (t[0][2] - t[0][1]) > 20000]  # the 250 MB pass takes > 20 us
    big = big[-400:]
    out = {"sweeps_analysed": len(big)}
    if big:
        dur = [sum((t[k][2] - t[k][1]) for t in big) / len(big) / 1e3 for k in range(3)]
        gap01 = sum(t[1][1] - t[0][2] for t in big) / len(big) / 1e3
        gap12 = sum(t[2][1] - t[1][2] for t in big) / len(big) / 1e3
        nxt = [(big[j + 1][0][1] - big[j][2][2]) / 1e3 for j in range(len(big) - 1)
               if big[j + 1][0][1] - big[j][2][2] < 50000]  # same graph or back-to-back graphs
        period = [(big[j + 1][0][1] - big[j][0][1]) / 1e3 for j in range(len(big) - 1)
                  if big[j + 1][0][1] - big[j][0][1] < 300000]
        out.update({
            "kernel_us": {"fused_pass": dur[0], "reduce_exchange": dur[1], "slab_apply_exchange": dur[2]},
            "gap_us": {"pass->reduce": gap01, "reduce->apply": gap12,
                       "apply->next pass (incl. the residual check every 10th sweep)": sum(nxt) / max(len(nxt), 1)},
            "sweep_period_us_avg": sum(period) / max(len(period), 1),
            "sweep_period_us_median": sorted(period)[len(period) // 2] if period else None,
        })
    json.dump(out, open(os.path.join(O, "slab_gaps.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
