#!/usr/bin/env python3
"""Summarises the rocprofv3 databases of tools_profile_tv1d.sh (run ON the GPU box: the databases
are too large to travel) into gpurun_out/tv1d_profile.json: per-kernel time of the last prox,
HBM bytes (2*FETCH_SIZE + WRITE_SIZE, KB units, gfx950 correction) and the resulting GB/s."""
import json
import os
import re
import sqlite3

ROOT = os.path.dirname(os.path.abspath(__file__))
O = os.path.join(ROOT, "gpurun_out")


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return n.split("(")[0].replace("eps::k::", "")[:90]


def main():
    con = sqlite3.connect(os.path.join(O, "prof_tv", "tv_results.db"))
    rows = [(short(r[0]), r[1], r[2]) for r in con.execute("select name,start,end from kernels order by start")]
    # the prox is called iters+1 = 2 times; keep the kernels of the library only, second half
    ours = [r for r in rows if not r[0].startswith("at::") and "rocclr" not in r[0] and "rocprim" not in r[0].lower()
            and not r[0].startswith("hipcub") and "Cijk" not in r[0]]
    half = ours[len(ours) // 2:]
    agg = {}
    for n, s, e in half:
        a = agg.setdefault(n, [0, 0.0])
        a[0] += 1
        a[1] += (e - s) / 1e3
    span_ms = (half[-1][2] - half[0][1]) / 1e6
    busy_ms = sum(v[1] for v in agg.values()) / 1e3
    traffic = {}
    for nm, db, dbn in (("FETCH_SIZE", "prof_tv_fetch", "tvf"), ("WRITE_SIZE", "prof_tv_write", "tvw")):
        c = sqlite3.connect(os.path.join(O, db, dbn + "_results.db"))
        q = ("select kernel_name, count(*), sum(value) from counters_collection where counter_name=? "
             "group by kernel_name")
        for kn, cnt, tot in c.execute(q, (nm,)):
            traffic.setdefault(short(kn), {})[nm] = (cnt, tot)
    out = {"span_ms_last_prox": span_ms, "kernel_busy_ms_last_prox": busy_ms, "kernels": []}
    tot_bytes = 0.0
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        tr = traffic.get(n, {})
        f = tr.get("FETCH_SIZE", (0, 0.0))
        w = tr.get("WRITE_SIZE", (0, 0.0))
        # counters were collected over both prox calls of the process: scale to one
        calls_all = max(f[0], w[0], 1)
        byt = (2 * f[1] + w[1]) * 1024.0 * c / calls_all
        tot_bytes += byt
        out["kernels"].append({"kernel": n, "launches": c, "total_us": round(t, 1),
                               "hbm_bytes": byt, "GBs": byt / (t * 1e-6) / 1e9 if t else None})
    out["hbm_bytes_last_prox"] = tot_bytes
    out["avg_GBs_over_busy_time"] = tot_bytes / (busy_ms * 1e-3) / 1e9
    json.dump(out, open(os.path.join(O, "tv1d_profile.json"), "w"), indent=1)
    print(json.dumps({k: out[k] for k in ("span_ms_last_prox", "kernel_busy_ms_last_prox", "hbm_bytes_last_prox",
                                          "avg_GBs_over_busy_time")}))
    for k in out["kernels"][:12]:
        print(k)


if __name__ == "__main__":
    main()
