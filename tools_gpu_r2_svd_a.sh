#!/bin/bash
# SVD baseline: timings at 2048 / 4096 / 10^4 and a kernel trace of two cold sweeps at 10^4
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout -k 10 600 python tools_microbench.py svd:2048 svd:4096 svd:10000 > gpurun_out/svd_a.jsonl 2> gpurun_out/svd_a.err; echo "rc=$?"; cat gpurun_out/svd_a.jsonl
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_svd -o svd -- python3 $R/tools_microbench.py svd:10000:2 > $R/gpurun_out/svd_a_prof.log 2>&1; echo "prof rc=$?"
cd $R; python3 - <<'PY'
import sqlite3, re
con = sqlite3.connect('gpurun_out/prof_svd/svd_results.db')
rows = list(con.execute("select name,total_calls,total_duration,average,percentage from top_kernels order by total_duration desc"))
with open('gpurun_out/svd_a_kernels.txt', 'w') as f:
    for n, c, t, a, p in rows[:16]:
        line = "%-90s calls %6d total_us %12.1f avg_us %9.2f pct %5.2f" % (re.sub(r"\(anonymous namespace\)::", "", n).split("(")[0][:90], c, t, a, p)
        print(line); f.write(line + "\n")
PY
rm -rf gpurun_out/prof_svd
