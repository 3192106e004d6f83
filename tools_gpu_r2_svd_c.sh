#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
for ns in "0 0" "1 0" "1 1"; do set -- $ns
cd /tmp && export TMPDIR=/tmp
EPSILON_HIP_SVD_REVERSE=$1 EPSILON_HIP_SVD_NTV=$2 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_svd -o svd -- python3 $R/tools_microbench.py svd:10000:2 > $R/gpurun_out/svd_c_prof.log 2>&1; echo "rev ntv = $ns prof rc=$?"
cd $R; python3 - <<'PY'
import sqlite3, re
con = sqlite3.connect('gpurun_out/prof_svd/svd_results.db')
rows = list(con.execute("select name,total_calls,total_duration,average,percentage from top_kernels order by total_duration desc"))
for n, c, t, a, p in rows[:4]:
    print("%-50s calls %6d total_us %12.1f avg_us %9.2f pct %5.2f" % (re.sub(r"\(anonymous namespace\)::", "", n).split("(")[0][:50], c, t, a, p))
PY
rm -rf gpurun_out/prof_svd
done
