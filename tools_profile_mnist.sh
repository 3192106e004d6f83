R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_mn -o mn -- python3 $R/bench_mnist.py > /dev/null 2>&1
cd $R; python3 - <<'PY'
import sqlite3, re
con = sqlite3.connect('gpurun_out/prof_mn/mn_results.db')
rows = list(con.execute("select name,total_calls,total_duration,average,percentage from top_kernels order by total_duration desc"))
for n, c, t, a, p in rows[:12]:
    print("%-60s calls %5d total_us %10.1f avg_us %9.2f" % (re.sub(r"\(anonymous namespace\)::", "", n).split("(")[0][:60], c, t, a))
PY
rm -rf gpurun_out/prof_mn
