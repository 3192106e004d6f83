#!/bin/bash
# config 5 evidence after the matrix-core block Jacobi: full-size prox certificate, bench_rpca
# (default stop and 5 sweeps), SVD microbenchmark, kernel trace of two cold sweeps at 10^4
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_full_size.py -x -q -m gpu -k "nuclear" > gpurun_out/svd_f_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/svd_f_tests.log
timeout -k 10 600 python bench_rpca.py > gpurun_out/rpca_default.json 2> gpurun_out/rpca_default.err; echo "rpca default rc=$?"; cat gpurun_out/rpca_default.json
timeout -k 10 600 python bench_rpca.py --sweeps 5 > gpurun_out/rpca_5sweeps.json 2> gpurun_out/rpca_5sweeps.err; echo "rpca 5 rc=$?"; cat gpurun_out/rpca_5sweeps.json
timeout -k 10 600 python tools_microbench.py svd:2048 svd:4096 svd:10000 > gpurun_out/svd_micro.jsonl 2> gpurun_out/svd_micro.err; echo "rc=$?"; cat gpurun_out/svd_micro.jsonl
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_svd -o svd -- python3 $R/tools_microbench.py svd:10000:2 > $R/gpurun_out/svd_prof.log 2>&1; echo "prof rc=$?"
cd $R; python3 - <<'PY'
import sqlite3, re, csv
con = sqlite3.connect('gpurun_out/prof_svd/svd_results.db')
rows = list(con.execute("select name,total_calls,total_duration,average,percentage from top_kernels order by total_duration desc"))
with open('gpurun_out/svd_kernel_stats.csv', 'w', newline='') as f:
    w = csv.writer(f); w.writerow(["Name", "Calls", "TotalDurationUs", "AverageUs", "Percentage"])
    for n, c, t, a, p in rows[:12]:
        w.writerow([re.sub(r"\(anonymous namespace\)::", "", n).split("(")[0][:90], c, "%.1f" % t, "%.2f" % a, "%.2f" % p])
print(open('gpurun_out/svd_kernel_stats.csv').read())
PY
rm -rf gpurun_out/prof_svd
