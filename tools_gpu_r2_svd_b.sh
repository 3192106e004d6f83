#!/bin/bash
# block Jacobi on the matrix-core kernels: parity tests, then timings + kernel trace at 10^4
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_prox_more.py tests/test_gpu_parity.py tests/test_oracle_ref.py -x -q -m gpu -k "nuclear or jacobi or symmetric_functions or robust or lambda_max or log_det or semidefinite" > gpurun_out/svd_b_tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/svd_b_tests.log
EPSILON_HIP_SVD_VERBOSE=1 timeout -k 10 600 python tools_microbench.py svd:2048 svd:4096 svd:10000 > gpurun_out/svd_b.jsonl 2> gpurun_out/svd_b.err; echo "rc=$?"; cat gpurun_out/svd_b.jsonl
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_svd -o svd -- python3 $R/tools_microbench.py svd:10000:2 > $R/gpurun_out/svd_b_prof.log 2>&1; echo "prof rc=$?"
cd $R; python3 - <<'PY'
import sqlite3, re
con = sqlite3.connect('gpurun_out/prof_svd/svd_results.db')
rows = list(con.execute("select name,total_calls,total_duration,average,percentage from top_kernels order by total_duration desc"))
with open('gpurun_out/svd_b_kernels.txt', 'w') as f:
    for n, c, t, a, p in rows[:10]:
        line = "%-70s calls %6d total_us %12.1f avg_us %9.2f pct %5.2f" % (re.sub(r"\(anonymous namespace\)::", "", n).split("(")[0][:70], c, t, a, p)
        print(line); f.write(line + "\n")
PY
rm -rf gpurun_out/prof_svd
