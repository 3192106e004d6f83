#!/bin/bash
# Config 3 (tv_1d n = 1e8): bench line, kernel trace and HBM counters of the parallel TV prox.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R && timeout -k 10 300 python3 bench_tv1d.py --cpu-n 10000000 > $O/tv1d.json 2> $O/tv1d.err || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_tv -o tv -- python3 $R/bench_tv1d.py --iters 1 --cpu-n 1000 > /dev/null 2> $O/tv_stats.err || exit 2
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d $O/prof_tv_fetch -o tvf -- python3 $R/bench_tv1d.py --iters 1 --cpu-n 1000 > /dev/null 2> $O/tv_fetch.err || exit 3
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d $O/prof_tv_write -o tvw -- python3 $R/bench_tv1d.py --iters 1 --cpu-n 1000 > /dev/null 2> $O/tv_write.err || exit 4
cd $R && python3 tools_profile_tv1d.py > $O/tv1d_profile.txt 2>&1
rm -rf $O/prof_tv $O/prof_tv_fetch $O/prof_tv_write
