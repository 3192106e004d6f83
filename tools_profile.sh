#!/bin/bash
# Collects the judged evidence on the GPU box: bench line, rocprofv3 kernel stats, PMC traffic.
# Usage (through gpurun):  bash tools_profile.sh   -> files under gpurun_out/
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $R && timeout -k 10 400 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_stats -o stats -- python3 $R/bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/rocprof_stats.err || exit 2
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d $O/prof_fetch -o fetch -- python3 $R/bench.py --no-cpu-baseline --no-time-to-eps --steps 20 --warmup 5 > /dev/null 2> $O/rocprof_fetch.err || exit 3
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d $O/prof_write -o write -- python3 $R/bench.py --no-cpu-baseline --no-time-to-eps --steps 20 --warmup 5 > /dev/null 2> $O/rocprof_write.err || exit 4
find $O/prof_stats $O/prof_fetch $O/prof_write -type f | head -40
