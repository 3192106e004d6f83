#!/bin/bash
# Kernel trace of one explicit inverse at n = 10^4 (tools_microbench.py inverse1) -> per-kernel /
# per-grid totals, split at the phase boundaries, in gpurun_out/inverse_trace.txt
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_inv -o inv -- python3 $R/tools_microbench.py inverse1 > $O/inverse1.log 2> $O/inverse1.err || exit 2
cd $R && python3 tools_trace_inverse.py > $O/inverse_trace.txt 2>&1
