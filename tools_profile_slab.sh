#!/bin/bash
# Rank-of-8 rehearsal of the sharded sweep (one GPU plays one rank of 8: 6272-column slab, 1/8 of the
# inverse apply, 8-slot exchange windows): bench line + kernel trace -> per-launch gap table.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
ARGS="--force-sharded --rehearse-ranks 8 --n 6272 --steps 400 --no-cpu-baseline"
cd $R && timeout -k 10 300 python3 bench.py $ARGS > $O/slab_peer8.json 2> $O/slab_peer8.err || exit 1
cd $R && timeout -k 10 300 python3 bench.py --force-sharded --peer off --n 6272 --steps 400 --no-cpu-baseline --no-time-to-eps > $O/slab_rccl.json 2> $O/slab_rccl.err || exit 2
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_slab -o slab -- python3 $R/bench.py $ARGS --no-profile > /dev/null 2> $O/slab_trace.err || exit 3
cd $R && python3 tools_profile_slab.py > $O/slab_gaps.txt 2>&1
rm -rf $O/prof_slab
