R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_inv -o inv -- python3 $R/bench.py --no-cpu-baseline --steps 20 --warmup 5 > /dev/null 2>&1
cd $R && python3 tools_trace_inverse.py gpurun_out/prof_inv/inv_results.db; rm -rf gpurun_out/prof_inv
