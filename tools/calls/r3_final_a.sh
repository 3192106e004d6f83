#!/bin/bash
# round-3 final evidence, part 1: full GPU suite, smoke, bench + rocprof + PMC passes
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
rm -f gpurun_out/under_load_notes.txt
timeout -k 10 900 python -m pytest tests -q -m gpu -x > gpurun_out/final_gpu_suite.log 2>&1; echo "suite rc=$?"; tail -3 gpurun_out/final_gpu_suite.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash tools_profile.sh > gpurun_out/final_profile.log 2>&1; echo "profile rc=$?"; tail -5 gpurun_out/final_profile.log
python - <<'PY'
import json
d=json.loads(open("gpurun_out/bench_default.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","init_s","time_to_eps_s","iters_to_eps")}, d["roofline"]["frac"], d["cpu_baseline"]["value"], d["cpu_baseline"]["kind"])
PY
cat gpurun_out/under_load_notes.txt 2>/dev/null
