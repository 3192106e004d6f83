#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out
mkdir -p $O
timeout -k 10 900 python3 bench_suite.py > $O/r3s_suite.jsonl 2> $O/r3s_suite.err; python3 - <<'PY'
import json
for l in open("gpurun_out/r3s_suite.jsonl"):
    d = json.loads(l)
    print("%-22s solve %.4f s (init %.4f, loop %.4f) iters %d %s obj %.5g  ref %s s obj %s" % (d["problem"], d["solve_s"], d["init_s"], d["loop_s"], d["iterations"], d["state"], d["objective"], d["reference"].get("ref_total_s", d["reference"].get("ref_solve_s")), d["reference"].get("ref_objective")))
PY
