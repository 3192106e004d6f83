#!/bin/bash
# Round 3, GPU call R: fp64 peer exchange
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_ranks.py -q -x -k "sharded or peer or bench or ranks" > $O/r3r_t.log 2>&1; rc=$?; tail -4 $O/r3r_t.log
[ $rc -ne 0 ] && { tail -80 $O/r3r_t.log; exit 1; }
# one rank of 8 in fp64 through the window (timing rehearsal, as profiles/r02_slab_rank_of_8_*)
timeout -k 10 300 python3 bench.py --force-sharded --rehearse-ranks 8 --n 6272 --steps 200 --dtype f64 --no-cpu-baseline > $O/r3r_slab_f64.json 2> $O/r3r_slab_f64.err; tail -c 1500 $O/r3r_slab_f64.json; tail -3 $O/r3r_slab_f64.err
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -q -x -k "more_benchmark_problems" 2>&1 | tail -3
timeout -k 10 900 python3 bench_suite.py > $O/r3r_suite.jsonl 2> $O/r3r_suite.err; python3 - <<'PY'
import json
for l in open("gpurun_out/r3r_suite.jsonl"):
    d = json.loads(l)
    print("%-22s solve %.4f s (init %.4f, loop %.4f) iters %d %s obj %.5g  ref %s s obj %s" % (d["problem"], d["solve_s"], d["init_s"], d["loop_s"], d["iterations"], d["state"], d["objective"], d["reference"].get("ref_total_s", d["reference"].get("ref_solve_s")), d["reference"].get("ref_objective")))
PY
tail -3 $O/r3r_suite.err
