#!/bin/bash
# Round 3, GPU call P: the full GPU suite on the ring GEMM + general split-K tail, then timings
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out
mkdir -p $O
timeout -k 10 1000 python3 -m pytest tests -q -m gpu -x > $O/r3p_suite.log 2>&1; rc=$?; echo "suite rc=$rc"; tail -3 $O/r3p_suite.log
[ $rc -ne 0 ] && { tail -60 $O/r3p_suite.log; exit 1; }
EPSILON_HIP_BENCH_RANDOM=1 timeout -k 10 300 python3 - <<'PY' 2>&1 | tee $O/r3p_gemm.txt
import sys, ctypes
sys.path.insert(0, ".")
from epsilon_amd import _solve
L = _solve.lib()
_solve.set_option("dtype", "f32")
def gemm(ta, tb, M, N, K, lower, iters=4):
    ms = ctypes.c_double()
    _solve._check(L.eps_bench_gemm(ctypes.c_int(ta), ctypes.c_int(tb), ctypes.c_int64(M), ctypes.c_int64(N),
                                   ctypes.c_int64(K), ctypes.c_int(lower), ctypes.c_int(iters), ctypes.byref(ms)))
    return ms.value
for name, a in [("gram NT 1e4x1e4x5e4 same", (0, 1, 10000, 10000, 50000, 2)), ("syrk NT two operands", (0, 1, 10000, 10000, 50000, 1)),
                ("gemm NN 1e4^3", (0, 0, 10000, 10000, 10000, 0)), ("gemm NN 1e4^3 lower", (0, 0, 10000, 10000, 10000, 1)),
                ("syrk TN 1e4^3 same", (1, 0, 10000, 10000, 10000, 2)), ("gemm NT 4096^3", (0, 1, 4096, 4096, 4096, 0))]:
    print("%-28s %.3f ms" % (name, gemm(*a)), flush=True)
PY
EPSILON_HIP_SVD_TRACE=1 timeout -k 10 300 python3 tools_bench_nuclear_prox.py 10000 > $O/r3p_nuclear.jsonl 2> $O/r3p_nuclear.err; cat $O/r3p_nuclear.jsonl; grep "polar route" $O/r3p_nuclear.err | tail -4
timeout -k 10 400 python3 bench_rpca.py > $O/r3p_rpca_default.json 2> $O/r3p_rpca.err; python3 -c "
import json; d=json.load(open('$O/r3p_rpca_default.json')); print({k:d[k] for k in ('solve_s','sweeps','state','first_sweep_s','constraint_rel_err')})"
timeout -k 10 400 python3 bench_rpca.py --sweeps 8 > $O/r3p_rpca_8sweeps.json 2>> $O/r3p_rpca.err; python3 -c "
import json; d=json.load(open('$O/r3p_rpca_8sweeps.json')); print({k:d[k] for k in ('solve_s','sweeps','sweep_s','constraint_rel_err')})"
timeout -k 10 300 python3 bench.py > $O/r3p_bench.json 2> $O/r3p_bench.err; python3 -c "
import json; d=json.loads(open('$O/r3p_bench.json').read().strip().splitlines()[-1]); print({k:d.get(k) for k in ('value','ms_per_step','init_s','time_to_eps_s')}, d['roofline']['frac'], d['init_breakdown'].get('gram_syrk_ms'))"
