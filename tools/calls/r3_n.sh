#!/bin/bash
# Round 3, GPU call N: split-f16 GEMM - XCD-aware patch order and LDS-DMA staging, A/B against the
# register-staged kernel in ONE process; parity of every variant first.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
for v in ring lds reg; do
  EPSILON_HIP_GEMM_STAGE=$v timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -k "gram_f16 or gemm_f16 or dense_inverse or gemm_long or gemm_mfma" -x -q > $O/r3n_t_$v.log 2>&1; rc=$?
  echo "stage=$v: $(tail -1 $O/r3n_t_$v.log)"
  [ $rc -ne 0 ] && { tail -40 $O/r3n_t_$v.log; exit 1; }
done
EPSILON_HIP_BENCH_RANDOM=1 timeout -k 10 600 python3 - <<'PY' 2>&1 | tee $O/r3n_ab.txt
import os, sys, ctypes
sys.path.insert(0, ".")
from epsilon_amd import _solve
L = _solve.lib()
_solve.set_option("dtype", "f32")
def gemm(ta, tb, M, N, K, lower, iters=4):
    ms = ctypes.c_double()
    _solve._check(L.eps_bench_gemm(ctypes.c_int(ta), ctypes.c_int(tb), ctypes.c_int64(M), ctypes.c_int64(N),
                                   ctypes.c_int64(K), ctypes.c_int(lower), ctypes.c_int(iters), ctypes.byref(ms)))
    return ms.value
shapes = [("gram NT 1e4x1e4x5e4 same", (0, 1, 10000, 10000, 50000, 2)),
          ("syrk NT 1e4x1e4x5e4 two operands", (0, 1, 10000, 10000, 50000, 1)),
          ("gemm NN 1e4^3", (0, 0, 10000, 10000, 10000, 0)),
          ("gemm NN 1e4^3 lower", (0, 0, 10000, 10000, 10000, 1)),
          ("syrk TN 1e4^3 same", (1, 0, 10000, 10000, 10000, 2)),
          ("gemm NT 4096^3", (0, 1, 4096, 4096, 4096, 0))]
variants = [("reg", "0"), ("lds", "1"), ("ring", "1"), ("ring", "0")]
for rnd in range(2):
    for name, a in shapes:
        row = []
        for st, od in variants:
            os.environ["EPSILON_HIP_GEMM_STAGE"] = st
            os.environ["EPSILON_HIP_GEMM_ORDER"] = od
            row.append("%s/order%s %.3f" % (st, od, gemm(*a)))
        print("round %d  %-34s %s" % (rnd, name, "   ".join(row)), flush=True)
PY
# the polar route with the symmetric products on lower tiles only (default variant)
EPSILON_HIP_SVD_TRACE=1 timeout -k 10 600 python3 -m pytest tests/test_gpu_prox_more.py -k "polar or nuclear or svd" -x -q > $O/r3n_t_polar.log 2>&1; rc=$?
grep "polar route" $O/r3n_t_polar.log | tail -4; tail -2 $O/r3n_t_polar.log; [ $rc -ne 0 ] && { tail -50 $O/r3n_t_polar.log; exit 1; }
timeout -k 10 400 python3 bench_rpca.py > $O/r3n_rpca_default.json 2> $O/r3n_rpca.err; python3 -c "
import json; d=json.load(open('$O/r3n_rpca_default.json')); print({k:d[k] for k in ('solve_s','sweeps','state','first_sweep_s','constraint_rel_err')})"
timeout -k 10 300 python3 bench.py > $O/r3n_bench.json 2> $O/r3n_bench.err; python3 -c "
import json; d=json.loads(open('$O/r3n_bench.json').read().strip().splitlines()[-1]); print({k:d.get(k) for k in ('value','ms_per_step','init_s','time_to_eps_s')}, d.get('roofline'))"
