#!/bin/bash
# Round 3, GPU call M: the GEMM-only polar route of the nuclear-norm prox
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
EPSILON_HIP_SVD_TRACE=1 timeout -k 10 600 python3 -m pytest tests/test_gpu_prox_more.py -k "polar or nuclear or svd" -x -q > $O/r3m_t1.log 2>&1; rc=$?
grep "polar route" $O/r3m_t1.log | head -20; tail -3 $O/r3m_t1.log; [ $rc -ne 0 ] && { tail -50 $O/r3m_t1.log; exit 1; }
EPSILON_HIP_SVD_TRACE=1 timeout -k 10 300 python3 tools_bench_nuclear_prox.py 10000 > $O/r3m_nuclear.jsonl 2> $O/r3m_nuclear.err; cat $O/r3m_nuclear.jsonl; grep "polar route" $O/r3m_nuclear.err | head
python3 - <<'PY'
import sys, json, time, ctypes
sys.path.insert(0, ".")
import numpy as np, torch
from epsilon_amd import _solve, ir
from epsilon_amd.wire import ProxFunction
# kernel-level timing of the polar route at n = 1e4
n = 10000
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(3)
Y = torch.randn(n, 10, generator=g, device=dev) @ torch.randn(10, n, generator=g, device=dev)
mask = torch.rand(n, n, generator=g, device=dev) < 0.1
Y += mask * (10.0 * torch.randn(n, n, generator=g, device=dev))
yb = Y.t().contiguous().double().cpu().numpy().tobytes()
Xv = ir.variable(n, n, "var:X"); expr = ir.prox(ProxFunction.NORM_NUCLEAR, Xv)
_solve.set_option("dtype", "f32")
_solve.set_option("profile_filter", "polar_prox,partial_svd,gemm,syrk")
_solve.profile_enable(True); _solve.profile_reset()
t0 = time.time(); got = _solve.eval_prox(expr.proto.SerializeToString(), 1.0, expr.data, {"var:X": yb}); dt = time.time() - t0
tags = _solve.profile_dump(); _solve.profile_enable(False)
print("eval_prox %.3f s" % dt)
for t, (c, ms) in sorted(tags.items(), key=lambda kv: -kv[1][1])[:12]:
    print("  %-40s x%-4d %9.2f ms" % (t, c, ms))
PY
timeout -k 10 600 python3 -m pytest tests/test_gpu_full_size.py -k nuclear -x -q 2>&1 | tail -3
timeout -k 10 400 python3 bench_rpca.py > $O/r3m_rpca_default.json 2> $O/r3m_rpca.err; python3 -c "
import json; d=json.load(open('$O/r3m_rpca_default.json')); print({k:d[k] for k in ('solve_s','sweeps','state','first_sweep_s','constraint_rel_err')})"
timeout -k 10 400 python3 bench_rpca.py --sweeps 5 > $O/r3m_rpca_5sweeps.json 2>> $O/r3m_rpca.err; python3 -c "
import json; d=json.load(open('$O/r3m_rpca_5sweeps.json')); print({k:d[k] for k in ('solve_s','sweeps','sweep_s','constraint_rel_err')})"
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -k "robust_pca or nuclear" -x -q 2>&1 | tail -3
