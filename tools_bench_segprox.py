#!/usr/bin/env python3
"""Prox operators that solve ONE long slice (n = 1e7 by default): wall time of eval_prox and the
three most expensive profile tags (live HIP-event timers)."""
import sys, time, json
import numpy as np
sys.path.insert(0, "/root/repo")
import os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from epsilon_amd import _solve, ir
from epsilon_amd.wire import ProxFunction
_solve.set_option("dtype", "f32")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10**7
rng = np.random.RandomState(0)
v = rng.randn(n)
for name in ("MAX", "SUM_LARGEST", "LOG_SUM_EXP", "NORM_2", "SUM_EXP", "NORM_1"):
    try:
        X = ir.variable(n, 1, "var:x")
        kw = {}
        extra = {}
        if name == "SUM_LARGEST":
            from epsilon_amd import wire
            extra["sum_largest_params"] = wire.SumLargestParams(k=10)
        e = ir.prox(getattr(ProxFunction, name), X, **extra)
        fb = e.proto.SerializeToString()
        data = {"var:x": v.tobytes()}
        for rep in range(2):
            _solve.profile_enable(True); _solve.profile_reset()
            t0 = time.time()
            got = _solve.eval_prox(fb, 1.0, e.data, data)
            dt = time.time() - t0
            tags = _solve.profile_dump(); _solve.profile_enable(False)
        top = sorted(tags.items(), key=lambda kv: -kv[1][1])[:3]
        print(name, "wall %.3f s" % dt, [(t, round(ms, 3)) for t, (c, ms) in top], flush=True)
    except Exception as ex:
        print(name, "ERR", str(ex)[:200], flush=True)
for name in ("MAX", "SUM_LARGEST", "LOG_SUM_EXP", "NORM_2", "SUM_EXP", "NORM_1"):
    try:
        X = ir.variable(n, 1, "var:x")
        t = ir.variable(1, 1, "var:t")
        extra = {}
        if name == "SUM_LARGEST":
            from epsilon_amd import wire
            extra["sum_largest_params"] = wire.SumLargestParams(k=10)
        e = ir.prox(getattr(ProxFunction, name), X, t, epigraph=True, **extra)
        fb = e.proto.SerializeToString()
        data = {"var:x": v.tobytes(), "var:t": np.array([0.5]).tobytes()}
        for rep in range(2):
            _solve.profile_enable(True); _solve.profile_reset()
            t0 = time.time()
            got = _solve.eval_prox(fb, 1.0, e.data, data)
            dt = time.time() - t0
            tags = _solve.profile_dump(); _solve.profile_enable(False)
        top = sorted(tags.items(), key=lambda kv: -kv[1][1])[:3]
        print(name, "epigraph wall %.3f s" % dt, [(t_, round(ms, 3)) for t_, (c, ms) in top], flush=True)
    except Exception as ex:
        print(name, "epigraph ERR", str(ex)[:200], flush=True)
