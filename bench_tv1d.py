#!/usr/bin/env python3
"""BASELINE.json configs[2]: tv_1d n=10^8 fused-lasso prox on 1 MI355X (a parity-test case, not
the judged bench line).  Generates the reference's signal (python/epopt/problems/tv_1d.py:5-20:
piecewise-constant x0 + unit noise, lam = sqrt(n)) on the device, times the exact parallel prox
through the C-ABI, checks the KKT certificate and times the C DP oracle on a bounded sample."""
import argparse
import ctypes
import json
import sys
import time

sys.path.insert(0, ".")
import numpy as np  # noqa: E402
import torch  # noqa: E402

from epsilon_amd import _solve  # noqa: E402


def make_signal(n, device, seed=0):
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    k = max(int(np.sqrt(n) / 2), 1)
    idx = torch.randint(0, n, (k, 2), generator=g, device=device)
    idx, _ = idx.sort(dim=1)
    steps = 10 * (torch.rand(k, generator=g, device=device, dtype=torch.float64) - 0.5)
    diff = torch.zeros(n + 1, device=device, dtype=torch.float64)
    diff.index_add_(0, idx[:, 0], steps)
    diff.index_add_(0, idx[:, 1], -steps)
    x0 = 1.0 + torch.cumsum(diff[:n], 0)
    return x0 + torch.randn(n, generator=g, device=device, dtype=torch.float64)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=10 ** 8)
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--cpu-n", type=int, default=2 * 10 ** 7)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    tdt = torch.float32 if a.dtype == "f32" else torch.float64
    v = make_signal(a.n, dev).to(tdt)
    x = torch.empty_like(v)
    lam = float(np.sqrt(a.n))
    torch.cuda.synchronize()
    L = _solve.lib()
    lev = ctypes.c_int()
    kind = 1 if a.dtype == "f32" else 2
    times = []
    for _ in range(a.iters + 1):
        t0 = time.time()
        _solve._check(L.eps_tv1d_device(ctypes.c_void_p(v.data_ptr()), ctypes.c_void_p(x.data_ptr()),
                                        ctypes.c_size_t(a.n), ctypes.c_int(kind), ctypes.c_double(lam),
                                        ctypes.byref(lev)))
        times.append(time.time() - t0)
    t = min(times[1:])
    sz = 4 if a.dtype == "f32" else 8
    # KKT certificate on the device result (fp64): c_k = cumsum(x - v)
    xd, vd = x.double(), v.double()
    c = torch.cumsum(xd - vd, 0)
    d = xd[1:] - xd[:-1]
    ck = c[:-1]
    jump = d != 0
    viol_bound = float((ck.abs().max() - lam).clamp(min=0))
    viol_jump = float((ck[jump] - lam * torch.sign(d[jump])).abs().max()) if bool(jump.any()) else 0.0
    viol_end = float(c[-1].abs())
    pieces = int(jump.sum()) + 1
    out = {"config": "tv_1d n=%d %s lam=sqrt(n)" % (a.n, a.dtype), "seconds": t, "levels": lev.value,
           "constant_pieces": pieces, "algorithmic_GBs": 2 * a.n * sz / t / 1e9,
           "frac_of_hbm_peak": 2 * a.n * sz / t / 1e9 / 8000.0,
           "kkt": {"bound": viol_bound, "jump_sign": viol_jump, "end": viol_end, "lam": lam}}
    # CPU: the C DP oracle (fp64, 1 thread) on a bounded prefix of the same signal
    from oracle import c_oracle
    m = min(a.n, a.cpu_n)
    vh = v[:m].double().cpu().numpy()
    t0 = time.time()
    xh = c_oracle.tv1d(vh, lam)
    tc = time.time() - t0
    out["cpu_dp"] = {"n": m, "seconds": tc, "elements_per_s": m / tc,
                     "gpu_elements_per_s": a.n / t, "note": "oracle/lasso_sweep.c tv1d_prox, 1 thread"}
    if m == a.n:
        out["max_abs_diff_vs_cpu_dp"] = float(np.abs(xh - x.double().cpu().numpy()).max())
    print(json.dumps(out))


if __name__ == "__main__":
    main()
