#!/usr/bin/env python3
"""Nuclear-norm prox of an n x n matrix resident in HBM (device blob), two spectra:
  low  : rank-10 part (sigma ~ n / 100) + dense noise with ||E||_2 ~ 2 < lam = 4  -> the thresholded
         partial SVD applies (10 singular values above lam)
  rpca : the reference's robust-PCA matrix (problems/robust_pca.py:5-22), lam = 1 -> numerically
         full rank above lam, the partial route gives up and the block Jacobi SVD runs.
One JSON line per case: seconds of eval_prox through the C ABI (device blob in, host result out is
excluded: the call is timed with the result left on the device when the binding offers it)."""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from epsilon_amd import _solve, ir  # noqa: E402
from epsilon_amd.wire import ProxFunction  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    _solve.set_option("dtype", "f32")
    Xv = ir.variable(n, n, "var:X")
    expr = ir.prox(ProxFunction.NORM_NUCLEAR, Xv)
    fb = expr.proto.SerializeToString()
    for case in ("low", "rpca"):
        if case == "low":
            Y = (torch.randn(n, 10, generator=g, device=dev) @ torch.randn(10, n, generator=g, device=dev)) / 100.0
            Y += 0.01 * torch.randn(n, n, generator=g, device=dev)
            lam = 4.0
        else:
            Y = torch.randn(n, 10, generator=g, device=dev) @ torch.randn(10, n, generator=g, device=dev)
            mask = torch.rand(n, n, generator=g, device=dev) < 0.1
            Y += mask * (10.0 * torch.randn(n, n, generator=g, device=dev))
            lam = 1.0
        yb = Y.t().contiguous().double().cpu().numpy().tobytes()  # column-major float64 bytes
        for rep in range(2):
            _solve.profile_enable(True)
            _solve.profile_reset()
            torch.cuda.synchronize()
            t0 = time.time()
            got = _solve.eval_prox(fb, lam, expr.data, {"var:X": yb})
            dt = time.time() - t0
            tags = _solve.profile_dump()
            _solve.profile_enable(False)
            part = sum(ms for t, (c, ms) in tags.items() if t.startswith("partial_svd"))
            full = sum(ms for t, (c, ms) in tags.items() if t.startswith("block_jacobi_svd") and t.endswith(":%dx%d" % (n, n)))
            print(json.dumps({"case": case, "n": n, "lam": lam, "rep": rep, "eval_prox_s_incl_host_copies": round(dt, 4),
                              "partial_svd_ms": round(part, 2), "full_jacobi_ms": round(full, 2)}), flush=True)
            del got


if __name__ == "__main__":
    main()
