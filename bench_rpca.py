#!/usr/bin/env python3
"""BASELINE.json configs[4] on ONE MI355X: robust PCA n x n (default 10^4), norm_nuclear(L) +
lam*norm_1(S) s.t. L + S = M (reference python/epopt/problems/robust_pca.py shape: rank-r plus
sparse corruption), solved to the reference's default tolerance through the C ABI.  The
nuclear-norm prox is the block one-sided Jacobi SVD of kernels_svd.hip, warm-started from the
previous sweep's right singular vectors.  One JSON line (a parity-test configuration, not the
judged bench line; the 8-GPU consensus form of this config is not built)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=10000)
    ap.add_argument("--rank", type=int, default=10)
    ap.add_argument("--max-iterations", type=int, default=300)
    ap.add_argument("--sweeps", type=int, default=0,
                    help="run exactly this many sweeps with the stopping rule off (per-sweep timing)")
    a = ap.parse_args()
    import numpy as np
    import torch
    from epsilon_amd import _solve, problems, wire
    _solve.set_option("dtype", "f32")
    t0 = time.time()
    prob, info = problems.robust_pca(a.n, r=a.rank, seed=0)
    pb, data = prob.SerializeToString(), prob.expression_data()
    t_build = time.time() - t0
    params = (wire.SolverParams(max_iterations=a.sweeps, ignore_stopping_criteria=True) if a.sweeps
              else wire.SolverParams(max_iterations=a.max_iterations))
    s = _solve.Solver(pb, params.SerializeToString(), data)
    torch.cuda.synchronize()
    t0 = time.time()
    s.init()
    per_iter = []
    done = 0
    while True:
        t1 = time.time()
        k = s.run(1)
        torch.cuda.synchronize()
        if k == 0:
            break
        per_iter.append(time.time() - t1)
        done += k
        if done % 10 == 0:
            print("sweep %d: %.2f s" % (done, per_iter[-1]), file=sys.stderr, flush=True)
    t_solve = time.time() - t0
    st, x = s.result()
    S = wire.SolverStatus.FromString(st)
    L = np.frombuffer(x["var:L"]).reshape(a.n, a.n, order="F")
    Sp = np.frombuffer(x["var:S"]).reshape(a.n, a.n, order="F")
    M = info["M"]
    out = {
        "workload": "robust PCA %dx%d, rank-%d + 10%% sparse corruption, lam=%g, fp32" % (a.n, a.n, a.rank, info["lam"]),
        "solve_s": t_solve, "sweeps": done, "state": ["NOT_STARTED", "INITIALIZING", "RUNNING", "OPTIMAL",
                                                      "MAX_ITERATIONS_REACHED", "ERROR"][S.state],
        "first_sweep_s": per_iter[0], "median_sweep_s": float(np.median(per_iter)),
        "sweep_s": [round(t, 3) for t in per_iter[:12]],
        "residuals": {"r": S.residuals.r_norm, "s": S.residuals.s_norm, "eps_pri": S.residuals.epsilon_primal,
                      "eps_dual": S.residuals.epsilon_dual},
        "constraint_rel_err": float(np.linalg.norm(L + Sp - M) / np.linalg.norm(M)),
        "nnz_fraction_S": float(np.mean(Sp != 0)), "n_gpus": 1, "data": "synthetic", "ir_build_s": t_build,
    }
    s.close()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
