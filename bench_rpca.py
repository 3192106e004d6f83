#!/usr/bin/env python3
"""BASELINE.json configs[4] on ONE MI355X: robust PCA n x n (default 10^4), norm_nuclear(L) +
lam*norm_1(S) s.t. L + S = M (reference python/epopt/problems/robust_pca.py shape: rank-r plus
sparse corruption), solved to the reference's default tolerance through the C ABI.  The
nuclear-norm prox is the block one-sided Jacobi SVD of kernels_svd.hip, warm-started from the
previous sweep's right singular vectors.  With --gpus N (torchrun, one rank per GPU) the matrix is
split by ROWS over the ranks: L, S and the constraint row are sharded, the nuclear-norm prox runs
the row-sharded block Jacobi (one all-reduce of the panel Grams per rotation step, V and the
singular values replicated), the l1 prox and the updates are local.  One JSON line (a parity-test
configuration, not the judged bench line).

  python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 bench_rpca.py --gpus 8
  (--comm host: ranks share the visible GPUs through gloo - a rehearsal on a 1-GPU box)"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", "--n", "--cols", dest="n", type=int, default=10000,
                    help="matrix size (use --size under torchrun: its parser claims the prefix --n)")
    ap.add_argument("--rank", type=int, default=10)
    ap.add_argument("--max-iterations", type=int, default=300)
    ap.add_argument("--sweeps", type=int, default=0,
                    help="run exactly this many sweeps with the stopping rule off (per-sweep timing)")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--comm", default="rccl", choices=["rccl", "host"])
    a = ap.parse_args()
    # no rank environment: start the ranks as a child torch.distributed.run (before torch / HIP)
    from epsilon_amd import launch
    launch.self_launch_if_needed(__file__, a.gpus)
    import numpy as np
    import torch
    import torch.distributed as dist
    from epsilon_amd import _solve, problems, wire
    from epsilon_amd import dist as edist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert torch.cuda.is_available() and world == a.gpus
    if a.comm == "host":
        local_rank = local_rank % torch.cuda.device_count()
        os.environ["EPSILON_HIP_DEVICE"] = str(local_rank)
    torch.cuda.set_device(local_rank)
    if world > 1:
        if a.comm == "host":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    _solve.set_option("dtype", "f32")
    if world > 1:
        edist.init_comm(rank, world, backend=a.comm)
        _solve.comm_warmup(1 << 16)
    t0 = time.time()
    M = problems.robust_pca_data(a.n, r=a.rank, seed=0)  # identical on every rank
    lo, hi = edist.column_range(a.n, rank, world, align=1)
    info = dict(M=M[lo:hi], lam=0.1)
    prob = problems.robust_pca_ir(np.ascontiguousarray(M[lo:hi]), info["lam"])
    del M
    pb, data = prob.SerializeToString(), prob.expression_data()
    if world > 1:
        _solve.shard_keys(["var:L", "var:S", "constraint:0"])
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
    L = np.frombuffer(x["var:L"]).reshape(hi - lo, a.n, order="F")
    Sp = np.frombuffer(x["var:S"]).reshape(hi - lo, a.n, order="F")
    M = info["M"]  # this rank's rows
    out = {
        "workload": "robust PCA %dx%d, rank-%d + 10%% sparse corruption, lam=%g, fp32, rows sharded x%d"
                    % (a.n, a.n, a.rank, info["lam"], world),
        "solve_s": t_solve, "sweeps": done, "state": ["NOT_STARTED", "INITIALIZING", "RUNNING", "OPTIMAL",
                                                      "MAX_ITERATIONS_REACHED", "ERROR"][S.state],
        "first_sweep_s": per_iter[0], "median_sweep_s": float(np.median(per_iter)),
        "sweep_s": [round(t, 3) for t in per_iter[:12]],
        "residuals": {"r": S.residuals.r_norm, "s": S.residuals.s_norm, "eps_pri": S.residuals.epsilon_primal,
                      "eps_dual": S.residuals.epsilon_dual},
        "constraint_rel_err": float(np.linalg.norm(L + Sp - M) / np.linalg.norm(M)),
        "nnz_fraction_S": float(np.mean(Sp != 0)), "n_gpus": world, "data": "synthetic", "ir_build_s": t_build,
        "note": "constraint_rel_err / nnz_fraction_S are those of rank 0's rows",
    }
    s.close()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        _solve.comm_shutdown()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
