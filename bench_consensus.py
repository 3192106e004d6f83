#!/usr/bin/env python3
"""Consensus-form lasso (SURVEY.md 8(e) mode E2) on N GPUs of one node: the EXAMPLES (rows of A)
are split over the ranks, every rank owns one term f_g(x_g) = |A_g x_g - b_g|^2, and
lam |z|_1 couples them through x_g - z = 0.  Per sweep the only data-path collective is the
z-averaging all-reduce of n floats; residual norms are a handful of doubles per check.  Not the
judged bench line (that is bench.py, the column-sharded mode with the single-GPU iterates); this
one measures the consensus mode BASELINE.json's north_star describes.

  python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 bench_consensus.py --gpus N
  (--comm host: N ranks sharing the visible GPUs through gloo, a rehearsal on a 1-GPU box)
"""
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
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--m", "--rows", dest="m", type=int, default=10000)
    ap.add_argument("--n", "--cols", dest="n", type=int, default=50000)
    ap.add_argument("--comm", default="rccl", choices=["rccl", "host"])
    args = ap.parse_args()
    # no rank environment: start the ranks as a child torch.distributed.run (before torch / HIP)
    from epsilon_amd import launch
    launch.self_launch_if_needed(__file__, args.gpus)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    import torch.distributed as dist
    import bench
    from epsilon_amd import _solve, ir, problems, wire
    from epsilon_amd import dist as edist

    assert torch.cuda.is_available(), "needs a HIP device"
    assert world == args.gpus
    if args.comm == "host":
        local_rank = local_rank % torch.cuda.device_count()
        os.environ["EPSILON_HIP_DEVICE"] = str(local_rank)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("gloo" if args.comm == "host" else "nccl",
                                **({} if args.comm == "host" else {"device_id": device}))
    _solve.set_option("dtype", "f32")
    if world > 1:
        edist.init_comm(rank, world, backend=args.comm)
        _solve.comm_warmup(1 << 16)

    m, n = args.m, args.n
    At, b, lam = bench.make_instance(m, n, device)  # the full instance, identical on every rank
    lo, hi = problems.consensus_row_range(m, rank, world)
    Ag = At[:, lo:hi].contiguous()                  # (n, m_g) == column-major m_g x n
    del At
    torch.cuda.empty_cache()
    data = {}
    c = ir.store_device(Ag.data_ptr(), hi - lo, n, "f32", data, "A_rows")
    x = ir.variable(n, 1, "var:x_local")
    z = ir.variable(n, 1, problems.CONSENSUS_Z)
    f = ir.prox(wire.ProxFunction.SUM_SQUARE,
                ir.add(ir.linear_map(ir.dense_matrix(constant=c, data=data), x),
                       ir.linear_map(ir.scalar(-1, hi - lo), ir.constant(b[lo:hi].double().cpu().numpy()))),
                alpha=1.0)
    h = ir.prox(wire.ProxFunction.NORM_1, z, alpha=lam)
    prob = ir.Problem([f, h], [ir.zero(ir.add(x, ir.linear_map(ir.scalar(-1, n), z)))])
    pb, blobs = prob.SerializeToString(), prob.expression_data()

    def declare():
        if world > 1:
            _solve.shard_keys(["var:x_local", "constraint:0"])
            _solve.shard_consensus_terms(True)

    def barrier():
        torch.cuda.synchronize()
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    declare()
    s = _solve.Solver(pb, wire.SolverParams(max_iterations=50000).SerializeToString(), blobs)
    barrier()
    t0 = time.time()
    s.init()
    barrier()
    t_init = time.time() - t0
    s.run(-1)
    barrier()
    t_total = time.time() - t0
    st = wire.SolverStatus.FromString(s.result()[0])
    s.close()

    s = _solve.Solver(pb, wire.SolverParams(max_iterations=10 ** 9,
                                            ignore_stopping_criteria=True).SerializeToString(), blobs)
    s.init()
    s.run(args.warmup)
    barrier()
    t0 = time.time()
    done = s.run(args.steps)
    barrier()
    dt = time.time() - t0
    assert done == args.steps
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=device if args.comm == "rccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    s.close()
    if rank == 0:
        print(json.dumps({
            "metric": "ADMM iters/sec, consensus-form dense Lasso 1e4x5e4", "value": args.steps / dt,
            "unit": "iter/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "strong",
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "consensus lasso m=%d n=%d, rows split x%d, z-averaging all-reduce of n floats"
                                   % (m, n, world)},
            "init_s": t_init, "time_to_eps_s": t_total, "iters_to_eps": st.num_iterations + 1,
            "state_at_eps": ["NOT_STARTED", "INITIALIZING", "RUNNING", "OPTIMAL",
                             "MAX_ITERATIONS_REACHED", "ERROR"][st.state],
            "residuals": {"r": st.residuals.r_norm, "s": st.residuals.s_norm,
                          "eps_pri": st.residuals.epsilon_primal, "eps_dual": st.residuals.epsilon_dual},
        }), flush=True)
    if dist.is_initialized():
        _solve.comm_shutdown()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
