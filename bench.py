#!/usr/bin/env python3
"""Headline benchmark: ADMM iterations/sec (+ wall-clock-to-eps) on dense Lasso
m=10^4 x n=5*10^4, fp32, through the C-ABI of libepsilon_hip.so.

    python bench.py --gpus N --steps K --warmup W

A "step" is one ADMM sweep of the compiled lasso (reference
src/epsilon/algorithms/prox_admm.cc:134-159) including the residual check the reference does
every `epoch_iterations` sweeps.  Inputs are synthetic (reference recipe
python/epopt/problems/lasso.py:8-15 + problem_util.py:9-42, x0 density 0.01 as
problems/benchmark.py:37), generated on the device and resident in HBM before any timing.
For N > 1 the same problem is column-sharded over the ranks (strong scaling) with one RCCL
all-reduce of m floats per sweep.  Rank 0 prints ONE JSON line.
"""

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# multi-process GPU work on this pool needs dmabuf IPC (RCCL fails with hipIpcGetMemHandle otherwise)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s measured copy ceiling)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=200)
    p.add_argument("--warmup", type=int, default=20)
    p.add_argument("--m", type=int, default=10000)
    p.add_argument("--n", type=int, default=50000)
    p.add_argument("--dtype", default="f32", choices=["f32", "f64"])
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-time-to-eps", action="store_true")
    p.add_argument("--no-profile", action="store_true",
                   help="diagnostic: no HIP-event kernel timers in the timed region (no roofline)")
    p.add_argument("--profile-all", action="store_true", help="time every tagged kernel")
    p.add_argument("--comm", default="rccl", choices=["rccl", "host"],
                   help="host: rehearsal of the N > 1 path on fewer GPUs than ranks (gloo + "
                        "host-callback collectives, ranks share devices) - not a judged configuration")
    p.add_argument("--force-sharded", action="store_true",
                   help="rehearsal on one GPU: run the sharded code path (RCCL communicator of "
                        "size 1) - not a judged configuration")
    return p.parse_args()


def make_instance(m, n, device, seed=0, rho=0.01, sigma=0.05, cols=None):
    """Synthetic lasso data on the device.  A is column-major m x n == a contiguous torch
    tensor of shape (n, m).  `cols` = (start, stop) keeps only that column slab resident
    (column sharding); b and lambda are those of the full problem either way."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    At = torch.randn(n, m, generator=g, device=device, dtype=torch.float32)
    At /= At.norm(dim=1, keepdim=True)  # unit l2 columns of A
    nnz = int(round(rho * n))
    perm = torch.randperm(n, generator=g, device=device)[:nnz]
    x0 = torch.zeros(n, device=device, dtype=torch.float32)
    x0[perm] = torch.randn(nnz, generator=g, device=device, dtype=torch.float32)
    b = At.t().matmul(x0) + sigma * torch.randn(m, generator=g, device=device, dtype=torch.float32)
    lam = 0.5 * float(At.matmul(b).abs().max())
    if cols is not None:
        At = At[cols[0]:cols[1]].contiguous()
    torch.cuda.synchronize()
    return At, b, lam


def build_problem(At, b, lam, key="A"):
    from epsilon_amd import ir, problems
    n, m = At.shape
    data = {}
    c = ir.store_device(At.data_ptr(), m, n, "f32", data, key)
    A_map = ir.dense_matrix(constant=c, data=data)
    prob = problems.lasso_ir(A_map, ir.constant(b.double().cpu().numpy()), lam, n)
    return prob


def cpu_baseline(At, b, lam, budget_s=20.0):
    """The oracle's plain-C restatement of the same sweep (oracle/lasso_sweep.c), fp64, one
    thread (the reference is single-threaded by design: tools/run_benchmarks.sh:15-17),
    timed on the host cores of this box on a bounded number of full-size sweeps."""
    from oracle import c_oracle
    n, m = At.shape
    A = np.asfortranarray(At.t().double().cpu().numpy())  # m x n column-major fp64
    bb = b.double().cpu().numpy()
    # The m x m cached operator: a sweep's cost does not depend on its values, and forming the
    # true inverse on one CPU thread at m = 10^4 takes tens of minutes (2.3 m^3 flop), so the
    # timing sample uses a synthetic symmetric operator of the right size.
    rng = np.random.RandomState(0)
    Minv = rng.randn(m, m) * (0.01 / np.sqrt(m))
    Minv = np.asfortranarray((Minv + Minv.T) / 2 + 0.2 * np.eye(m))
    st = c_oracle.LassoState(n)
    t0 = time.time()
    c_oracle.lasso_run(A, Minv, bb, lam, st, 1, abs_tol=0, rel_tol=0)
    t1 = time.time() - t0
    k = int(max(2, min(30, budget_s / max(t1, 1e-3))))
    t0 = time.time()
    done = c_oracle.lasso_run(A, Minv, bb, lam, st, k, abs_tol=0, rel_tol=0)
    dt = time.time() - t0
    return {
        "value": done / dt, "unit": "iter/s", "cores": 1, "kind": "port",
        "sample": "%d full-size sweeps (m=%d n=%d fp64, A = the GPU instance, synthetic m x m "
                  "operator) of oracle/lasso_sweep.c, gcc -O3, 1 thread" % (done, m, n),
        "ms_per_step": 1e3 * dt / done,
    }


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    import torch.distributed as dist
    from epsilon_amd import _solve, wire

    assert torch.cuda.is_available(), "bench.py needs a HIP device"
    if args.comm == "host":
        local_rank = local_rank % torch.cuda.device_count()
        os.environ["EPSILON_HIP_DEVICE"] = str(local_rank)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if args.comm == "host" and world > 1:
        dist.init_process_group("gloo")
    elif args.force_sharded and world == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29653")
        os.environ["EPSILON_HIP_FORCE_SHARDED"] = "1"
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
    elif world > 1:
        dist.init_process_group("nccl", device_id=device)
    assert world == args.gpus, "launch with torchrun --nproc-per-node %d" % args.gpus

    m, n = args.m, args.n
    _solve.set_option("dtype", args.dtype)
    sharded = world > 1 or args.force_sharded
    # process warm-up (untimed, local to the rank, before any communicator exists): one small
    # solve of the same structure, so that the code objects are loaded, the library's streams /
    # events exist and its buffer pool is primed before anything is timed (on a fresh box the
    # first Init otherwise carries ~20 ms of that)
    from epsilon_amd import problems
    wp, _ = problems.lasso(512, 2048, seed=1)
    _solve.solve(wp.SerializeToString(), [], wire.SolverParams(max_iterations=50).SerializeToString(),
                 wp.expression_data())
    if sharded:
        from epsilon_amd import dist as edist
        cols = edist.column_range(n, rank, world)
        comm_used = args.comm
        if args.comm == "rccl" and world > 1:
            # every rank must be able to load RCCL through the library's own binding, or none
            # uses it (a rank that fails alone would leave the others waiting in the init)
            try:
                _solve.comm_unique_id()
                ok = 1
            except Exception as e:
                print("rank %d: cannot bind RCCL (%s)" % (rank, e), file=sys.stderr, flush=True)
                ok = 0
            flag = torch.tensor([ok], dtype=torch.int32, device=device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0:
                args.comm = "host"
                comm_used = "host-staged fallback (torch.distributed all_reduce)"
        try:
            edist.init_comm(rank, world, backend=args.comm)
            # RCCL connects lazily on the first collective: do that (and a 100 MB one, the size
            # class of the Gram all-reduce) before anything is timed
            _solve.comm_warmup(1 << 16)
            if args.comm == "rccl":
                _solve.comm_warmup(25 * (1 << 20))
        except Exception as e:  # the library's own RCCL binding failed: say so, keep measuring
            if args.comm != "rccl":
                raise
            print("rank %d: RCCL communicator failed (%s); falling back to collectives staged "
                  "through torch.distributed" % (rank, e), file=sys.stderr, flush=True)
            try:
                _solve.comm_shutdown()
            except Exception:
                pass
            edist.init_comm(rank, world, backend="host")
            _solve.comm_warmup(1 << 16)
            comm_used = "host-staged fallback (torch.distributed all_reduce)"
    else:
        cols = None
    At, b, lam = make_instance(m, n, device, cols=cols)
    prob = build_problem(At, b, lam)
    pb, data = prob.SerializeToString(), prob.expression_data()

    def barrier():
        torch.cuda.synchronize()
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    def new_solver(params):
        s = _solve.Solver(pb, params.SerializeToString(), data)
        if sharded:
            edist.mark_sharded(s, prob)
        return s

    out = {}
    out["process_warmup"] = "one untimed lasso 512x2048 solve (50 sweeps) on every rank"
    # ---- wall-clock-to-eps at the reference defaults (benchmark.py:130-136: max_iterations 50000)
    if not args.no_time_to_eps:
        s = new_solver(wire.SolverParams(max_iterations=50000))
        # two HIP-event brackets inside Init: the Gram SYRK (the MFMA contraction of the
        # least-squares prox) and the explicit inverse
        _solve.set_option("profile_filter", "syrk:%d" % (m * m) + ",spd_inverse")
        _solve.profile_enable(True)
        _solve.profile_reset()
        barrier()
        t0 = time.time()
        s.init()
        barrier()
        t_init = time.time() - t0
        init_prof = _solve.profile_dump()
        _solve.profile_enable(False)
        s.run(-1)
        barrier()
        t_total = time.time() - t0
        st = wire.SolverStatus.FromString(s.result()[0])
        gram = init_prof.get("syrk:%dx%d" % (m * m, At.shape[0]))
        if gram and gram[0]:
            gram_ms = gram[1] / gram[0]
            flops = float(m) * (m + 1) * At.shape[0]  # lower triangle incl. diagonal, 2 flops / MAC
            out["init_breakdown"] = {
                "gram_syrk_ms": gram_ms,
                "gram": {"bound": "mfma", "kernel": "GemmMfmaF32PipeKernel<true,true> (SYRK A A^T, lower tiles)",
                         "achieved": flops / (gram_ms * 1e-3) / 1e12, "peak": 157.3, "unit": "TFLOP/s",
                         "frac": flops / (gram_ms * 1e-3) / 1e12 / 157.3,
                         "gemm_equivalent_TFLOPs": 2.0 * m * m * At.shape[0] / (gram_ms * 1e-3) / 1e12},
                "explicit_inverse_ms": (init_prof.get("spd_inverse:%d" % m, (0, 0.0))[1]),
            }
        out.update(init_s=t_init, time_to_eps_s=t_total,
                   iters_to_eps=st.num_iterations + 1,
                   state_at_eps=["NOT_STARTED", "INITIALIZING", "RUNNING", "OPTIMAL",
                                 "MAX_ITERATIONS_REACHED", "ERROR"][st.state])
        s.close()
        del s

    # ---- iterations/sec: exactly K timed sweeps after W warm-up sweeps
    params = wire.SolverParams(max_iterations=10 ** 9, ignore_stopping_criteria=True)
    s = new_solver(params)
    s.init()
    # live HIP-event timers only around the dominant kernel (each bracket costs ~1 us of stream
    # time; timing every small kernel would tax the sweep being measured by ~7 %)
    _solve.set_option("profile_filter",
                      "" if args.profile_all else "lasso_fused,gemv_n:%dx%d" % (m, At.shape[0]))
    _solve.profile_enable(not args.no_profile)
    s.run(args.warmup)
    _solve.profile_reset()
    barrier()
    t0 = time.time()
    done = s.run(args.steps)
    barrier()
    dt = time.time() - t0
    assert done == args.steps, (done, args.steps)
    prof = _solve.profile_dump()
    _solve.profile_enable(False)
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=device if args.comm == "rccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ms_per_step = 1e3 * dt / args.steps
    s.close()

    if rank != 0:
        _solve.comm_shutdown()
        dist.destroy_process_group()
        return

    sz = 4 if args.dtype == "f32" else 8
    n_loc = At.shape[0]
    # dominant kernel.  Fused path: ONE pass over the (local) data matrix does the work of
    # both mat-vecs of SURVEY 8(d) (K2 of this sweep + K1 of the next), so its algorithmic
    # bytes are 2*m*n*s while it moves m*n*s.  Unfused path: K1 = A v, m*n*s.
    fused_tag = "lasso_fused:%dx%d" % (m, n_loc)
    if fused_tag in prof:
        tag, kname = fused_tag, "LassoFusedStreamKernel<10> (K2 of sweep k + prox chain + K1 of sweep k+1)"
        alg_bytes, moved = 2 * m * n_loc * sz, m * n_loc * sz
    else:
        tag, kname = "gemv_n:%dx%d" % (m, n_loc), "GemvNKernel<float,4> (K1: y = A v)"
        alg_bytes = moved = m * n_loc * sz
    cnt, tot_ms = prof.get(tag, (0, 0.0))
    roofline = None
    if cnt:
        avg_ms = tot_ms / cnt
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tfile):
            try:
                traffic = json.load(open(tfile)).get(tag)
            except Exception:
                traffic = None
        roofline = {"bound": "hbm", "kernel": kname,
                    "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                    "avg_launch_ms": avg_ms, "launches": cnt,
                    "algorithmic_bytes_per_launch": alg_bytes,
                    "min_hbm_bytes_per_launch": moved,
                    "moved_GBs": moved / (avg_ms * 1e-3) / 1e9,
                    "frac_on_moved_bytes": moved / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
    # measured HBM ceilings on this device ("fraction of achievable" beside the
    # fraction of the vendor peak): the resident matrix read once by a plain streaming kernel,
    # and a device copy of it
    if roofline is not None and args.dtype == "f32":
        import ctypes
        L = _solve.lib()
        nbytes = At.numel() * 4
        ms = ctypes.c_double()
        ceil = {}
        for name, mode in (("read_nt", 0), ("read", 1), ("copy", 2)):
            best = 0.0
            for grid in (512, 1024, 2048, 4096):  # the best launch shape is the ceiling
                _solve._check(L.eps_bench_stream(ctypes.c_void_p(At.data_ptr()), ctypes.c_size_t(nbytes),
                                                 ctypes.c_int(mode), ctypes.c_int(grid), ctypes.c_int(10),
                                                 ctypes.byref(ms)))
                moved_b = nbytes * (2 if mode == 2 else 1)
                best = max(best, moved_b / (ms.value * 1e-3) / 1e9)
            ceil[name + "_GBs"] = best
        best_read = max(ceil["read_nt_GBs"], ceil["read_GBs"])
        roofline["achievable"] = dict(ceil, note="StreamReadKernel / StreamCopyKernel over the same %.2f GB, "
                                      "HIP events, best of four grid sizes, 10 launches each" % (nbytes / 1e9),
                                      frac_of_achievable_read=roofline["moved_GBs"] / best_read)
    sweep_bytes = (2 * m * n_loc + m * m) * sz
    out.update({
        "metric": "ADMM iters/sec, dense Lasso 1e4x5e4", "value": args.steps / dt,
        "unit": "iter/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": "lasso m=%d n=%d dense %s, PROX_ADMM, x0 density 0.01" % (m, n, args.dtype),
                   "parallelism": "1 GPU" if world == 1 else "column-sharded x%d, 1 all-reduce/sweep" % world,
                   **({"comm": comm_used} if sharded else {})},
        "roofline": roofline,
        "sweep": {"algorithmic_bytes_per_sweep_per_gpu": sweep_bytes,
                  "achieved_GBs": sweep_bytes / (ms_per_step * 1e-3) / 1e9,
                  "frac_of_peak": sweep_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS},
        "kernels": {k: {"launches": c, "avg_ms": t / c} for k, (c, t) in prof.items() if c},
    })
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(At, b, lam)
    print(json.dumps(out), flush=True)
    if dist.is_initialized():
        _solve.comm_shutdown()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
