#!/usr/bin/env python3
"""Headline benchmark: ADMM iterations/sec (+ wall-clock-to-eps) on dense Lasso
m=10^4 x n=5*10^4, fp32, through the C-ABI of libepsilon_hip.so.

    python bench.py --gpus N --steps K --warmup W

A "step" is one ADMM sweep of the compiled lasso (reference
src/epsilon/algorithms/prox_admm.cc:134-159) including the residual check the reference does
every `epoch_iterations` sweeps.  Inputs are synthetic (reference recipe
python/epopt/problems/lasso.py:8-15 + problem_util.py:9-42, x0 density 0.01 as
problems/benchmark.py:37), generated on the device and resident in HBM before any timing.
For N > 1 the same problem is column-sharded over the ranks (strong scaling), one m-float exchange
per sweep.  Launched as one rank per GPU by torch.distributed.run - or, when called without a
rank environment, by bench.py itself, which then starts that command as a child process.  Rank 0
prints ONE JSON line.
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
    # (--rows / --cols: the spellings to use under torch.distributed.run, whose own parser trips
    # over "--m" as an ambiguous abbreviation of its options)
    p.add_argument("--m", "--rows", dest="m", type=int, default=10000)
    p.add_argument("--n", "--cols", dest="n", type=int, default=50000)
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
    p.add_argument("--peer", default="auto", choices=["auto", "off"],
                   help="auto: per-sweep exchanges on the one-shot peer-write window when every rank "
                        "can set it up (falls back to RCCL collectives, agreed across ranks); off: RCCL")
    p.add_argument("--rehearse-ranks", type=int, default=0,
                   help="with --force-sharded: this process plays ONE rank of that many (slab of the "
                        "inverse apply, window slots and exchange traffic of that rank count; pass "
                        "--n = the rank's column slab).  Timing rehearsal only: the iterates are not "
                        "a solve's - not a judged configuration")
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


def cpu_info():
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    # a container's CPU share (cgroup quota) is what the process can really use
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    if quota is not None:
        usable = max(1, min(usable, int(quota + 0.5)))
    return model, os.cpu_count() or 1, usable


def cpu_baseline(At, b, lam, iters_to_eps=None, budget_s=10.0):
    """The reference's CPU path timed on this box's host cores, on a bounded sample of the same
    workload.  Returns (cpu_baseline, cpu_baseline_all_cores):

    * cpu_baseline, kind "reference": the unrolled lasso sweep with its three mat-vecs through the
      dgemv_ of the reference's OWN tree (third_party/eigen/blas, compiled from there into
      oracle/_ref/libref.so by oracle/Makefile; oracle/ref_driver.cc: ref_lasso_sweeps), fp64, ONE
      thread - the reference is single-threaded by design (tools/run_benchmarks.sh:15-17).  Init
      (Gram + explicit inverse) through the same tree's dgemm_ and Eigen::LDLT
      (linear_map_multiply.cc:14-37, dense_matrix_impl.cc:21-30) at 1/4 scale, extrapolated by the
      m^2 n / m^3 laws - an estimate, stated as such.
    * cpu_baseline_all_cores, kind "port": the plain-C restatement (oracle/lasso_sweep.c) with
      OpenMP over the mat-vecs on all usable cores - what the box could do, not what the reference does.
    Falls back to the port at one thread (kind "port") where oracle/_ref was never built."""
    from oracle import c_oracle
    n, m = At.shape
    A = np.asfortranarray(At.t().double().cpu().numpy())  # m x n column-major fp64
    bb = b.double().cpu().numpy()
    # The m x m cached operator: a sweep's cost does not depend on its values, and forming the
    # true inverse on one CPU thread at m = 10^4 takes minutes (2.3 m^3 flop), so the
    # timing sample uses a synthetic symmetric operator of the right size.
    rng = np.random.RandomState(0)
    Minv = rng.randn(m, m) * (0.01 / np.sqrt(m))
    Minv = np.asfortranarray((Minv + Minv.T) / 2 + 0.2 * np.eye(m))
    model, logical, usable = cpu_info()
    # all cores the process may use; capped at 64 (the mat-vecs are memory-bound long before)
    threads_all = max(1, min(usable, c_oracle.max_threads(), 64))
    ref = None
    try:
        from oracle import ref_lib
        if ref_lib.available():
            ref_lib.dgemv(A[:, :8], np.zeros(8))  # load the library
            ref = ref_lib
    except Exception:
        ref = None

    def time_sweeps(run, budget):
        st = c_oracle.LassoState(n)
        t0 = time.time()
        run(st, 1)
        t1 = time.time() - t0
        k = int(max(2, min(40, budget / max(t1, 1e-3))))
        t0 = time.time()
        run(st, k)
        return k, time.time() - t0

    def port_run(threads):
        def run(st, k):
            c_oracle.set_threads(threads)
            c_oracle.lasso_run(A, Minv, bb, lam, st, k, abs_tol=0, rel_tol=0)
        return run

    def entry(done, dt, threads, kind, sample, init_s, init_note):
        o = {"value": done / dt, "unit": "iter/s", "cores": threads, "kind": kind, "sample": sample,
             "ms_per_step": 1e3 * dt / done, "cpu_model": model, "host_logical_cpus": logical,
             "host_usable_cpus": usable, "init_s_estimate": init_s, "init_estimate_note": init_note}
        if iters_to_eps:
            o["time_to_eps_s_estimate"] = init_s + iters_to_eps * dt / done
            o["time_to_eps_note"] = ("Init estimate + %d sweeps (the GPU run's count to OPTIMAL) at the "
                                     "measured CPU rate" % iters_to_eps)
        return o

    # ---- one thread: the reference tree's own BLAS / LDLT where built
    if ref is not None:
        done, dt = time_sweeps(lambda st, k: ref.lasso_sweeps(A, Minv, bb, lam, st, k), budget_s)
        # Init on a 1/4-scale instance (2500 x 12500 at the default size): ~1.6e11 flop of dgemm_ +
        # the LDLT inverse, ~10-20 s on one thread
        ms, ns = max(64, m // 4), max(64, n // 4)
        As = np.asfortranarray(A[:ms, :ns])
        t0 = time.time()
        G = ref.dgemm(As, As, tb=True)
        t_gram = time.time() - t0
        t0 = time.time()
        ref.ldlt_inverse(np.eye(ms) + 2 * G)
        t_inv = time.time() - t0
        inv_full_s = t_inv * (float(m) / ms) ** 3
        init_s = t_gram * (float(m) / ms) ** 2 * (float(n) / ns) + inv_full_s
        one = entry(done, dt, 1, "reference",
                    "%d full-size sweeps (m=%d n=%d fp64, A = the GPU instance, synthetic m x m operator): "
                    "the unrolled sweep of prox_admm.cc:131-169 with all three mat-vecs through dgemv_ of the "
                    "reference's own tree (oracle/_ref/libref.so = third_party/eigen/blas, g++ -O3 -DNDEBUG), 1 thread"
                    % (done, m, n), init_s,
                    "Gram through the same tree's dgemm_ (%.2f s) + Eigen::LDLT solve(I) (%.2f s) timed at %d x %d "
                    "and scaled by m^2 n resp. m^3 to %d x %d" % (t_gram, t_inv, ms, ns, m, n))
    else:
        done, dt = time_sweeps(port_run(1), budget_s)
        ms, ns = max(64, m // 8), max(64, n // 8)
        As = np.asfortranarray(A[:ms, :ns])
        c_oracle.set_threads(1)
        t0 = time.time()
        G = c_oracle.gram(As)
        t_gram = time.time() - t0
        import scipy.linalg
        t0 = time.time()
        c, low = scipy.linalg.cho_factor(np.eye(ms) + 2 * G)
        scipy.linalg.cho_solve((c, low), np.eye(ms))
        t_inv = time.time() - t0
        inv_full_s = t_inv * (float(m) / ms) ** 3
        init_s = t_gram * (float(m) / ms) ** 2 * (float(n) / ns) + inv_full_s
        one = entry(done, dt, 1, "port",
                    "%d full-size sweeps (m=%d n=%d fp64) of oracle/lasso_sweep.c, gcc -O3, 1 thread "
                    "(oracle/_ref not built on this box)" % (done, m, n), init_s,
                    "plain-C Gram + LAPACK Cholesky inverse (scipy) at %d x %d, scaled" % (ms, ns))
    # ---- all usable cores: the OpenMP port (what the box could do; not the reference's configuration)
    done_a, dt_a = time_sweeps(port_run(threads_all), budget_s / 2)
    ms, ns = max(64, m // 8), max(64, n // 8)
    As = np.asfortranarray(A[:ms, :ns])
    c_oracle.set_threads(threads_all)
    t0 = time.time()
    c_oracle.gram(As)
    t_gram_a = time.time() - t0
    c_oracle.set_threads(1)
    # the inverse is not threaded in the port: the one-thread figure of the first leg, scaled
    allc = entry(done_a, dt_a, threads_all, "port",
                 "%d full-size sweeps (m=%d n=%d fp64) of oracle/lasso_sweep.c, gcc -O3 -fopenmp, %d threads"
                 % (done_a, m, n, threads_all),
                 t_gram_a * (float(m) / ms) ** 2 * (float(n) / ns) + inv_full_s,
                 "plain-C blocked Gram on %d threads at %d x %d scaled by m^2 n + the one-thread inverse of the first leg"
                 % (threads_all, ms, ns))
    return one, allc


def main():
    args = parse()
    # `python bench.py --gpus N` without a rank environment: start the N ranks as a child
    # torch.distributed.run (decided before torch / HIP are touched; exits with the child's code)
    from epsilon_amd import launch
    launch.self_launch_if_needed(__file__, args.gpus)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    import torch.distributed as dist
    from epsilon_amd import _solve, wire

    assert torch.cuda.is_available(), "bench.py needs a HIP device"
    if args.comm == "rccl" and world > torch.cuda.device_count():
        sys.exit("bench.py: %d ranks but %d visible GPU(s): one rank per GPU over RCCL needs %d devices "
                 "(--comm host rehearses the N > 1 path with ranks sharing the devices)"
                 % (world, torch.cuda.device_count(), world))
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
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d (launch %d ranks, or none and let bench.py "
                 "start them)" % (args.gpus, world, args.gpus))

    m, n = args.m, args.n
    _solve.set_option("dtype", args.dtype)
    sharded = world > 1 or args.force_sharded
    # process warm-up (untimed, local to the rank, before any communicator exists): one small
    # solve of the same structure, so that the code objects are loaded, the library's streams /
    # events exist and its buffer pool is primed before anything is timed (on a fresh box the
    # first Init otherwise carries ~20 ms of that)
    # (2048 x 8192: the smallest shape whose setup takes the SAME kernels as the timed one - the
    # split-f16 product and its conversion passes, the blocked Cholesky steps, the triangular
    # products - a 512 x 2048 solve loaded the small-matrix variants instead and left ~3 ms of
    # lazy code-object loading inside the timed Init)
    from epsilon_amd import problems
    wp, _ = problems.lasso(2048, 8192, seed=1)
    _solve.solve(wp.SerializeToString(), [], wire.SolverParams(max_iterations=50).SerializeToString(),
                 wp.expression_data())
    del wp
    peer_on, peer_why = False, "single GPU"
    if sharded:
        from epsilon_amd import dist as edist
        cols = edist.column_range(n, rank, world)
        # A failure of the library's RCCL binding is FATAL here (no host-staged fallback: a line
        # measured over staged collectives would not be the judged configuration); --comm host
        # is an explicit rehearsal mode and is labelled as such in the output.
        peer_on, peer_why = edist.init_comm(rank, world, backend=args.comm, peer=False)
        # RCCL connects lazily on the first collective: do that (and a 100 MB one, the size
        # class of the Gram all-reduce) before anything is timed
        _solve.comm_warmup(1 << 16)
        if args.comm == "rccl":
            _solve.comm_warmup(25 * (1 << 20))
        if args.peer == "auto":
            peer_on, peer_why = _solve.comm_enable_peer(max(m * (2 if args.dtype == "f64" else 1), 16384), args.rehearse_ranks)
        else:
            peer_on, peer_why = False, "--peer off"
        comm_used = ("one-shot xGMI peer-write window inside the sweep kernels + hipGraph replay; "
                     "RCCL for the Gram all-reduce and the residual scalars" if peer_on
                     else "RCCL all-reduce + all-gather per sweep (peer window: %s)" % peer_why)
        if args.comm == "host":
            comm_used = "REHEARSAL over host-staged collectives (--comm host); " + comm_used
        if args.rehearse_ranks > 1:
            comm_used = "REHEARSAL of one rank of %d on one GPU (timing only); " % args.rehearse_ranks + comm_used
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
    out["process_warmup"] = "one untimed lasso 2048x8192 solve (50 sweeps) on every rank: loads the code objects of the kernels the timed sizes use"
    # ---- N > 1: the peer-window sweep has to reproduce the RCCL-path sweep on THIS machine before
    # it is measured (31 sweeps of the real problem each way, residuals and iterates compared,
    # verdict agreed across the ranks); otherwise the window is dropped on every rank and the
    # line below is measured - and labelled - on the RCCL path.
    if sharded and peer_on and world > 1:
        def short_solve():
            sv = new_solver(wire.SolverParams(max_iterations=31, abs_tol=0.0, rel_tol=0.0))
            sv.init()
            sv.run(-1)
            st_b, xs = sv.result()
            sv.close()
            st = wire.SolverStatus.FromString(st_b)
            vec = np.concatenate([np.frombuffer(xs[k]) for k in sorted(xs)])
            return np.array([st.residuals.r_norm, st.residuals.s_norm]), vec
        ok, why_not = 1, ""
        try:
            r_peer, x_peer = short_solve()
        except Exception as e:  # a timed-out exchange raises on every rank at the same check
            ok, why_not = 0, "peer-window solve failed: %s" % e
            r_peer = x_peer = None
        _solve.comm_disable_peer()
        r_rccl, x_rccl = short_solve()
        if ok:
            scale = max(1e-30, float(np.abs(x_rccl).max()))
            if not (np.allclose(r_peer, r_rccl, rtol=1e-3, atol=1e-9) and
                    float(np.abs(x_peer - x_rccl).max()) <= 1e-3 * scale):
                ok, why_not = 0, "peer-window iterates differ from the RCCL path: residuals %s vs %s" % (r_peer, r_rccl)
        flag = torch.tensor([ok], dtype=torch.int32, device=device if args.comm == "rccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 1:
            peer_on, _ = _solve.comm_enable_peer(max(m * (2 if args.dtype == "f64" else 1), 16384), 0)
        else:
            peer_on = False
        if not peer_on:
            comm_used = ("RCCL all-reduce + all-gather per sweep (peer window dropped after the validation "
                         "solve: %s)" % (why_not or "a peer rank reported a mismatch"))
        out["peer_window_validation"] = ("31 sweeps each way on this problem: residuals within 1e-3, iterates "
                                         "within 1e-3 of the largest entry" if peer_on else "failed - " + (why_not or "on a peer rank"))
    # ---- wall-clock-to-eps at the reference defaults (benchmark.py:130-136: max_iterations 50000)
    if not args.no_time_to_eps and args.rehearse_ranks <= 1:
        s = new_solver(wire.SolverParams(max_iterations=50000))
        # two HIP-event brackets inside Init: the Gram SYRK (the MFMA contraction of the
        # least-squares prox) and the explicit inverse
        _solve.set_option("profile_filter", "syrk:%d" % (m * m) + ",syrk_f16split:%d" % (m * m) + ",spd_inverse")
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
        gram16 = init_prof.get("syrk_f16split:%dx%d" % (m * m, At.shape[0]))
        if gram16 and gram16[0]:
            gram_ms = gram16[1] / gram16[0]
            k_loc = At.shape[0]
            mfma_flops = 3.0 * float(m) * (m + 1) * k_loc  # three f16 products per f32 one, lower tiles
            out["init_breakdown"] = {
                "gram_syrk_ms": gram_ms,
                "gram": {"bound": "mfma", "kernel": "SyrkSplitF16Kernel (A A^T lower tiles, f16 MFMA on two-term split "
                                                    "operands, f32 accumulate) + AbsMax + SplitConvert",
                         "achieved": mfma_flops / (gram_ms * 1e-3) / 1e12, "peak": 2500.0, "unit": "TFLOP/s",
                         "frac": mfma_flops / (gram_ms * 1e-3) / 1e12 / 2500.0,
                         "f32_equivalent_TFLOPs": float(m) * (m + 1) * k_loc / (gram_ms * 1e-3) / 1e12,
                         "note": "achieved counts the f16 MFMA flops actually issued (3 per f32 multiply-add of the "
                                 "lower triangle) over the whole bracket, conversion passes included; peak = dense "
                                 "f16 MFMA; the exact-f32 MFMA kernel (EPSILON_HIP_GRAM_F16SPLIT=0) does the same "
                                 "product at 0.73 of ITS 157.3 TFLOP/s peak in 43.8 ms"},
                "explicit_inverse_ms": (init_prof.get("spd_inverse:%d" % m, (0, 0.0))[1]),
            }
        elif gram and gram[0]:
            gram_ms = gram[1] / gram[0]
            flops = float(m) * (m + 1) * At.shape[0]  # lower triangle incl. diagonal, 2 flops / MAC
            out["init_breakdown"] = {
                "gram_syrk_ms": gram_ms,
                "gram": {"bound": "mfma",
                         "kernel": ("GemmMfmaF32PipeKernel<true,true> (SYRK A A^T, lower tiles)" if args.dtype == "f32"
                                    else "GemmMfmaF64PipeKernel<true,true> (SYRK A A^T, lower tiles; v_mfma_f64_16x16x4_f64)"),
                         "achieved": flops / (gram_ms * 1e-3) / 1e12,
                         "peak": 157.3 if args.dtype == "f32" else 78.6, "unit": "TFLOP/s",
                         "frac": flops / (gram_ms * 1e-3) / 1e12 / (157.3 if args.dtype == "f32" else 78.6),
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
    # With the peer window the sweeps between residual checks are replayed from hipGraphs, which
    # carry no events: the timed region then runs un-instrumented and the kernel durations come
    # from a second, eager region of the same K sweeps.
    graph_mode = sharded and peer_on and os.environ.get("EPSILON_HIP_GRAPH", "") != "0"
    _solve.profile_enable(not args.no_profile and not graph_mode)
    s.run(args.warmup)
    _solve.profile_reset()
    barrier()
    t0 = time.time()
    done = s.run(args.steps)
    barrier()
    dt = time.time() - t0
    assert done == args.steps, (done, args.steps)
    if graph_mode and not args.no_profile:
        _solve.set_option("profile_filter", "lasso_fused,peer_,symv")
        _solve.profile_enable(True)
        _solve.profile_reset()
        barrier()
        t1 = time.time()
        s.run(args.steps)
        barrier()
        out["eager_instrumented_ms_per_step"] = 1e3 * (time.time() - t1) / args.steps
        out["kernel_timing"] = ("second region of %d eager sweeps with HIP events around every sweep "
                                "kernel (the timed region replays hipGraphs)" % args.steps)
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
        # `achieved` / `frac` are on the bytes the kernel MUST move (the matrix once, m*n*s): a
        # roofline fraction cannot exceed 1.  The fused pass does the work of both mat-vecs of
        # SURVEY 8(d) (2*m*n*s algorithmic bytes in the two-pass formulation); that equivalence
        # is reported separately and is not a fraction of anything.
        roofline = {"bound": "hbm", "kernel": kname,
                    "achieved": moved / (avg_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": moved / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "traffic": traffic,
                    "traffic_source": "profiles/traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of "
                                      "this command, corrected per MI355X_MICROARCH.md; not re-measured by this run)",
                    "avg_launch_ms": avg_ms, "launches": cnt,
                    "min_hbm_bytes_per_launch": moved,
                    "two_pass_equivalent": {"algorithmic_bytes_per_launch": alg_bytes,
                                            "equivalent_GBs": achieved,
                                            "note": "what two separate mat-vec passes would have to stream "
                                                    "to do this launch's work; > peak is possible, it is not a roofline fraction"}}
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
                                      frac_of_achievable_read=roofline["achieved"] / best_read)
    sweep_bytes = (2 * m * n_loc + m * m) * sz
    # bytes a sweep of THIS build has to move per GPU: the matrix once, the lower triangle of the
    # cached inverse (or its row slab), the per-workgroup partials written and read once
    G_eff = args.rehearse_ranks if args.rehearse_ranks > 1 else world
    inv_elems = m * (m + 1) // 2 if G_eff < 3 else m * (-(-m // G_eff))
    moved_sweep = (m * n_loc + inv_elems) * sz
    out.update({
        "metric": "ADMM iters/sec, dense Lasso 1e4x5e4", "value": args.steps / dt,
        "unit": "iter/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": "lasso m=%d n=%d dense %s, PROX_ADMM, x0 density 0.01" % (m, n, args.dtype),
                   "parallelism": "1 GPU" if world == 1 else "column-sharded x%d, 1 all-reduce/sweep" % world,
                   **({"comm": comm_used} if sharded else {})},
        "roofline": roofline,
        "sweep": {"moved_bytes_per_sweep_per_gpu": moved_sweep,
                  "moved_GBs": moved_sweep / (ms_per_step * 1e-3) / 1e9,
                  "frac_on_moved_bytes": moved_sweep / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                  "two_pass_algorithmic_bytes_per_sweep_per_gpu": sweep_bytes,
                  "two_pass_equivalent_GBs": sweep_bytes / (ms_per_step * 1e-3) / 1e9},
        "kernels": {k: {"launches": c, "avg_ms": t / c} for k, (c, t) in prof.items() if c},
    })
    if world == 1 and not args.no_cpu_baseline and not args.force_sharded:
        one, allc = cpu_baseline(At, b, lam, iters_to_eps=out.get("iters_to_eps"))
        out["cpu_baseline"] = one
        out["cpu_baseline_all_cores"] = allc
    print(json.dumps(out), flush=True)
    if dist.is_initialized():
        _solve.comm_shutdown()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
