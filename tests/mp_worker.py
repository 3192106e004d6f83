"""Worker for the multi-process tests (launched once per rank with RANK / WORLD_SIZE /
MASTER_ADDR / MASTER_PORT in the environment).

mode "oracle": numpy emulation of the column-sharded (E1) lasso sweep with gloo collectives -
               checks the partitioning and where the collectives sit (runs on CPU).
mode "oracle_consensus": numpy emulation of the consensus-form (E2) lasso, rows of A split over
               the ranks, z-averaging and residual sums over gloo (runs on CPU).
mode "hip_consensus": the HIP solver on the per-rank consensus problem (host-callback comm).
mode "oracle_rpca": numpy emulation of the row-sharded robust PCA sweep: singular value
               thresholding through the all-reduced Gram matrix Y^T Y (the reference's own route,
               ortho_invariant.cc:36-50), norms through all-reduced partial sums (runs on CPU).
mode "hip_rpca": robust PCA with the matrix split by ROWS over the ranks (row-sharded SVD in the
               nuclear-norm prox; m is the matrix size, n is ignored).
mode "hip"   : the real HIP solver, one rank per process, collectives through the host-callback
               backend over gloo (ranks may share one GPU).
Each rank writes its slice of the result to <out>/rank<r>.npz.
"""

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    mode, out_dir, m, n, seed, max_iter = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
    import torch
    import torch.distributed as dist
    from epsilon_amd import dist as edist
    from epsilon_amd import ir, problems, wire

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    A, b = problems.regression_data(m, n, seed=seed)
    lam = 0.5 * np.abs(A.T.dot(b)).max()
    lo, hi = edist.column_range(n, rank, world, align=1)
    Ag = np.asfortranarray(A[:, lo:hi])
    ng = hi - lo

    def allreduce(x):
        t = torch.from_numpy(np.ascontiguousarray(x))
        dist.all_reduce(t)
        return t.numpy()

    if mode in ("oracle_consensus", "hip_consensus"):
        lam = 0.3 * np.abs(A.T.dot(b)).max()
        lo, hi = problems.consensus_row_range(m, rank, world)
        Ag, bg = A[lo:hi], b[lo:hi]
        if mode == "hip_consensus":
            from epsilon_amd import _solve
            _solve.set_option("dtype", os.environ.get("EPS_TEST_DTYPE", "f64"))
            edist.init_comm(rank, world, backend="host")
            prob = problems.consensus_lasso_local(Ag, bg, lam)
            _solve.shard_keys(["var:x_local", "constraint:0"])
            _solve.shard_consensus_terms(True)
            params = wire.SolverParams(max_iterations=max_iter)
            st, x = _solve.solve(prob.SerializeToString(), [], params.SerializeToString(),
                                 prob.expression_data())
            S = wire.SolverStatus.FromString(st)
            np.savez(os.path.join(out_dir, "rank%d.npz" % rank),
                     x0=np.frombuffer(x["var:x_local"]), x1=np.frombuffer(x[problems.CONSENSUS_Z]),
                     lo=lo, hi=hi,
                     status=np.array([S.num_iterations, S.residuals.r_norm, S.residuals.s_norm,
                                      S.residuals.epsilon_primal, S.residuals.epsilon_dual]),
                     state=S.state)
            _solve.comm_shutdown()
        else:
            # per-rank sweep of [f_g, h] with x_g - z = 0 (prox_admm.cc:135-147 unrolled); the only
            # data-path collective is the sum over ranks inside the z update
            G = world
            Minv = np.linalg.inv(np.eye(hi - lo) + 2 * Ag.dot(Ag.T))
            xg = np.zeros(n); z = np.zeros(n); u = np.zeros(n); y0 = np.zeros(n); y1 = np.zeros(n)
            it, status = 0, None
            while it < max_iter:
                y1_prev = y1.copy()
                u = u - y0 - y1
                u = u + y0
                w = Minv.dot(bg - Ag.dot(u))
                xg = u + 2 * Ag.T.dot(w)
                y0 = xg.copy()
                u = u - y0
                u = u + y1
                v = allreduce(-u) / G                    # z-averaging: n doubles
                z = np.sign(v) * np.maximum(np.abs(v) - lam / G, 0)
                y1 = -z
                u = u - y1
                if it % 10 == 0:
                    s = allreduce(np.array([np.sum((xg - z) ** 2), np.sum((y1 - y1_prev) ** 2),
                                            np.sum(z ** 2), np.sum(u ** 2)]))
                    usum = allreduce(u.copy())
                    onehot = np.zeros(G); onehot[rank] = np.sqrt(np.sum(xg ** 2))
                    xmax = allreduce(onehot).max()       # max over the per-rank terms
                    r, sn = np.sqrt(s[0]), np.sqrt(s[1])
                    ep = 1e-4 * np.sqrt(G * n) + 1e-2 * max(xmax, np.sqrt(s[2]))
                    ed = 1e-4 * np.sqrt((G + 1) * n) + 1e-2 * np.sqrt(s[3] + np.sum(usum ** 2))
                    status = (it, r, sn, ep, ed)
                    if r <= ep and sn <= ed:
                        break
                it += 1
            np.savez(os.path.join(out_dir, "rank%d.npz" % rank), x0=xg, x1=z, lo=lo, hi=hi,
                     status=np.array(status))
        dist.barrier()
        dist.destroy_process_group()
        return
    if mode == "oracle_rpca":
        nn = m
        M = problems.robust_pca_data(nn, r=3, density=0.1, seed=seed)
        lo, hi = edist.column_range(nn, rank, world, align=1)
        Mg = M[lo:hi]
        lam_s = 0.1
        b = -Mg
        shape = Mg.shape
        u = np.zeros(shape); y0 = np.zeros(shape); y1 = np.zeros(shape)
        Lg = np.zeros(shape); Sg = np.zeros(shape)

        def svt_rows(Y):  # prox of ||.||_* on the row-sharded matrix: eig of the summed Gram
            G = allreduce(Y.T.dot(Y) + (1e-15 / world) * np.eye(nn))
            w, V = np.linalg.eigh(G)
            sig = np.sqrt(np.maximum(w, 0))
            shr = np.maximum(sig - 1.0, 0)
            scale = np.where(sig > 0, shr / np.where(sig > 0, sig, 1), 0)
            return (Y.dot(V) * scale).dot(V.T)

        def nsq(*arrs):
            return allreduce(np.array([np.sum(a ** 2) for a in arrs]))

        it, status = 0, None
        while it < max_iter:
            y1_prev = y1.copy()
            u = u - b - y0 - y1
            u = u + y0
            Lg = svt_rows(u)
            y0 = Lg.copy()
            u = u - y0
            u = u + y1
            Sg = np.sign(u) * np.maximum(np.abs(u) - lam_s, 0)
            y1 = Sg.copy()
            u = u - y1
            if it % 10 == 0:
                q = nsq(y0 + y1 + b, y1 - y1_prev, b, y0, y1, u)
                r, sn = np.sqrt(q[0]), np.sqrt(q[1])
                ep = 1e-4 * np.sqrt(nn * nn) + 1e-2 * np.sqrt(max(q[2], q[3], q[4]))
                ed = 1e-4 * np.sqrt(2 * nn * nn) + 1e-2 * np.sqrt(2 * q[5])
                status = (it, r, sn, ep, ed)
                if r <= ep and sn <= ed:
                    break
            it += 1
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank), x0=Lg, x1=Sg, lo=lo, hi=hi,
                 status=np.array(status))
        dist.barrier()
        dist.destroy_process_group()
        return
    if mode == "hip_rpca":
        from epsilon_amd import _solve
        nn = m
        M = problems.robust_pca_data(nn, r=3, density=0.1, seed=seed)
        lo, hi = edist.column_range(nn, rank, world, align=1)
        _solve.set_option("dtype", os.environ.get("EPS_TEST_DTYPE", "f64"))
        edist.init_comm(rank, world, backend="host")
        prob = problems.robust_pca_ir(np.ascontiguousarray(M[lo:hi]), 0.1)
        _solve.shard_keys(["var:L", "var:S", "constraint:0"])
        params = wire.SolverParams(max_iterations=max_iter)
        st, x = _solve.solve(prob.SerializeToString(), [], params.SerializeToString(),
                             prob.expression_data())
        S = wire.SolverStatus.FromString(st)
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank),
                 x0=np.frombuffer(x["var:L"]).reshape(hi - lo, nn, order="F"),
                 x1=np.frombuffer(x["var:S"]).reshape(hi - lo, nn, order="F"), lo=lo, hi=hi,
                 status=np.array([S.num_iterations, S.residuals.r_norm, S.residuals.s_norm,
                                  S.residuals.epsilon_primal, S.residuals.epsilon_dual]),
                 state=S.state)
        _solve.comm_shutdown()
        dist.barrier()
        dist.destroy_process_group()
        return
    if mode == "hip_mnist":
        # BASELINE.json configs[3] shape: multiclass hinge, SAMPLES sharded over the ranks
        from epsilon_amd import _solve
        k, nf = 3, n
        X, Y = problems.multiclass_hinge_data(m, nf, k, seed=seed)
        lo, hi = edist.column_range(m, rank, world, align=1)
        c_vec = -(X.T.dot(Y)).reshape(1, -1, order="F")
        _solve.set_option("dtype", os.environ.get("EPS_TEST_DTYPE", "f64"))
        edist.init_comm(rank, world, backend="host")
        _solve.comm_warmup(64)  # checked all-reduce + all-gather across the ranks
        prob, _ = problems.multiclass_hinge(X[lo:hi], Y[lo:hi], 0.1, c_vec=c_vec)
        _solve.shard_keys(["max_entries:t", "non_negative:y", "constraint:0"])
        params = wire.SolverParams(max_iterations=max_iter)
        st, x = _solve.solve(prob.SerializeToString(), [], params.SerializeToString(),
                             prob.expression_data())
        S = wire.SolverStatus.FromString(st)
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank),
                 x0=np.frombuffer(x["max_entries:t"]), x1=np.frombuffer(x["var:Theta"]), lo=lo, hi=hi,
                 status=np.array([S.num_iterations, S.residuals.r_norm, S.residuals.s_norm,
                                  S.residuals.epsilon_primal, S.residuals.epsilon_dual]),
                 state=S.state)
        _solve.comm_shutdown()
        dist.barrier()
        dist.destroy_process_group()
        return
    if mode == "oracle":
        # unrolled compiled-lasso sweep on this rank's column slab
        G = allreduce(Ag.dot(Ag.T))                      # Gram: one all-reduce at Init
        Minv = np.linalg.inv(np.eye(m) + 2 * G)
        x0 = np.zeros(ng); x1 = np.zeros(ng); u = np.zeros(ng); y0 = np.zeros(ng); y1 = np.zeros(ng)
        it = 0
        status = None
        while it < max_iter:
            y1_prev = y1.copy()
            u = u - y0 - y1
            u = u + y0
            t = allreduce(Ag.dot(u))                     # the one all-reduce per sweep
            w = Minv.dot(b - t)
            x0 = u + 2 * Ag.T.dot(w)
            y0 = x0.copy()
            u = u - y0
            u = u + y1
            v = -u
            x1 = np.sign(v) * np.maximum(np.abs(v) - lam, 0)
            y1 = -x1
            u = u - y1
            if it % 10 == 0:
                s = allreduce(np.array([np.sum((x0 - x1) ** 2), np.sum((y1 - y1_prev) ** 2),
                                        np.sum(x0 ** 2), np.sum(x1 ** 2), np.sum(u ** 2)]))
                r, sn = np.sqrt(s[0]), np.sqrt(s[1])
                ep = 1e-4 * np.sqrt(n) + 1e-2 * max(np.sqrt(s[2]), np.sqrt(s[3]))
                ed = 1e-4 * np.sqrt(2 * n) + 1e-2 * np.sqrt(2 * s[4])
                status = (it, r, sn, ep, ed)
                if r <= ep and sn <= ed:
                    break
            it += 1
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank), x0=x0, x1=x1, lo=lo, hi=hi,
                 status=np.array(status))
    else:
        from epsilon_amd import _solve
        dtype = os.environ.get("EPS_TEST_DTYPE", "f64")
        _solve.set_option("dtype", dtype)
        want_peer = os.environ.get("EPS_TEST_PEER") == "1"
        on, why = edist.init_comm(rank, world, backend="host", peer=want_peer)
        assert on == want_peer, why
        prob = problems.lasso_ir(ir.dense_matrix(Ag), ir.constant(b), lam, ng)
        edist.mark_sharded(None, prob)
        params = wire.SolverParams(max_iterations=max_iter)
        _solve.profile_enable(True)
        _solve.profile_reset()
        st, x = _solve.solve(prob.SerializeToString(), [], params.SerializeToString(),
                             prob.expression_data())
        tags = sorted(t.split(":")[0] for t in _solve.profile_dump())
        _solve.profile_enable(False)
        S = wire.SolverStatus.FromString(st)
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank),
                 x0=np.frombuffer(x["separate:var:x:sum_square"]), x1=np.frombuffer(x["var:x"]),
                 lo=lo, hi=hi, tags=np.array(sorted(set(tags))),
                 status=np.array([S.num_iterations, S.residuals.r_norm, S.residuals.s_norm,
                                  S.residuals.epsilon_primal, S.residuals.epsilon_dual]),
                 state=S.state)
        _solve.comm_shutdown()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
