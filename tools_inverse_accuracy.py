import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from epsilon_amd import _solve, ir
rng = np.random.RandomState(5)
n = 2000
Q, _ = np.linalg.qr(rng.randn(n, n))
for cond in (1e2, 1e4, 1e5):
    ev = np.exp(np.linspace(0, np.log(cond), n))
    M = (Q * ev).dot(Q.T)
    M = (0.5 * (M + M.T)).astype(np.float32).astype(np.float64)
    _solve.set_option("dtype", "f32")
    W = _solve.linear_map_inverse(ir.dense_matrix(M))
    Wx = np.linalg.inv(M)
    print("cond %.0e: |W M - I|_max %.3e  |W - inv|_F/|inv|_F %.3e" % (cond, np.abs(W.dot(M) - np.eye(n)).max(), np.linalg.norm(W - Wx) / np.linalg.norm(Wx)), flush=True)
