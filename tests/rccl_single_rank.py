"""Single-rank RCCL smoke: the sharded code path with a real RCCL communicator of size 1
(EPSILON_HIP_FORCE_SHARDED=1).  Prints 'RCCL_OK <max abs diff vs oracle>'."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["EPSILON_HIP_FORCE_SHARDED"] = "1"
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29631")

import torch.distributed as dist  # noqa: E402

from epsilon_amd import _solve, problems, wire  # noqa: E402
from epsilon_amd import dist as edist  # noqa: E402
from oracle import epsilon_oracle as orc  # noqa: E402

dist.init_process_group("gloo", rank=0, world_size=1)
_solve.set_option("dtype", "f64")
edist.init_comm(0, 1, backend="rccl")
_solve.comm_warmup(1 << 12)  # checked all-reduce + all-gather through RCCL
prob, info = problems.lasso(40, 101, seed=3)
edist.mark_sharded(None, prob)
pb, sb = prob.SerializeToString(), wire.SolverParams().SerializeToString()
st, x = _solve.solve(pb, [], sb, prob.expression_data())
st_o, x_o = orc.solve(pb, [], sb, prob.expression_data())
a, b = wire.SolverStatus.FromString(st), wire.SolverStatus.FromString(st_o)
assert a.state == b.state and a.num_iterations == b.num_iterations, (a, b)
d = max(np.abs(np.frombuffer(x[k]) - np.frombuffer(x_o[k])).max() for k in x_o)
assert d < 1e-9, d
# the row-sharded robust PCA (panel-Gram all-reduce per Jacobi step, max-reduction of the row
# counts) and the consensus form (scalar block sums, max over per-rank term norms) through RCCL
M = problems.robust_pca_data(24, r=3, density=0.1, seed=2)
prob = problems.robust_pca_ir(M, 0.1)
_solve.shard_keys(["var:L", "var:S", "constraint:0"])
pb, sb = prob.SerializeToString(), wire.SolverParams(max_iterations=400).SerializeToString()
st, x = _solve.solve(pb, [], sb, prob.expression_data())
st_o, x_o = orc.solve(pb, [], sb, prob.expression_data())
a, b = wire.SolverStatus.FromString(st), wire.SolverStatus.FromString(st_o)
assert a.state == b.state and a.num_iterations == b.num_iterations, (a, b)
d2 = max(np.abs(np.frombuffer(x[k]) - np.frombuffer(x_o[k])).max() for k in x_o)
assert d2 < 1e-6, d2
A, bb = problems.regression_data(30, 17, seed=4)
lam = 0.3 * np.abs(A.T.dot(bb)).max()
prob = problems.consensus_lasso_local(A, bb, lam)
_solve.shard_keys(["var:x_local", "constraint:0"])
_solve.shard_consensus_terms(True)
pb, sb = prob.SerializeToString(), wire.SolverParams().SerializeToString()
st, x = _solve.solve(pb, [], sb, prob.expression_data())
ref = problems.consensus_lasso(A, bb, lam, 1)
st_o, x_o = orc.solve(ref.SerializeToString(), [], sb, ref.expression_data())
a, b = wire.SolverStatus.FromString(st), wire.SolverStatus.FromString(st_o)
assert a.state == b.state and a.num_iterations == b.num_iterations, (a, b)
d3 = np.abs(np.frombuffer(x[problems.CONSENSUS_Z]) - np.frombuffer(x_o[problems.CONSENSUS_Z])).max()
assert d3 < 1e-9, d3
_solve.comm_shutdown()
dist.destroy_process_group()
print("RCCL_OK %.3e %.3e %.3e" % (d, d2, d3))
