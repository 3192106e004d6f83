"""Single-rank checks of the peer-window sweep (one process, one GPU, RCCL communicator of size 1,
sharded code path forced on):
  * the sweep through the exchange kernels == the oracle;
  * sweeps replayed from a hipGraph are BIT-identical to the eager launches;
  * the rank-of-8 rehearsal window passes its self test and runs (timing-only mode).
Prints 'PEER_OK ...'."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["EPSILON_HIP_FORCE_SHARDED"] = "1"
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29633")

import torch.distributed as dist  # noqa: E402

from epsilon_amd import _solve, problems, wire  # noqa: E402
from epsilon_amd import dist as edist  # noqa: E402
from oracle import epsilon_oracle as orc  # noqa: E402

dist.init_process_group("gloo", rank=0, world_size=1)
_solve.set_option("dtype", "f32")
on, why = edist.init_comm(0, 1, backend="rccl", peer=True)
assert on, why


def run(m, n, seed, sweeps, graph):
    os.environ["EPSILON_HIP_GRAPH"] = graph
    prob, info = problems.lasso(m, n, seed=seed)
    edist.mark_sharded(None, prob)
    # tolerances of zero: never OPTIMAL, so both the device and the oracle run every sweep
    params = wire.SolverParams(max_iterations=10 ** 6, abs_tol=0.0, rel_tol=0.0)
    s = _solve.Solver(prob.SerializeToString(), params.SerializeToString(), prob.expression_data())
    s.init()
    _solve.profile_reset()
    _solve.profile_enable(True)
    s.run(1)
    tags = _solve.profile_dump()
    _solve.profile_enable(False)
    assert any(t.startswith("peer_reduce_exchange") for t in tags), sorted(tags)
    s.run(sweeps - 1)
    st, x = s.result()
    s.close()
    return prob, x


# NB: EPSILON_HIP_GRAPH is read once per process (static) - so the eager arm runs in a child
if len(sys.argv) > 1 and sys.argv[1] == "eager":
    prob, x = run(200, 500, 3, 45, "0")
    np.savez(sys.argv[2], **{k.replace(":", "_"): np.frombuffer(v) for k, v in x.items()})
    _solve.comm_shutdown()
    dist.destroy_process_group()
    print("PEER_EAGER_DONE")
    sys.exit(0)

prob, xg = run(200, 500, 3, 45, "1")
# oracle after the same 45 sweeps
params = wire.SolverParams(max_iterations=45, abs_tol=0.0, rel_tol=0.0)
st_o, x_o = orc.solve(prob.SerializeToString(), [], params.SerializeToString(), prob.expression_data())
d = max(np.abs(np.frombuffer(xg[k]) - np.frombuffer(x_o[k])).max() for k in x_o)
assert d < 5e-4, d
if len(sys.argv) > 2 and sys.argv[1] == "graph":
    np.savez(sys.argv[2], **{k.replace(":", "_"): np.frombuffer(v) for k, v in xg.items()})

# rehearsal: this process plays rank 0 of 8 (self test inside enable_peer covers both channels)
on, why = _solve.comm_enable_peer(0, 8)
assert on, why
prob, info = problems.lasso(256, 640, seed=5)
edist.mark_sharded(None, prob)
s = _solve.Solver(prob.SerializeToString(),
                  wire.SolverParams(max_iterations=10 ** 6, ignore_stopping_criteria=True).SerializeToString(),
                  prob.expression_data())
s.init()
assert s.run(25) == 25
s.close()
_solve.comm_shutdown()
dist.destroy_process_group()
print("PEER_OK %.3e" % d)
