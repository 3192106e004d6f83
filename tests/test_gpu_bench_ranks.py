"""bench.py's N > 1 path, rehearsed on ONE GPU: two ranks share the device (gloo + host-staged
collectives, `--comm host`; the peer-write window and the hipGraph replay are the real ones) and
run BASELINE.json configs[1] at full size.  The column-sharded solve is the same iteration as the
1-GPU one (reference algorithms/prox_admm.cc:131-160), so it has to stop after the same number of
sweeps - which it did not while the fused Cholesky step wrote the factored diagonal block into
the matrix its late-starting workgroups were still reading (a shared GPU delays them): 91-261
sweeps instead of 51, differently on every run.  Also checks that the peer window survives its
validation solve (its iterates equal the collective path's).  The two-rank run is started as
`python bench.py --gpus 2 --comm host` WITHOUT torchrun: bench.py launches its own ranks."""

import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _bench(ranks, self_launch=False):
    common = ["--no-cpu-baseline", "--steps", "20", "--warmup", "5"]
    if ranks == 1:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + common
    elif self_launch:
        # the way the driver calls the N = 1 case, with N > 1: no torchrun, no rank environment -
        # bench.py has to start its ranks itself (epsilon_amd/launch.py)
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--comm", "host"] + common
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks),
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
               os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--comm", "host"] + common
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_two_ranks_on_one_gpu_stop_after_the_same_sweeps():
    one = _bench(1)
    two = _bench(2, self_launch=True)
    assert one["state_at_eps"] == "OPTIMAL" and two["state_at_eps"] == "OPTIMAL"
    assert two["iters_to_eps"] == one["iters_to_eps"], (one["iters_to_eps"], two["iters_to_eps"])
    assert two["n_gpus"] == 2 and "column-sharded x2" in two["config"]["parallelism"]
    # the peer window passed its validation against the collective path
    assert "peer-write window inside the sweep kernels" in two["config"]["comm"], two["config"]["comm"]
