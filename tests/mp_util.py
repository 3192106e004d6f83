import os
import socket
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def run_ranks(world, mode, out_dir, m, n, seed=0, max_iter=10000, env_extra=None, timeout=600):
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ)
        env.update(RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="2")
        env.update(env_extra or {})
        procs.append(subprocess.Popen(
            [sys.executable, os.path.join(HERE, "mp_worker.py"), mode, out_dir, str(m), str(n),
             str(seed), str(max_iter)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o.decode(errors="replace"))
    for r, p in enumerate(procs):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, outs[r][-3000:])
    parts = [np.load(os.path.join(out_dir, "rank%d.npz" % r)) for r in range(world)]
    x0 = np.concatenate([p["x0"] for p in parts])
    x1 = np.concatenate([p["x1"] for p in parts])
    return x0, x1, [p["status"] for p in parts], parts
