"""Self-launch of the multi-rank benchmarks: `python bench.py --gpus N` with no rank environment
starts N fresh ranks itself.

The driver launches the N > 1 case as `python -m torch.distributed.run ... bench.py --gpus N`
(RANK / WORLD_SIZE set); a user - or a driver - calling `python bench.py --gpus N` directly gets
the same run: this module spawns that command as a CHILD process and the parent exits with the
child's code.  It must be called before anything initialises the GPU (no `import torch`, no HIP
call): replacing or forking a process that holds a HIP context is not allowed on this pool, a
fresh child of a process that never touched the GPU is.

No reference counterpart (the reference has no distributed mode).
"""

import os
import socket
import subprocess
import sys


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


# torch.distributed.run's own parser treats "--m" / "--n" in the script's arguments as ambiguous
# abbreviations of its options: the benchmarks accept these spellings too
_RESPELL = {"--m": "--rows", "--n": "--cols"}


def launched_by_torchrun():
    return "RANK" in os.environ and "WORLD_SIZE" in os.environ


def self_launch_if_needed(script, gpus, argv=None):
    """If `gpus` > 1 and this process is not already one rank of a launch, run the script again as
    `gpus` ranks under torch.distributed.run (one node, 127.0.0.1 rendezvous), wait, and exit with
    its return code.  Returns normally otherwise."""
    if gpus <= 1 or launched_by_torchrun():
        return
    argv = list(sys.argv[1:] if argv is None else argv)
    out = []
    for a in argv:
        key, eq, val = a.partition("=")
        out.append(_RESPELL.get(key, key) + (eq + val if eq else ""))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(script)] + out
    sys.stderr.write("[launch] no RANK in the environment: starting %d ranks: %s\n" % (gpus, " ".join(cmd)))
    sys.stderr.flush()
    rc = subprocess.call(cmd, env=env)
    sys.exit(rc)
