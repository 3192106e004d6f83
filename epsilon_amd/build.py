"""Build libepsilon_hip.so (gfx950) in-tree with hipcc.

`python -m epsilon_amd.build` or `epsilon_amd.build.build()`.  Objects go to
epsilon_amd/csrc/_build/, the shared library to epsilon_amd/libepsilon_hip.so; both are
git-ignored but travel to the GPU box with the snapshot.
"""

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_build")
LIB = os.path.join(HERE, "libepsilon_hip.so")

HOST_SOURCES = ["wire.cc", "device.cc", "comm.cc", "linear_map.cc", "sparse.cc", "block.cc", "affine.cc", "prox.cc", "prox_more.cc",
                "admm.cc", "capi.cc"]
# -ffp-contract=off for the elementwise / prox kernels: thresholds and projections must pick
# the same branch and produce the same bits as a plain IEEE evaluation.
DEVICE_SOURCES = [("kernels_vec.hip", ["-ffp-contract=off"]),
                  ("kernels_prox.hip", ["-ffp-contract=off"]),
                  ("kernels_fused.hip", ["-ffp-contract=off"]),
                  ("kernels_peer.hip", ["-ffp-contract=off"]),
                  ("kernels_gemv.hip", []),
                  ("kernels_gemv_multi.hip", []),
                  ("kernels_gemm.hip", []),
                  ("kernels_gemm_f16split.hip", []),
                  ("kernels_gemm_f64.hip", []),
                  ("kernels_factor.hip", []),
                  ("kernels_sparse.hip", []),
                  ("kernels_segprox.hip", ["-ffp-contract=off"]),
                  ("kernels_svd.hip", ["-ffp-contract=off"]),
                  ("kernels_tv.hip", ["-ffp-contract=off"]),
                  ("kernels_tv3.hip", ["-ffp-contract=off", "-fno-honor-nans"])]

COMMON = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wall", "-Wno-unused-result",
          "-I" + os.path.join(HERE, "..", "include")]


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _newer(src, dst, deps):
    if not os.path.exists(dst):
        return True
    t = os.path.getmtime(dst)
    return any(os.path.getmtime(d) > t for d in [src] + deps)


def _compile(job):
    src, obj, flags, deps = job
    if not _newer(src, obj, deps):
        return None
    cmd = [hipcc()] + COMMON + flags + ["-x", "hip", "-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, " ".join(cmd), r.stderr))
    return r.stderr


def build(verbose=False, jobs=None):
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(HERE, "..", "include", "epsilon_hip.h"))
    work = []
    for s in HOST_SOURCES:
        work.append((os.path.join(CSRC, s), os.path.join(OBJ, s + ".o"), [], headers))
    for s, flags in DEVICE_SOURCES:
        work.append((os.path.join(CSRC, s), os.path.join(OBJ, s + ".o"), flags, headers))
    jobs = jobs or min(8, os.cpu_count() or 1)
    with ThreadPoolExecutor(max_workers=jobs) as ex:
        for warn in ex.map(_compile, work):
            if verbose and warn:
                sys.stderr.write(warn)
    objs = [w[1] for w in work]
    if _newer(objs[0], LIB, objs[1:]):
        cmd = [hipcc(), "-shared", "-fPIC", "--offload-arch=gfx950", "-o", LIB] + objs + ["-ldl"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s" % r.stderr)
    return LIB


if __name__ == "__main__":
    print(build(verbose=True))
