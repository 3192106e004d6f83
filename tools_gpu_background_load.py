#!/usr/bin/env python3
"""Keeps the GPU busy from a second process (bursts of matrix products and device copies through
the library's own microbenchmark entry points, with short pauses) while another command runs - a
timing perturbation for the tests: kernels whose workgroups depend on one another's progress, or
that read memory another workgroup of the same launch writes, only show it when CUs are taken
away from them (the Cholesky race of round 2 was found this way).  No torch: the child is up in
about a second.  Prints "READY" once the first burst has run.

usage: tools_gpu_background_load.py <seconds>      (tests/test_gpu_under_load.py starts it as a child)
"""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from epsilon_amd import _solve  # noqa: E402

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
L = _solve.lib()
ms = ctypes.c_double()
t_end = time.time() + secs
i = 0
while time.time() < t_end:
    _solve._check(L.eps_bench_gemm(ctypes.c_int(0), ctypes.c_int(1), ctypes.c_int64(4096), ctypes.c_int64(4096),
                                   ctypes.c_int64(4096), ctypes.c_int(0), ctypes.c_int(3), ctypes.byref(ms)))
    _solve._check(L.eps_bench_stream(None, ctypes.c_size_t(1 << 30), ctypes.c_int(2), ctypes.c_int(0),
                                     ctypes.c_int(2), ctypes.byref(ms)))
    if i == 0:
        print("READY", flush=True)
    i += 1
    if i % 7 == 0:
        time.sleep(0.003)
print("background load done, %d bursts" % i, flush=True)
