#!/usr/bin/env python3
"""A few launches of the split-f16 product for a counter pass (rocprofv3 --pmc ... -- python3 tools_gemm_pmc.py):
the Gram product of config 2 (10^4 x 5*10^4, random operands) and one 10^4-cubed product."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("EPSILON_HIP_BENCH_RANDOM", "1")
from epsilon_amd import _solve  # noqa: E402

L = _solve.lib()
_solve.set_option("dtype", "f32")
ms = ctypes.c_double()
for (ta, tb, M, N, K, lower) in [(0, 1, 10000, 10000, 50000, 2), (0, 0, 10000, 10000, 10000, 0)]:
    _solve._check(L.eps_bench_gemm(ctypes.c_int(ta), ctypes.c_int(tb), ctypes.c_int64(M), ctypes.c_int64(N),
                                   ctypes.c_int64(K), ctypes.c_int(lower), ctypes.c_int(2), ctypes.byref(ms)))
    print("%dx%dx%d lower=%d: %.3f ms" % (M, N, K, lower, ms.value), flush=True)
