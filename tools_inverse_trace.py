#!/usr/bin/env python3
"""One explicit SPD inverse at n = 10^4 (fp32) for a kernel trace: rocprofv3 --kernel-trace --stats -- python3 tools_inverse_trace.py"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from epsilon_amd import _solve  # noqa: E402

L = _solve.lib()
_solve.set_option("dtype", "f32")
ms = ctypes.c_double()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
_solve._check(L.eps_bench_spd_inverse(ctypes.c_int64(n), ctypes.c_int(2), ctypes.byref(ms)))
print("spd_inverse n=%d: %.3f ms" % (n, ms.value), flush=True)
