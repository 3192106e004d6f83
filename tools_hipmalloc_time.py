#!/usr/bin/env python3
"""How long hipMalloc / hipFree / a first-touch memset of large buffers take on this box (diagnosis of the
first Init of a process: its 2.4 GB of split operands and workspaces are allocated inside the timed region)."""
import ctypes
import time

hip = ctypes.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
hip.hipFree.argtypes = [ctypes.c_void_p]
hip.hipMemset.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]
hip.hipDeviceSynchronize()
p = ctypes.c_void_p()
hip.hipMalloc(ctypes.byref(p), 1 << 20)
for mb in (64, 400, 1000, 1000, 2000):
    t0 = time.perf_counter()
    q = ctypes.c_void_p()
    rc = hip.hipMalloc(ctypes.byref(q), mb << 20)
    t1 = time.perf_counter()
    hip.hipMemset(q, 0, mb << 20)
    hip.hipDeviceSynchronize()
    t2 = time.perf_counter()
    hip.hipMemset(q, 0, mb << 20)
    hip.hipDeviceSynchronize()
    t3 = time.perf_counter()
    hip.hipFree(q)
    t4 = time.perf_counter()
    print("%5d MB: hipMalloc %.3f ms (rc %d), first memset %.3f ms, second memset %.3f ms, hipFree %.3f ms" %
          (mb, (t1 - t0) * 1e3, rc, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3), flush=True)
