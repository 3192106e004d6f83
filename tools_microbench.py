#!/usr/bin/env python3
"""Kernel micro-benchmarks on the GPU box (not part of the judged bench): prints one line per
case with ms and the algorithmic rate."""
import ctypes
import sys

sys.path.insert(0, ".")
from epsilon_amd import _solve  # noqa: E402

L = _solve.lib()


def gemv(trans, rows, cols, iters=50, dtype="f32"):
    _solve.set_option("dtype", dtype)
    ms = ctypes.c_double()
    _solve._check(L.eps_bench_gemv(ctypes.c_int(trans), ctypes.c_int64(rows), ctypes.c_int64(cols),
                                   ctypes.c_int(iters), ctypes.byref(ms)))
    sz = 4 if dtype == "f32" else 8
    print("gemv_%s %dx%d %s: %.4f ms  %.0f GB/s (dense bytes)" % (["n", "t", "sym"][trans], rows, cols, dtype,
                                                                ms.value, rows * cols * sz / ms.value / 1e6), flush=True)


def gemm(ta, tb, M, N, K, lower=0, iters=3, dtype="f32"):
    _solve.set_option("dtype", dtype)
    ms = ctypes.c_double()
    _solve._check(L.eps_bench_gemm(ctypes.c_int(ta), ctypes.c_int(tb), ctypes.c_int64(M), ctypes.c_int64(N),
                                   ctypes.c_int64(K), ctypes.c_int(lower), ctypes.c_int(iters), ctypes.byref(ms)))
    fl = 2.0 * M * N * K
    print("gemm %s%s %dx%dx%d lower=%d %s: %.3f ms  %.1f TFLOP/s (GEMM-equivalent)" %
          ("T" if ta else "N", "T" if tb else "N", M, N, K, lower, dtype, ms.value, fl / ms.value / 1e9), flush=True)


def inverse(n, iters=2, dtype="f32"):
    _solve.set_option("dtype", dtype)
    ms = ctypes.c_double()
    _solve.set_option("profile_filter", "potrf,trtri,lauum")
    _solve.profile_enable(True)
    _solve.profile_reset()
    _solve._check(L.eps_bench_spd_inverse(ctypes.c_int64(n), ctypes.c_int(iters), ctypes.byref(ms)))
    prof = _solve.profile_dump()
    _solve.profile_enable(False)
    phases = ", ".join("%s %.2f" % (k.split(":")[0], t / c) for k, (c, t) in sorted(prof.items()) if c)
    print("spd_inverse n=%d %s: %.2f ms  (%s)" % (n, dtype, ms.value, phases), flush=True)


def inverse_columns(n, cnt, iters=2, dtype="f32"):
    _solve.set_option("dtype", dtype)
    ms = ctypes.c_double()
    _solve._check(L.eps_bench_spd_inverse_columns(ctypes.c_int64(n), ctypes.c_int64(cnt), ctypes.c_int(iters),
                                                  ctypes.byref(ms)))
    print("spd_inverse_columns n=%d cnt=%d %s: %.2f ms" % (n, cnt, dtype, ms.value), flush=True)


def prox(kind, n=10 ** 8, iters=20, dtype="f32"):
    """standalone elementwise prox kernels (reference prox/scaled_zone.cc:90-101, non_negative.cc:8)"""
    _solve.set_option("dtype", dtype)
    ms = ctypes.c_double()
    _solve._check(L.eps_bench_prox(ctypes.c_int(kind), ctypes.c_int64(n), ctypes.c_int(iters), ctypes.byref(ms)))
    sz = 4 if dtype == "f32" else 8
    arrays = 3 if kind == 1 else 2
    name = ["ScaledZoneVecKernel (scalar parameters)", "ScaledZoneVecKernel (per-element threshold)",
            "MaxZeroKernel"][kind]
    print('{"kernel": "%s", "n": %d, "dtype": "%s", "ms": %.4f, "algorithmic_bytes": %d, "GBs": %.0f, '
          '"frac_of_hbm_peak": %.3f}' % (name, n, dtype, ms.value, arrays * n * sz, arrays * n * sz / ms.value / 1e6,
                                         arrays * n * sz / ms.value / 1e6 / 8000.0), flush=True)


def svd(n, m=None, rank=10, max_sweeps=40, perturb=1e-3, dtype="f32"):
    """SVD behind the nuclear-norm prox on the reference's robust-PCA matrix (ortho_invariant.cc:36-50)"""
    m = m or n
    _solve.set_option("dtype", dtype)
    ms_c, ms_w = ctypes.c_double(), ctypes.c_double()
    sw_c, sw_w = ctypes.c_int(), ctypes.c_int()
    d = (ctypes.c_double * 6)()
    _solve._check(L.eps_bench_svd(ctypes.c_int64(m), ctypes.c_int64(n), ctypes.c_int(rank), ctypes.c_int(max_sweeps),
                                  ctypes.c_double(perturb), ctypes.byref(ms_c), ctypes.byref(sw_c),
                                  ctypes.byref(ms_w), ctypes.byref(sw_w), d))
    print('{"kernel": "JacobiSvd", "m": %d, "n": %d, "dtype": "%s", "cold_ms": %.1f, "cold_sweeps": %d, '
          '"warm_ms": %.1f, "warm_sweeps": %d, "perturb": %g, "cold_defects": [%.2e, %.2e, %.2e], '
          '"warm_defects": [%.2e, %.2e, %.2e]}' % (m, n, dtype, ms_c.value, sw_c.value, ms_w.value, sw_w.value,
                                                   perturb, d[0], d[1], d[2], d[3], d[4], d[5]), flush=True)


if __name__ == "__main__":
    what = sys.argv[1:] or ["gemv", "gemm", "inverse"]
    if "prox" in what:
        for dt in ("f32", "f64"):
            for kind in (0, 1, 2):
                prox(kind, dtype=dt)
    for w in what:
        if w.startswith("svd"):  # svd:<n>[:<max_sweeps>[:<rank>]]
            parts = w.split(":")
            svd(int(parts[1]) if len(parts) > 1 else 4096, max_sweeps=int(parts[2]) if len(parts) > 2 else 40,
                rank=int(parts[3]) if len(parts) > 3 else 10)
    if "gemv" in what:
        gemv(0, 10000, 50000)
        gemv(1, 10000, 50000)
        gemv(0, 10000, 10000)
        gemv(1, 10000, 10000)
        gemv(2, 10000, 10000)
        gemv(0, 10000, 6272)
        gemv(1, 10000, 6272)
    if "gemm" in what:
        gemm(0, 1, 10000, 10000, 50000, lower=1, iters=2)
        gemm(0, 1, 4096, 4096, 4096, iters=5)
        gemm(0, 0, 4096, 4096, 4096, iters=5)
        gemm(1, 0, 4096, 4096, 4096, iters=5)
    if "gemm64" in what:
        for mode in ("auto", "generic", "mfma_simple"):
            _solve.set_option("gemm", mode)
            print("EPSILON_HIP_GEMM=%s" % mode, flush=True)
            gemm(0, 1, 4096, 4096, 4096, iters=3, dtype="f64")
            gemm(0, 0, 4096, 4096, 4096, iters=3, dtype="f64")
            gemm(1, 0, 4096, 4096, 4096, iters=3, dtype="f64")
            gemm(1, 1, 4096, 4096, 4096, iters=3, dtype="f64")
            if mode != "mfma_simple":
                gemm(0, 1, 10000, 10000, 50000, lower=1, iters=1, dtype="f64")
        _solve.set_option("gemm", "auto")
        inverse(10000, iters=1, dtype="f64")
        _solve.set_option("dtype", "f32")
    if "syrk" in what:
        gemm(0, 1, 10000, 10000, 50000, lower=2, iters=2)
    if "inverse1" in what:
        inverse(10000, iters=1)
    if "inverse" in what:
        inverse(10000)
        inverse(2048)
        inverse_columns(10000, 1250)
        inverse_columns(10000, 2500)
        inverse_columns(10000, 5000)
