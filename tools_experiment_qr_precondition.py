#!/usr/bin/env python3
"""Experiment: does QR preconditioning (one-sided Jacobi on R^T of Y = QR, Drmac-Veselic) cut the
sweeps of the block Jacobi SVD on the reference's robust-PCA matrix?  The QR comes from torch here
(an experiment, not the product path); the decomposition is the library's."""
import ctypes, json, sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from epsilon_amd import _solve

def run(Y):
    L = _solve.lib()
    ms, sw = ctypes.c_double(), ctypes.c_int()
    m, n = Y.shape
    Yc = Y.t().contiguous()  # column-major m x n
    _solve._check(L.eps_bench_svd_device(ctypes.c_void_p(Yc.data_ptr()), ctypes.c_int64(m), ctypes.c_int64(n),
                                         ctypes.c_int(40), ctypes.byref(ms), ctypes.byref(sw)))
    return ms.value, sw.value

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev); g.manual_seed(0)
    _solve.set_option("dtype", "f32")
    Y = torch.randn(n, 10, generator=g, device=dev) @ torch.randn(10, n, generator=g, device=dev)
    Y += (torch.rand(n, n, generator=g, device=dev) < 0.1) * (10.0 * torch.randn(n, n, generator=g, device=dev))
    out = {"n": n}
    out["plain_ms"], out["plain_sweeps"] = run(Y)
    R = torch.linalg.qr(Y.double(), mode="r").R.float()
    out["Rt_ms"], out["Rt_sweeps"] = run(R.t().contiguous())
    R2 = torch.linalg.qr(R.t().double(), mode="r").R.float()
    out["R2t_ms"], out["R2t_sweeps"] = run(R2.t().contiguous())
    print(json.dumps(out))

if __name__ == "__main__":
    main()
