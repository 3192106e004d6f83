#!/usr/bin/env python3
"""Writes tests/golden/mnist_small.npz from the data file the reference ships for its MNIST
problems: python/epopt/problems/mnist_small.mat (X 2000 x 784 uint8 pixels, y 1 x 2000 uint8 labels;
used by python/epopt/problems/mnist.py:14-22 and named in docs/notebooks/mnist.rst).  The arrays
are DATA (a 2000-sample subset of MNIST), copied bit for bit; no reference code is involved.
Run in the build container only (the reference tree does not exist on the GPU box):

    python tests/golden/make_mnist_small.py
"""
import hashlib
import os

import numpy as np
import scipy.io

SRC = "/root/reference/python/epopt/problems/mnist_small.mat"
DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "mnist_small.npz")

d = scipy.io.loadmat(SRC)  # MATLAB v5 container: plain arrays, nothing is executed
X = np.ascontiguousarray(d["X"], dtype=np.uint8)
y = np.ascontiguousarray(d["y"].ravel(), dtype=np.uint8)
assert X.shape == (2000, 784) and y.shape == (2000,)
np.savez_compressed(DST, X=X, y=y)
print(DST, os.path.getsize(DST), "bytes; sha256(X) =", hashlib.sha256(X.tobytes()).hexdigest()[:16],
      "sha256(y) =", hashlib.sha256(y.tobytes()).hexdigest()[:16])
