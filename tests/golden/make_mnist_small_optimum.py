#!/usr/bin/env python3
"""Independent reference value for the multiclass hinge problem on tests/golden/mnist_small.npz
(docs/notebooks/mnist.rst:88-105: X / 255, k = 10, lam = 1): L-BFGS (scipy) on the log-sum-exp
smoothing of the max, with continuation in the smoothing parameter mu; the EXACT objective of the
final point is an upper bound of the optimum, and  mu * m * log(k)  bounds how far the smoothed
optimum can be from it.  Writes tests/golden/mnist_small_optimum.json.  No solver code of this
repository or of the reference is involved (numpy + scipy only)."""
import json
import os
import time

import numpy as np
import scipy.optimize

HERE = os.path.dirname(os.path.abspath(__file__))
d = np.load(os.path.join(HERE, "mnist_small.npz"))
X = d["X"].astype(np.float64) / 255.0
y = d["y"].astype(int)
m, nf = X.shape
k = 10
Y = np.zeros((m, k))
Y[np.arange(m), y] = 1
lam = 1.0
XtY = X.T.dot(Y)


def exact(Th):
    S = X.dot(Th) + 1 - Y
    return float(S.max(axis=1).sum() - np.sum(XtY * Th) + lam * np.sum(Th ** 2))


def smoothed(th, mu):
    Th = th.reshape(nf, k)
    Sx = (X.dot(Th) + 1 - Y) / mu
    mx = Sx.max(axis=1, keepdims=True)
    lse = mx + np.log(np.exp(Sx - mx).sum(axis=1, keepdims=True))
    P = np.exp(Sx - lse)
    val = mu * lse.sum() - np.sum(XtY * Th) + lam * np.sum(Th ** 2)
    return val, (X.T.dot(P) - XtY + 2 * lam * Th).ravel()


th = np.zeros(nf * k)
t0 = time.time()
trace = []
for mu in (2e-2, 5e-3, 1e-3, 2e-4, 5e-5):
    r = scipy.optimize.minimize(smoothed, th, args=(mu,), jac=True, method="L-BFGS-B",
                                options=dict(maxiter=20000, maxfun=25000, ftol=1e-15, gtol=1e-8, maxcor=30))
    th = r.x
    trace.append(dict(mu=mu, iterations=int(r.nit), smoothed=float(r.fun), exact=exact(th.reshape(nf, k))))
    print(trace[-1], "%.1f s" % (time.time() - t0), flush=True)
out = dict(problem="multiclass hinge, X = mnist_small / 255 (2000 x 784), k = 10, lam = 1",
           objective_upper_bound=trace[-1]["exact"],
           objective_lower_bound=trace[-1]["smoothed"] - trace[-1]["mu"] * m * np.log(k),
           method="scipy L-BFGS-B on the log-sum-exp smoothing, continuation in mu", trace=trace)
json.dump(out, open(os.path.join(HERE, "mnist_small_optimum.json"), "w"), indent=1)
print(out["objective_lower_bound"], out["objective_upper_bound"])
