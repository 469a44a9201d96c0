"""Seeded point sets of the LARGE golden fixtures (tests/golden/golden_*_32k.npz).  TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.

The large fixtures hold sums and gradients of the float64 oracle on 32 768 points at the reference's TRAINED weights -- enough points
for a trained-weight gradient to cross many workgroup steps of the fused kernels (persistent accumulators, in-memory running sums).
Storing 32 768 points (and their residual vectors) would make the fixtures tens of megabytes, so the points are regenerated from a
seed by this function, in the generator (oracle/make_golden.py, build container) and in the tests alike; numpy's PCG64 stream behind
``default_rng(seed).random`` is stable across numpy versions.
"""
from __future__ import annotations

import numpy as np

N_LARGE = 32768


def wave_points(lb, ub, src, n=N_LARGE, seed=33331):
    """Uniform points of the box [lb, ub] outside the source disc ``src`` = (xc, yc, r)  (DelSrcPT, INF:619-622)."""
    lb, ub = np.asarray(lb, dtype=np.float64), np.asarray(ub, dtype=np.float64)
    rng = np.random.default_rng(seed)
    X = lb + (ub - lb) * rng.random((2 * n, 3))
    xc, yc, r = src
    X = X[(X[:, 0] - xc) ** 2 + (X[:, 1] - yc) ** 2 > r * r][:n]
    assert X.shape[0] == n
    return X


def plate_points(n=N_LARGE, seed=44441, r=0.1):
    """Uniform points of the quarter plate [0, .5]^2 x [0, 10] outside the hole (PLATE:881-882, PLATE:42)."""
    rng = np.random.default_rng(seed)
    X = np.array([0.5, 0.5, 10.0]) * rng.random((2 * n, 3))
    X = X[X[:, 0] ** 2 + X[:, 1] ** 2 > r * r][:n]
    assert X.shape[0] == n
    return X


def hole_points(n_theta=64, n_time=64, r=0.1):
    """Hole-edge points x time instants (the layout of PLATE:905-911, denser)."""
    th = np.linspace(0.0, np.pi / 2, n_theta)
    tt = np.linspace(0.0, 10.0, n_time)
    return np.stack([np.repeat(r * np.cos(th), n_time), np.repeat(r * np.sin(th), n_time), np.tile(tt, n_theta)], 1)
