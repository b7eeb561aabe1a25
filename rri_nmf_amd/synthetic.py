"""Seeded synthetic inputs for the RRI path.

Planted non-negative low rank plus noise; the same recipe feeds the golden
generator, the parity tests and bench.py, so every leg sees identical X / W0 / T0.
Host-side numpy only.
"""
import numpy as np


def planted_X(n, d, k, seed=0, dtype=np.float32, noise=0.01, density=0.3):
    """X = W* T* + noise*U with 30 %-dense uniform factors."""
    rs = np.random.RandomState(seed)
    Ws = rs.rand(n, k) * (rs.rand(n, k) < density)
    Ts = rs.rand(k, d) * (rs.rand(k, d) < density)
    X = Ws.dot(Ts) + noise * rs.rand(n, d)
    return np.ascontiguousarray(X.astype(dtype))


def scaled_init(X, k, seed=1):
    """W0, T0 = sqrt(mean(X)/k) * U: keeps W0 T0 at the scale of X so no topic
    collapses in the first sweeps (a raw U(0,1) init triggers resets)."""
    n, d = X.shape
    rs = np.random.RandomState(seed)
    a = np.sqrt(float(X.mean()) / k)
    W0 = a * rs.rand(n, k)
    T0 = a * rs.rand(k, d)
    return W0.astype(X.dtype), T0.astype(X.dtype)


def observed_mask(n, d, frac=0.05, seed=2, dtype=np.float32):
    """Dense 0/1 observation mask for the weighted (WRRI) flavour."""
    rs = np.random.RandomState(seed)
    return np.ascontiguousarray((rs.rand(n, d) < frac).astype(dtype))
