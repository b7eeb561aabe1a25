"""Both schedules against the CPU oracle at k beyond one round of Gram loads, sweep by sweep: is a growing difference between the
two schedules a bug of one of them, or what a k-step Gauss-Seidel chain does to two summation orders?
    python3 tools/onchip_large_k_check.py [n d k planted_rank sweeps [nndsvda]]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from rri_nmf_amd.engine import RRIEngine
from rri_nmf_amd.synthetic import planted_X, scaled_init
from oracle import rri_oracle as orc

n, d, k, rank, sweeps = (int(v) for v in (sys.argv[1:6] if len(sys.argv) >= 6 else (5000, 800, 47, 47, 3)))
X = planted_X(n, d, rank, seed=n + d, dtype=np.float32)
W0, T0 = scaled_init(X, k, seed=5)
if 'nndsvda' in sys.argv:       # the start the estimators make, instead of a random one
    from rri_nmf_amd import initialization
    W0, T0 = initialization.initialize_nmf(np.asarray(X, dtype=np.float64), k, init='nndsvda', random_state=0)


def relfro(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def run(on, s):
    os.environ['RRI_ONCHIP'] = '1' if on else '0'
    with RRIEngine(n, d, k, dtype=np.float32) as e:
        e.upload_X(X), e.set_W(W0), e.set_T(T0), e.set_params()
        e.sweep(s)
        return e.get_W(), e.get_T(), e.onchip_info()


X64 = np.asarray(X, dtype=np.float64)
if 'warm' in sys.argv:          # both schedules from the state after two sweeps (of the launch-per-phase schedule)
    W0, T0, _ = run(False, 2)
for s in range(1, sweeps + 1):
    Wa, Ta, ia = run(True, s)
    Wb, Tb, ib = run(False, s)
    Wc, Tc = W0.astype(np.float64).copy(), T0.astype(np.float64).copy()
    orc.plain_sweeps(X64, Wc, Tc, s)
    print('%s start, %d sweeps, k = %d, planted rank %d: on-chip %s vs launch-per-phase W %.2e T %.2e | on-chip vs oracle W %.2e T %.2e | '
          'launch-per-phase vs oracle W %.2e T %.2e' % ('warm' if 'warm' in sys.argv else 'random', s, k, rank, ia, relfro(Wa, Wb), relfro(Ta, Tb), relfro(Wa, Wc), relfro(Ta, Tc),
                                                      relfro(Wb, Wc), relfro(Tb, Tc)), flush=True)
