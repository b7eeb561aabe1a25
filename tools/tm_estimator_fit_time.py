"""NMF_TM_Estimator.fit on the reference's text fixture (tests/golden/g1: the estimator's default flag set, 30 sweeps), warm start
passed in so that only the loop is timed: the device loop (rri_sweep_until) against sweep by sweep, the persistent kernel against
the launch-per-phase schedule.
    python3 tools/tm_estimator_fit_time.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from conftest import load_golden                          # noqa: E402
from rri_nmf_amd import nmf as nmf_mod                    # noqa: E402

g = load_golden('g1_tm_estimator')
X = np.asarray(g['X'], dtype=np.float64)
n, d = X.shape
k = 5
W0, T0 = np.asarray(g['W0']), np.asarray(g['T0'])
kw = dict(max_iter=30, eps_stop=-1, compute_obj_each_iter=True, project_W_each_iter=False, w_row_sum=1.0, project_T_each_iter=True,
          t_row_sum=1.0, dtype=np.float32)
print('text fixture %d x %d, k = %d, 30 sweeps of the estimator\'s flag set' % (n, d, k))
for onchip in ('1', '0'):
    for chunk in ('1', '0'):
        os.environ['RRI_ONCHIP'], os.environ['RRI_NMF_CHUNK'] = onchip, chunk
        best = None
        for rep in range(4):
            t0 = time.perf_counter()
            out = nmf_mod.nmf(X, k, W_in=W0, T_in=T0, **kw)
            dt = time.perf_counter() - t0
            best = dt if best is None or (rep > 0 and dt < best) else best
        print('RRI_ONCHIP=%s RRI_NMF_CHUNK=%s: %.2f ms per call (best of 3 after a first), final objective %.12e'
              % (onchip, chunk, best * 1e3, out['obj_history'][-1]), flush=True)
