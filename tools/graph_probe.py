"""Captured sweeps (hipGraph) against eager launches at launch-bound sizes: RRI_GRAPH=0/1 python tools/graph_probe.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, '.')
from rri_nmf_amd.engine import RRIEngine
from rri_nmf_amd.synthetic import planted_X, scaled_init
for (n, d, k) in ((10000, 1000, 20), (2000, 500, 10), (20000, 5000, 20)):
    X = planted_X(n, d, k, seed=0, dtype=np.float32)
    W0, T0 = scaled_init(X.astype(np.float64), k, seed=1)
    for flags in (dict(), dict(project_T_each_iter=True, t_row_sum=1.0, w_row_sum=1.0)):
        with RRIEngine(n, d, k, dtype=np.float32) as e:
            e.upload_X(X); e.set_W(W0); e.set_T(T0); e.set_params(**flags)
            e.sweep(5); e.synchronize()
            t0 = time.perf_counter(); e.sweep(200); e.synchronize(); t1 = time.perf_counter() - t0
            t0 = time.perf_counter()
            for _ in range(100):
                e.sweep(1); e.objective()
            t2 = time.perf_counter() - t0
            print('RRI_GRAPH=%s %dx%d k=%d %-12s  sweep(200): %.1f sweeps/s   sweep(1)+objective loop: %.1f iterations/s   obj %.6e'
                  % (os.environ.get('RRI_GRAPH', '0'), n, d, k, 'topic model' if flags else 'plain', 200 / t1, 100 / t2, e.objective()))
