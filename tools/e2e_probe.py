"""End-to-end wall time of NMF_TM_Estimator.fit at C3 scale (host array in, host arrays out), by phase."""
import cProfile, pstats, sys, time
import numpy as np
sys.path.insert(0, '.')
import logging
from rri_nmf_amd import sklearn_interface as si
n, d, k = 100000, 10000, 50
rs = np.random.RandomState(0)
X = (rs.rand(n, k).astype(np.float32) ** 4) @ (rs.rand(k, d).astype(np.float32) ** 4)
X += 0.01 * rs.rand(n, d).astype(np.float32)
X /= X.sum(1, keepdims=True)
est = si.NMF_TM_Estimator(n, d, k, random_state=0, max_iter=30)
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
est.fit(X)
pr.disable()
print('fit wall %.2f s, sweeps run %d' % (time.perf_counter() - t0, len(est.nmf_outputs['iter_cputime'])))
pstats.Stats(pr).sort_stats('cumtime').print_stats(22)
