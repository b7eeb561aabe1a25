"""End-to-end wall time of NMF_RS_Estimator.fit at C5 scale (index pairs + ratings in, factors out), by phase."""
import cProfile, pstats, sys, time
import numpy as np
sys.path.insert(0, '.')
from rri_nmf_amd import sklearn_interface as si
n, d, k = 100000, 10000, 50
rs = np.random.RandomState(0)
m = 50000000
ij = np.column_stack((rs.randint(0, n, m), rs.randint(0, d, m)))
y = rs.randint(1, 6, m).astype(np.float64)
est = si.NMF_RS_Estimator(n, d, k, random_state=0, max_iter=int(sys.argv[1]) if len(sys.argv) > 1 else 10)
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
est.fit(ij, y)
pr.disable()
print('fit wall %.2f s, sweeps run %d' % (time.perf_counter() - t0, len(est.nmf_outputs['iter_cputime'])))
pstats.Stats(pr).sort_stats('cumtime').print_stats(26)
