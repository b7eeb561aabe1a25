"""End-to-end wall time of NMF_TM_Estimator(handle_tfidf, handle_normalization).fit on raw term counts at C3 scale:
preprocessing on the device (default float64 storage, as the host route yields; and float32 storage)."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from rri_nmf_amd import sklearn_interface as si
n, d, k = 100000, 10000, 50
rs = np.random.RandomState(0)
lam = (rs.rand(n, k).astype(np.float32) ** 4) @ (rs.rand(k, d).astype(np.float32) ** 4)
X = rs.poisson(lam * (3.0 / lam.mean())).astype(np.float32)
del lam
print('counts: %.1f %% non-zero' % (100.0 * np.count_nonzero(X) / X.size), flush=True)
for kw, name in (({}, 'float64 storage (default)'), ({'dtype': np.float32}, 'float32 storage')):
    est = si.NMF_TM_Estimator(n, d, k, random_state=0, max_iter=30, handle_tfidf=True, handle_normalization=True,
                              nmf_kwargs=kw)
    t0 = time.perf_counter()
    est.fit(X)
    t1 = time.perf_counter()
    W = est.transform(X[:5000])
    t2 = time.perf_counter()
    oh = est.nmf_outputs['obj_history']
    print('%-28s fit %.2f s (%d sweeps, objective %.6g -> %.6g), transform of 5000 documents %.2f s, idf in [%.3f, %.3f]'
          % (name, t1 - t0, len(oh), oh[0], oh[-1], t2 - t1, est.idf.min(), est.idf.max()), flush=True)
    assert np.all(np.diff(oh) <= 1e-12 * abs(oh[0])) and abs(W.sum(1) - 1).max() < 1e-9
