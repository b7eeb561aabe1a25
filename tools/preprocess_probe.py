"""Times tf-idf + row normalisation of a resident X on the device against matrixops on the host (C3 shape by default)."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from rri_nmf_amd.engine import RRIEngine
from rri_nmf_amd.matrixops import tfidf, normalize

n, d = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1000000, 4096)
rng = np.random.RandomState(0)
X = rng.poisson(0.5, size=(n, d)).astype(np.float32)
for store in (np.float32, np.float64):
    with RRIEngine(n, d, 4, dtype=store) as eng:
        t0 = time.perf_counter(); eng.upload_X(X); t1 = time.perf_counter()
        idf = eng.preprocess(tfidf=True, normalize=True); t2 = time.perf_counter()
        eng.upload_X(X); t3 = time.perf_counter()
        eng.preprocess(tfidf=idf, normalize=True); t4 = time.perf_counter()
        rows = eng.X_times(np.ones((d, 1)))
    print('store %s: upload %.3f s, tfidf+normalize %.4f s (given idf %.4f s), row sums in [%.6f, %.6f]'
          % (np.dtype(store).name, t1 - t0, t2 - t1, t4 - t3, rows.min(), rows.max()), flush=True)
m = min(n, 50000)
t0 = time.perf_counter()
Xt, idf_h = tfidf(X[:m].astype(np.float64), return_idf=True); Xn = normalize(Xt)
t1 = time.perf_counter()
print('host matrixops on %d rows: %.3f s -> %.1f s for %d rows' % (m, t1 - t0, (t1 - t0) * n / m, n))
