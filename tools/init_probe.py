"""Time the NNDSVD initialisation both ways at a large size: scikit-learn's randomized_svd on the host (what the
reference does, initialization.py:105) and the same algorithm with its products with X on the device; with a
breakdown of the device-assisted path.  usage: python tools/init_probe.py [n d k]"""
import sys
import time

import numpy as np

sys.path.insert(0, '.')
from rri_nmf_amd.engine import RRIEngine
from rri_nmf_amd import initialization as ini

n, d, k = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (100000, 10000, 50)
rs = np.random.RandomState(0)
X = (rs.rand(n, k).astype(np.float32) ** 4) @ (rs.rand(k, d).astype(np.float32) ** 4)
X += 0.01 * rs.rand(n, d).astype(np.float32)
t0 = time.perf_counter()
eng = RRIEngine(n, d, k, dtype=np.float32)
eng.upload_X(X)
t1 = time.perf_counter()
acc = {'X_times': 0.0, 'Xt_times': 0.0}
calls = {'X_times': 0, 'Xt_times': 0}
for name in ('X_times', 'Xt_times'):
    def wrap(f, name=name):
        def g(B):
            ta = time.perf_counter(); r = f(B); acc[name] += time.perf_counter() - ta; calls[name] += 1
            return r
        return g
    setattr(eng, name, wrap(getattr(eng, name)))
Wd, Td = ini.initialize_nmf(X, k, 'nndsvd', random_state=0, engine=eng)
t2 = time.perf_counter()
Wh, Th = ini.initialize_nmf(X, k, 'nndsvd', random_state=0)
t3 = time.perf_counter()
rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
print({'n': n, 'd': d, 'k': k, 'upload_s': round(t1 - t0, 3), 'device_init_s': round(t2 - t1, 3),
       'of_which': {kk: (calls[kk], round(v, 3)) for kk, v in acc.items()}, 'host_side_s': round(t2 - t1 - sum(acc.values()), 3),
       'host_init_s': round(t3 - t2, 3), 'relW': rel(Wd, Wh), 'relT': rel(Td, Th)})
