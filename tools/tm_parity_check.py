import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rri_nmf_amd.engine import RRIEngine
from rri_nmf_amd.synthetic import planted_X, scaled_init
from oracle import rri_oracle as orc
rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
TM = dict(project_T_each_iter=True, t_row_sum=1.0, w_row_sum=1.0)
n, d, k = 10000, 1000, 20
X = planted_X(n, d, k, seed=81, dtype=np.float32); X = X / X.sum(1, keepdims=True)
W0, T0 = scaled_init(X, k, seed=82); T0 = T0 / T0.sum(1, keepdims=True)
X64 = np.asarray(X, dtype=np.float64)
for sweeps in (1, 2, 3):
    ref = orc.nmf(X64, k, W_in=W0.astype(np.float64).copy(), T_in=T0.astype(np.float64).copy(), max_iter=sweeps, eps_stop=-1,
                  project_W_each_iter=False, do_final_project_W=False, **TM)
    for on in ('1', '0'):
        os.environ['RRI_ONCHIP'] = on
        for dt in (np.float32, np.float64):
            with RRIEngine(n, d, k, dtype=dt) as e:
                e.upload_X(X), e.set_W(W0), e.set_T(T0), e.set_params(**TM)
                e.sweep(sweeps)
                W, T = e.get_W(), e.get_T()
            rowerr = np.linalg.norm(T - ref['T'], axis=1) / np.linalg.norm(ref['T'], axis=1)
            print('sweeps', sweeps, 'RRI_ONCHIP', on, dt.__name__, 'W %.2e T %.2e' % (rel(W, ref['W']), rel(T, ref['T'])),
                  'worst T rows', np.argsort(-rowerr)[:3], rowerr.max(), flush=True)
