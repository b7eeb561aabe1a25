import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rri_nmf_amd.engine import RRIEngine
from rri_nmf_amd.synthetic import planted_X, scaled_init
from oracle import rri_oracle as orc
rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
for (n, d, k, kp) in ((5003, 1000, 64, 64), (5003, 1000, 40, 40), (5003, 1000, 33, 33), (5003, 1000, 32, 32), (2000, 300, 64, 64)):
    X = planted_X(n, d, kp, seed=3, dtype=np.float32)
    W0, T0 = scaled_init(X, k, seed=5)
    Wc, Tc = W0.astype(np.float64).copy(), T0.astype(np.float64).copy()
    orc.plain_sweeps(np.asarray(X, dtype=np.float64), Wc, Tc, 1)
    for on in ('1', '0'):
        os.environ['RRI_ONCHIP'] = on
        with RRIEngine(n, d, k, dtype=np.float32) as e:
            e.upload_X(X), e.set_W(W0), e.set_T(T0), e.set_params(reset_topic_method=None)
            e.sweep(1)
            W, T = e.get_W(), e.get_T()
            colerr = np.linalg.norm(W - Wc, axis=0) / (np.linalg.norm(Wc, axis=0) + 1e-300)
            rowerr = np.linalg.norm(T - Tc, axis=1) / (np.linalg.norm(Tc, axis=1) + 1e-300)
            print(n, d, k, 'RRI_ONCHIP=' + on, 'vs oracle after 1 sweep: W %.2e T %.2e' % (rel(W, Wc), rel(T, Tc)),
                  'first bad topic W', int(np.argmax(colerr > 1e-9)) if (colerr > 1e-9).any() else None,
                  'T', int(np.argmax(rowerr > 1e-9)) if (rowerr > 1e-9).any() else None, flush=True)
