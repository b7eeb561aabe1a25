"""cProfile of one NMF_TM_Estimator.fit at BASELINE config 3's size (second fit of the process): where the HOST time outside the
device calls goes.   python3 tools/e2e_profile.py [n d k sweeps]"""
import cProfile
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, '.')
import torch
from rri_nmf_amd import sklearn_interface as si

n, d, k, sweeps = (int(v) for v in (sys.argv[1:5] if len(sys.argv) >= 5 else (100000, 10000, 50, 30)))
dev = torch.device('cuda', 0)
g = torch.Generator(device=dev)
g.manual_seed(0)
Ts = torch.rand(k, d, device=dev, generator=g) ** 4
Xd = torch.empty(n, d, device=dev, dtype=torch.float32)
for lo in range(0, n, 20000):
    lam = (torch.rand(min(20000, n - lo), k, device=dev, generator=g) ** 4) @ Ts
    Xd[lo:lo + lam.shape[0]] = torch.poisson(lam * (3.0 / lam.mean()), generator=g)
X = Xd.cpu().numpy()
del Xd, Ts, lam
torch.cuda.empty_cache()
for rnd in range(2):
    est = si.NMF_TM_Estimator(n, d, k, handle_tfidf=True, handle_normalization=True, max_iter=sweeps, nmf_kwargs={'dtype': np.float32, 'eps_stop': -1}, random_state=0)
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    est.fit(X)
    pr.disable()
    print('fit %d: %.3f s' % (rnd, time.perf_counter() - t0), flush=True)
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(28)
