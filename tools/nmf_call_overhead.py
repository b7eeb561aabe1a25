import os, sys, time
import numpy as np
sys.path.insert(0, '/root/repo')
from rri_nmf_amd import nmf as nmf_mod
from rri_nmf_amd.engine import RRIEngine
from rri_nmf_amd.synthetic import planted_X, scaled_init
n, d, k = 10000, 1000, 20
X = planted_X(n, d, k, seed=1, dtype=np.float32)
W0, T0 = scaled_init(X, k, seed=2)
for it in (60,):
    for rep in range(3):
        t0 = time.perf_counter()
        out = nmf_mod.nmf(X, k, W_in=W0, T_in=T0, max_iter=it, eps_stop=-1, dtype=np.float32)
        t1 = time.perf_counter()
        print('nmf() %d sweeps: %.2f ms total, %.3f ms per sweep; keys %s' % (it, (t1 - t0) * 1e3, (t1 - t0) * 1e3 / it, sorted(out.keys())[:4]), flush=True)
with RRIEngine(n, d, k, dtype=np.float32) as e:
    e.upload_X(X), e.set_W(W0), e.set_T(T0), e.set_params()
    e.sweep(2)
    for mode in ('sweep(1) x60', 'sweep(1)+objective x60', 'sweep(60)'):
        t0 = time.perf_counter()
        if mode == 'sweep(60)':
            e.sweep(60)
        else:
            for _ in range(60):
                e.sweep(1)
                if 'objective' in mode:
                    e.objective()
        e.get_T()
        t1 = time.perf_counter()
        print('%s: %.3f ms per sweep' % (mode, (t1 - t0) * 1e3 / 60), flush=True)
