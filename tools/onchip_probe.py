#!/usr/bin/env python3
"""one small problem through the register-resident sweep and through the launch-per-phase schedule; prints both"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rri_nmf_amd.engine import RRIEngine
from rri_nmf_amd.synthetic import planted_X, scaled_init

n, d, k, sweeps = [int(v) for v in (sys.argv[1:5] + ['10000', '1000', '20', '50'][len(sys.argv[1:5]):])]
flags = dict(project_T_each_iter=True, t_row_sum=1.0, w_row_sum=1.0) if 'tm' in sys.argv[5:] else {}      # 'tm': the topic-model flag set
X = planted_X(n, d, k, seed=1, dtype=np.float32)
W0, T0 = scaled_init(X, k, seed=2)
res = {}
for on in ('1', '0'):
    os.environ['RRI_ONCHIP'] = on
    with RRIEngine(n, d, k, dtype=np.float32) as e:
        e.upload_X(X), e.set_W(W0), e.set_T(T0), e.set_params(**flags)
        print('RRI_ONCHIP=%s eligible/launches %r' % (on, e.onchip_info()), flush=True)
        e.sweep(2)
        e.synchronize()
        t0 = time.perf_counter()
        e.sweep(sweeps)
        e.synchronize()
        dt = time.perf_counter() - t0
        print('   %d sweeps in %.3f ms: %.1f sweeps/s, %.2f us per topic step; launches %r' % (sweeps, 1e3 * dt, sweeps / dt, 1e6 * dt / sweeps / k, e.onchip_info()), flush=True)
        res[on] = (e.get_W(), e.get_T())
        if os.environ.get('RRI_ONCHIP_TIMING') and on == '1':
            e.sweep(1)          # the library prints the sections of the previous launch when the next one starts
rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
print('on-chip vs launch-per-phase: W %.2e, T %.2e' % (rel(res['1'][0], res['0'][0]), rel(res['1'][1], res['0'][1])))
