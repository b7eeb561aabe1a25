"""What the stages of one handle cost at the text fixture's size (100 x 200, k = 5): create, upload, first / second sweep, objective,
a launch of 28 sweeps, read-back, close.
    python3 tools/handle_lifecycle_time.py"""
import sys, time
import numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rri_nmf_amd.engine import RRIEngine
n, d, k = 100, 200, 5
X = np.random.rand(n, d).astype(np.float32); W0 = np.random.rand(n, k); T0 = np.random.rand(k, d)
def t(): return time.perf_counter()
for rep in range(4):
    a = t(); e = RRIEngine(n, d, k, dtype=np.float32); b = t(); e.upload_X(X); c = t(); e.set_W(W0); e.set_T(T0); e.set_params(); d0 = t()
    e.sweep(1); f = t(); e.sweep(1); g = t(); e.objective(); h = t(); e.sweep(28); i = t(); W = e.get_W(); T = e.get_T(); j = t(); e.close(); kk = t()
    print('create %.0f us, upload X %.0f, set W/T/params %.0f, first sweep %.0f, second %.0f, objective %.0f, 28 sweeps %.0f, get W,T %.0f, close %.0f'
          % tuple(1e6 * v for v in (b - a, c - b, d0 - c, f - d0, g - f, h - g, i - g - (h - g), j - i, kk - j)))
