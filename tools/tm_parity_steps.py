"""topic-model flags, half step by half step: the device (rri_update_T_row / rri_update_W_col) against the oracle's functions on
the SAME W, T before each half step (no accumulated trajectory): where does a difference beyond summation order enter?"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rri_nmf_amd.engine import RRIEngine
from rri_nmf_amd.synthetic import planted_X, scaled_init
from oracle import rri_oracle as orc
rel = lambda a, b: float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-300))
TM = dict(project_T_each_iter=True, t_row_sum=1.0, w_row_sum=1.0)
n, d, k = 10000, 1000, 20
X = planted_X(n, d, k, seed=81, dtype=np.float32); X = X / X.sum(1, keepdims=True)
W0, T0 = scaled_init(X, k, seed=82); T0 = T0 / T0.sum(1, keepdims=True)
X64 = np.asarray(X, dtype=np.float64)
W, T = np.asarray(W0, dtype=np.float64).copy(), np.asarray(T0, dtype=np.float64).copy()
os.environ['RRI_ONCHIP'] = '0'
with RRIEngine(n, d, k, dtype=np.float64) as e:
    e.upload_X(X64), e.set_params(**TM)
    for t in range(6):
        e.set_W(W), e.set_T(T)                    # the device starts every half step from the oracle's state
        e.update_T_row(t)
        Td = e.get_T()
        wR, nw = orc.residual_products_T(X64, W, T, t)
        x, nt1 = orc.qf_min(-(wR - 0.0), nw + 0.0, s=1.0, ub=1.0)
        raw = np.maximum(wR, 0) / (nw + np.spacing(10))
        print('topic %d T row: device vs oracle %.2e | nw %.6g, sum of the unprojected row %.6g, nonzeros after projection %d / %d, row sum - 1: oracle %.2e device %.2e'
              % (t, rel(Td[t], x), nw, raw.sum(), int((x > 0).sum()), d, x.sum() - 1, Td[t].sum() - 1), flush=True)
        T[t] = x
        e.set_T(T)
        e.update_W_col(t)
        Wd = e.get_W()
        Rt, nt = orc.residual_products_W(X64, W, T, t)
        w, _ = orc.qf_min(-(Rt - 0.0), nt + 0.0, s=None, ub=1.0)
        print('        W col: device vs oracle %.2e' % rel(Wd[:, t], w), flush=True)
        W[:, t] = w
