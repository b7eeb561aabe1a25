"""sweeps/s of the topic-model flag set (T rows on the simplex every step) next to the plain flavour, C3 shape"""
import sys, time
import numpy as np
import torch
sys.path.insert(0, '.')
from rri_nmf_amd.engine import RRIEngine
n, d, k = 100000, 10000, 50
g = torch.Generator(device='cuda'); g.manual_seed(0)
Wt = torch.rand(n, k, device='cuda', generator=g) * (torch.rand(n, k, device='cuda', generator=g) < 0.3)
Tt = torch.rand(k, d, device='cuda', generator=g) * (torch.rand(k, d, device='cuda', generator=g) < 0.3)
X = (Wt @ Tt + 0.01 * torch.rand(n, d, device='cuda', generator=g)).float()
X /= X.sum(1, keepdim=True)
a = float((X.mean() / k) ** 0.5)
rs = np.random.RandomState(1)
W0, T0 = a * rs.rand(n, k), a * rs.rand(k, d)
for name, flags in (('plain', {}), ('topic model', dict(project_T_each_iter=True, t_row_sum=1.0, w_row_sum=1.0)),
                    ('topic model + W rows each sweep', dict(project_T_each_iter=True, t_row_sum=1.0, w_row_sum=1.0))):
    with RRIEngine(n, d, k, dtype=np.float32) as e:
        e.bind_X_device(X.data_ptr(), X.stride(0))
        e.set_W(W0), e.set_T(T0)
        e.set_params(**flags)
        e.sweep(2)
        e.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            e.sweep(1)
            if 'each sweep' in name:
                e.project_W_rows(1.0)
        e.synchronize()
        dt = (time.perf_counter() - t0) / 10
        e.sweep(1); e.objective(); e.sweep(1); e.synchronize()
        t1 = time.perf_counter(); o1 = e.objective(); to1 = time.perf_counter() - t1     # right after a sweep
        t1 = time.perf_counter(); e.project_W_rows(1.0); e.synchronize(); tp = time.perf_counter() - t1
        t1 = time.perf_counter(); o2 = e.objective(); to2 = time.perf_counter() - t1     # W changed: through the residual
        print('%-34s %.2f ms/sweep  %.1f sweeps/s   project_W_rows %.2f ms  objective after a sweep %.2f ms, through the residual %.2f ms'
              % (name, 1e3 * dt, 1 / dt, 1e3 * tp, 1e3 * to1, 1e3 * to2))
