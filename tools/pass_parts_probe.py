"""How much of the fused pass is the read and how much the two reductions: Z-only and Y-only passes (k = 1 runs one
of each per sweep) next to the fused pass of the k = 2 schedule, same X."""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from rri_nmf_amd.engine import RRIEngine
n, d = 100000, 10000
g = torch.Generator(device='cuda'); g.manual_seed(0)
X = torch.rand(n, d, device='cuda', generator=g)
torch.cuda.synchronize()
rs = np.random.RandomState(0)
for k in (1, 2):
    W0, T0 = 0.5 * rs.rand(n, k) + 0.1, 0.5 * rs.rand(k, d) + 0.1
    with RRIEngine(n, d, k, dtype=np.float32) as e:
        e.bind_X_device(X.data_ptr(), X.stride(0)); e.set_W(W0); e.set_T(T0); e.set_params()
        e.sweep(3); e.synchronize()
        t0 = time.perf_counter(); e.sweep(40); e.synchronize(); dt = (time.perf_counter() - t0) / 40
        if k == 1:
            t0 = time.perf_counter()
            for _ in range(20):
                e.update_T_row(0)
            tz = (time.perf_counter() - t0) / 20
            print('k=1: sweep = Z-only pass + Y-only pass + small kernels: %.1f us;  T half alone (Z-only pass + small + sync): %.1f us' % (1e6 * dt, 1e6 * tz))
        else:
            print('k=2: sweep = 2 fused passes + small kernels: %.1f us  -> %.1f us per fused pass and its small kernels' % (1e6 * dt, 1e6 * dt / 2))
