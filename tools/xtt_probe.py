"""X T^T (fold-in with T fixed) and rri_X_times at C3 scale: matrix cores vs vector ALU (RRI_RESID_MFMA=0)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from rri_nmf_amd.engine import RRIEngine
n, d, k = 100000, 10000, 50
g = torch.Generator(device='cuda'); g.manual_seed(0)
X = torch.rand(n, d, device='cuda', generator=g)
torch.cuda.synchronize()
rs = np.random.RandomState(0)
W0, T0 = 0.1 * rs.rand(n, k), 0.02 * rs.rand(k, d)
with RRIEngine(n, d, k, dtype=np.float32) as e:
    e.bind_X_device(X.data_ptr(), X.stride(0)); e.set_W(W0); e.set_T(T0); e.set_params(fix_T=True)
    e.sweep(1); e.synchronize()
    ts = []
    for _ in range(3):
        e.set_T(T0)                      # invalidates the cached X T^T
        t0 = time.perf_counter(); e.sweep(1); e.synchronize(); ts.append(time.perf_counter() - t0)
    B = rs.randn(d, 60)
    t0 = time.perf_counter(); out = e.X_times(B); t1 = time.perf_counter() - t0
    ref = (X[:2000].double() @ torch.from_numpy(B).cuda()).cpu().numpy()
    print('fold-in sweep incl. X T^T: %.2f ms   X_times(60 columns) incl. transfers: %.1f ms   rel err %.2e'
          % (1e3 * min(ts), 1e3 * t1, np.linalg.norm(out[:2000] - ref) / np.linalg.norm(ref)))
