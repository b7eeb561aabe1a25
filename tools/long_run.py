"""Long runs at the full C3 size: 300 sweeps per flavour, objective after every sweep must not increase."""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from rri_nmf_amd.engine import RRIEngine
import bench
n, d, k = 100000, 10000, 50
dev = torch.device('cuda:0')
X = bench.device_planted_shard(n, d, k, seed=0, device=dev)
X /= X.sum(1, keepdim=True)
a = float((X.mean(dtype=torch.float64) / k) ** 0.5)
rs = np.random.RandomState(1)
W0, T0 = a * rs.rand(n, k), a * rs.rand(k, d)
torch.cuda.synchronize()
for name, flags in (('plain', {}), ('topic model', dict(project_T_each_iter=True, t_row_sum=1.0, w_row_sum=1.0))):
    with RRIEngine(n, d, k, dtype=np.float32) as e:
        e.bind_X_device(X.data_ptr(), X.stride(0)); e.set_W(W0); e.set_T(T0); e.set_params(**flags)
        objs = [e.objective()]
        t0 = time.perf_counter()
        for s in range(300):
            e.sweep(1)
            objs.append(e.objective())
        dt = time.perf_counter() - t0
        W = e.get_W()
        worst = max((b - a_) / abs(a_) for a_, b in zip(objs, objs[1:]))
        print('%-12s 300 sweeps + objectives in %.1f s (%.1f iterations/s); objective %.6e -> %.6e; largest relative increase %.2e; resets %d; finite %s'
              % (name, dt, 300 / dt, objs[0], objs[-1], worst, e.n_resets_used, bool(np.isfinite(W).all())))
        assert worst < 1e-9 and np.isfinite(W).all()
