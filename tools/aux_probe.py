"""times the around-the-loop device operations at a given shape"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rri_nmf_amd.engine import RRIEngine
n, d, k = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dev = torch.device('cuda:0')
g = torch.Generator(device=dev); g.manual_seed(0)
X = torch.rand(n, d, device=dev, generator=g)
W0 = (0.1 * torch.rand(n, k, device=dev, generator=g, dtype=torch.float64)).cpu().numpy()
T0 = (0.1 * torch.rand(k, d, device=dev, generator=g, dtype=torch.float64)).cpu().numpy()
torch.cuda.synchronize()
def timed(f, reps=3):
    f(); t0 = time.perf_counter()
    for _ in range(reps): f()
    return (time.perf_counter() - t0) / reps * 1e3
with RRIEngine(n, d, k, dtype=np.float32) as e:
    e.bind_X_device(X.data_ptr(), X.stride(0)); e.set_W(W0); e.set_T(T0); e.set_params()
    e.sweep(1)
    print('objective       %.2f ms' % timed(e.objective))
    print('project_W_rows  %.2f ms' % timed(lambda: e.project_W_rows(1.0)))
    print('get_W + get_T   %.2f ms' % timed(lambda: (e.get_W(), e.get_T())))
    print('set_W           %.2f ms' % timed(lambda: e.set_W(W0)))
    print('argmax_rows     %.2f ms' % timed(e.argmax_rows))
    print('snapshot        %.2f ms' % timed(lambda: (e.snapshot(), e.synchronize())))

    e.set_params(fix_T=True, w_row_sum=1.0, t_row_sum=1.0)
    t0 = time.perf_counter(); e.sweep(4); e.project_W_rows(1.0); dt = time.perf_counter() - t0
    print('fold-in (fix_T, 4 sweeps + projection) first call  %.2f ms' % (dt * 1e3))
    t0 = time.perf_counter(); e.sweep(4); dt = time.perf_counter() - t0
    print('fold-in 4 more sweeps (X T^T cached)              %.2f ms' % (dt * 1e3))
