"""times only the pass kernel (and the whole sweep) for the current RRI_PASS_* environment"""
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
with RRIEngine(n, d, k, dtype=np.float32) as e:
    e.bind_X_device(X.data_ptr(), X.stride(0)); e.set_W(W0); e.set_T(T0); e.set_params()
    e.sweep(1)
    e.timing_enable(True)
    t0 = time.perf_counter(); e.sweep(2); wall = (time.perf_counter() - t0) / 2
    cnt, ms = e.timing_read(0)
    r1 = e.bench_rank1_update(3)
cfg = ' '.join('%s=%s' % (kk[9:], vv) for kk, vv in sorted(os.environ.items()) if kk.startswith('RRI_PASS_'))
print('%-36s pass %.4f ms %7.1f GB/s | sweep %.2f ms | rank1 %.4f ms %7.1f GB/s' % (
    cfg, ms / cnt, n * d * 4 / (ms / cnt) / 1e6, wall * 1e3, r1, 2 * n * d * 4 / r1 / 1e6))
