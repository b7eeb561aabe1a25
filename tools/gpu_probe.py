"""Quick on-GPU probe: times the streaming kernels at a given shape with device-generated data.
usage: python tools/gpu_probe.py n d k [sweeps] [dtype]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from rri_nmf_amd.engine import RRIEngine

n, d, k = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
sweeps = int(sys.argv[4]) if len(sys.argv) > 4 else 2
dt = np.float64 if (len(sys.argv) > 5 and sys.argv[5] == 'f64') else np.float32
tdt = torch.float64 if dt == np.float64 else torch.float32
dev = torch.device('cuda:0')
g = torch.Generator(device=dev); g.manual_seed(0)
Ws = torch.rand(n, k, device=dev, generator=g, dtype=tdt) * (torch.rand(n, k, device=dev, generator=g) < 0.3)
Ts = torch.rand(k, d, device=dev, generator=g, dtype=tdt) * (torch.rand(k, d, device=dev, generator=g) < 0.3)
X = Ws @ Ts
X += 0.01 * torch.rand(n, d, device=dev, generator=g, dtype=tdt)
a = float(torch.sqrt(X.mean() / k))
W0 = (a * torch.rand(n, k, device=dev, generator=g, dtype=tdt)).cpu().numpy()
T0 = (a * torch.rand(k, d, device=dev, generator=g, dtype=tdt)).cpu().numpy()
torch.cuda.synchronize()
es = X.element_size()
with RRIEngine(n, d, k, dtype=dt) as e:
    e.bind_X_device(X.data_ptr(), X.stride(0))
    e.set_W(W0); e.set_T(T0); e.set_params()
    e.sweep(1)
    e.timing_enable(True)
    t0 = time.perf_counter()
    e.sweep(sweeps)
    wall = time.perf_counter() - t0
    out = {}
    for kid, nm in ((0, 'pass'), (1, 'wcol'), (2, 'trow_chain')):
        cnt, ms = e.timing_read(kid)
        out[nm] = (cnt, ms / max(cnt, 1))
    e.timing_enable(False)
    t0 = time.perf_counter()
    e.sweep(sweeps)
    wall2 = time.perf_counter() - t0
    copy_ms = e.bench_stream_copy(5)
    r1_ms = e.bench_rank1_update(5)
bytes_pass = n * d * es
print('shape', n, d, k, dt.__name__, 'sweeps', sweeps)
print('wall/sweep (timed events on) %.3f ms ; (events off) %.3f ms -> %.2f sweeps/s' % (1e3 * wall / sweeps, 1e3 * wall2 / sweeps, sweeps / wall2))
for nm, (cnt, ms) in out.items():
    print('%-11s launches %5d  avg %.4f ms' % (nm, cnt, ms))
print('pass: %.1f GB/s algorithmic (n*d*s / launch) = %.1f%% of 8 TB/s' % (bytes_pass / out['pass'][1] / 1e6, 100 * bytes_pass / out['pass'][1] / 1e6 / 8000))
print('stream copy: %.4f ms -> %.1f GB/s (r+w)' % (copy_ms, 2 * bytes_pass / copy_ms / 1e6))
print('rank-one update (r+w, fused dots): %.4f ms -> %.1f GB/s = %.1f%% of 8 TB/s' % (r1_ms, 2 * bytes_pass / r1_ms / 1e6, 100 * 2 * bytes_pass / r1_ms / 1e6 / 8000))
