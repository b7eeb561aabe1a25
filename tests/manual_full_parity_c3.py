"""One-off: the device path against the CPU oracle at the FULL C3 size (100000 x 10000 fp32, k = 50), equal sweeps.
The oracle needs ~16 s per sweep on the box's host cores; not part of the test suite.  usage (from the repository root): python tests/manual_full_parity_c3.py [sweeps]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, '.')
from oracle import rri_oracle as orc
from rri_nmf_amd.engine import RRIEngine

S = int(sys.argv[1]) if len(sys.argv) > 1 else 2
N, D, K = 100000, 10000, 50
dev = torch.device('cuda:0')
g = torch.Generator(device=dev)
g.manual_seed(0)
Ts = torch.rand(K, D, device=dev, generator=g) * (torch.rand(K, D, device=dev, generator=g) < 0.3)
X = torch.empty(N, D, device=dev, dtype=torch.float32)
for lo in range(0, N, 25000):
    Ws = torch.rand(25000, K, device=dev, generator=g) * (torch.rand(25000, K, device=dev, generator=g) < 0.3)
    torch.matmul(Ws, Ts, out=X[lo:lo + 25000])
    X[lo:lo + 25000].add_(torch.rand(25000, D, device=dev, generator=g), alpha=0.01)
a = float(torch.sqrt(X.mean(dtype=torch.float64) / K))
W0 = (a * torch.rand(N, K, device=dev, generator=g, dtype=torch.float64)).cpu().numpy()
T0 = (a * torch.rand(K, D, device=dev, generator=g, dtype=torch.float64)).cpu().numpy()
with RRIEngine(N, D, K, dtype=np.float32) as e:
    e.bind_X_device(X.data_ptr(), X.stride(0)); e.set_W(W0); e.set_T(T0); e.set_params()
    t0 = time.perf_counter(); e.sweep(S); e.synchronize(); tg = time.perf_counter() - t0
    Wg, Tg = e.get_W(), e.get_T()
Xh = X.cpu().numpy().astype(np.float64)
del X
t0 = time.perf_counter()
Wc, Tc = orc.plain_sweeps(Xh, W0.copy(), T0.copy(), S)
tc = time.perf_counter() - t0
W1 = W0.copy(); W1[0, 0] = np.nextafter(W1[0, 0], 1.0)        # the reference against itself, start perturbed by 1 ulp
Wp, Tp = orc.plain_sweeps(Xh, W1, T0.copy(), S)
rel = lambda a_, b_: float(np.linalg.norm(a_ - b_) / np.linalg.norm(b_))
print({'sweeps': S, 'device_s': round(tg, 3), 'oracle_s': round(tc, 1),
       'relfro_W': rel(Wg, Wc), 'relfro_T': rel(Tg, Tc), 'relfro_WT_rows0_2000': rel(Wg[:2000] @ Tg, Wc[:2000] @ Tc),
       'oracle_self_sensitivity_1ulp': {'W': rel(Wp, Wc), 'T': rel(Tp, Tc)}})
