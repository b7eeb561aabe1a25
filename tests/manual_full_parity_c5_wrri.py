"""One-off: the weighted flavour against the CPU oracle at the FULL C5 size (100000 x 10000, k = 50, 5 % observed),
one sweep.  The oracle does two n x d x k products per topic step (minutes per sweep).  Not part of the test suite."""
import sys
import time

import numpy as np
import scipy.sparse as sp
import torch
from threadpoolctl import threadpool_limits

sys.path.insert(0, '.')
from oracle import rri_oracle as orc
from rri_nmf_amd.engine import RRIEngine

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
g.manual_seed(2)
Mask = torch.rand(N, D, device=dev, generator=g) < 0.05
nz = Mask.nonzero()
indptr = np.concatenate([[0], np.cumsum(torch.bincount(nz[:, 0], minlength=N).cpu().numpy())]).astype(np.int64)
A = sp.csr_matrix((X[Mask].cpu().numpy(), nz[:, 1].to(torch.int32).cpu().numpy(), indptr), shape=(N, D))
Mh = Mask.cpu().numpy().astype(np.float64)
del X, Mask, nz
torch.cuda.empty_cache()
flags = dict(t_row_sum=1.0, reset_topic_method=None)
out = {}
for name, dt in (('pattern-only, float64 residual', np.float64), ('pattern-only, fp32 residual', np.float32)):
    with RRIEngine(N, D, K, dtype=dt, weighted='sparse') as e:
        e.upload_observed_csr(A)
        e.set_W(W0); e.set_T(T0); e.set_params(**flags)
        e.sweep(1)
        out[name] = (e.get_W(), e.get_T(), e.objective())
Xh = A.toarray().astype(np.float64)
# round 4: the dense handles too (bit-packed mask; ONE read-modify-write pass per topic step + the mask-only correction)
for name, dt in (('dense, float64 residual', np.float64), ('dense, fp32 residual (what bench.py --config c5 times)', np.float32)):
    with RRIEngine(N, D, K, dtype=dt, weighted=True) as e:
        e.upload_X(Xh if dt == np.float64 else Xh.astype(dt))
        e.upload_mask(Mh if dt == np.float64 else Mh.astype(dt))
        e.set_W(W0); e.set_T(T0); e.set_params(**flags)
        e.sweep(1)
        out[name] = (e.get_W(), e.get_T(), e.objective())
print('device runs done; the oracle sweep takes ~5 minutes', flush=True)
t0 = time.perf_counter()
with threadpool_limits(limits=16):
    ref = orc.nmf(Xh, K, W_in=W0.copy(), T_in=T0.copy(), W_mat=Mh, max_iter=1, eps_stop=-1, compute_obj_each_iter=True, **flags)
tc = time.perf_counter() - t0
rel = lambda a_, b_: float(np.linalg.norm(a_ - b_) / np.linalg.norm(b_))
for name, (W, T, o) in out.items():
    print({'handle': name, 'relfro_W': rel(W, ref['W']), 'relfro_T': rel(T, ref['T']),
           'objective_rel_diff': abs(o - ref['obj_history'][-1]) / ref['obj_history'][-1]})
print({'oracle_seconds_per_sweep': round(tc, 1)})
