"""Where the wall time of NMF_TM_Estimator.fit goes at BASELINE config 3's size (100000 x 10000 fp32 term weights, k = 50,
30 sweeps, the objective after every sweep as the reference ships): the stages of one fit, then a second fit in the same
process (no one-off costs), then transform of 5000 documents.
    python3 tools/e2e_stages.py [n d k sweeps]      (VERDICT r3 "next" 7; profiles/r04_e2e_stages.txt)"""
import collections
import sys
import time

import numpy as np

sys.path.insert(0, '.')
import torch
from rri_nmf_amd import initialization, nmf as nmf_mod, sklearn_interface as si
from rri_nmf_amd.engine import RRIEngine

n, d, k, sweeps = (int(v) for v in (sys.argv[1:5] if len(sys.argv) >= 5 else (100000, 10000, 50, 30)))
t0 = time.perf_counter()
# term counts: Poisson draws around a planted low-rank intensity with heavy-tailed factors (most words rare), mean 3 per entry
dev = torch.device('cuda', 0)
g = torch.Generator(device=dev)
g.manual_seed(0)
Ts = torch.rand(k, d, device=dev, generator=g) ** 4
Xd = torch.empty(n, d, device=dev, dtype=torch.float32)
for lo in range(0, n, 20000):
    lam = (torch.rand(min(20000, n - lo), k, device=dev, generator=g) ** 4) @ Ts
    Xd[lo:lo + lam.shape[0]] = torch.poisson(lam * (3.0 / lam.mean()), generator=g)
X = Xd.cpu().numpy()
print('counts: %.1f %% non-zero' % (100.0 * float((Xd > 0).float().mean())))
del Xd, Ts, lam
torch.cuda.empty_cache()
print('X %d x %d fp32 made in %.1f s' % (n, d, time.perf_counter() - t0), flush=True)

acc = collections.OrderedDict()


def timed(owner, name, label):
    f = getattr(owner, name)

    def g(*a, **kw):
        t = time.perf_counter()
        try:
            return f(*a, **kw)
        finally:
            acc[label] = acc.get(label, 0.0) + time.perf_counter() - t
    setattr(owner, name, g)


timed(RRIEngine, '__init__', 'handle (rri_create, allocations)')
timed(RRIEngine, 'upload_X', 'upload of X (pageable host memory)')
timed(RRIEngine, 'preprocess', 'tf-idf + row normalisation on the device')
timed(nmf_mod, 'initialize_nmf', 'start: NNDSVD (randomized SVD + factors)')
timed(RRIEngine, 'range_finder', '   of it: range finder on the device')
timed(RRIEngine, 'X_times', '   of it: products X B returned to the host')
timed(RRIEngine, 'Xt_times', '   of it: products X^T Q returned to the host')
timed(RRIEngine, 'set_W', 'set W, T')
timed(RRIEngine, 'set_T', 'set W, T')
timed(RRIEngine, 'sweep', 'sweeps')
timed(RRIEngine, 'sweep_until', 'sweeps')
timed(RRIEngine, 'objective', 'objective after a sweep')
timed(RRIEngine, 'get_W', 'download W, T')
timed(RRIEngine, 'get_T', 'download W, T')
timed(RRIEngine, 'project_W_rows', 'final projection of the rows of W')
timed(RRIEngine, 'close', 'handle closed')

for kw, name in (({'dtype': np.float32}, 'float32 storage'), ({'dtype': np.float32}, 'float32 storage, second fit of the process'),
                 ({}, 'float64 storage (the default: what the host route would yield)')):
    acc.clear()
    est = si.NMF_TM_Estimator(n, d, k, random_state=0, max_iter=sweeps, handle_tfidf=True, handle_normalization=True, nmf_kwargs=dict(kw, eps_stop=-1))
    t0 = time.perf_counter()
    est.fit(X)
    wall = time.perf_counter() - t0
    oh = est.nmf_outputs['obj_history']
    print('\n%s: fit %.2f s, %d sweeps, objective %.6g -> %.6g' % (name, wall, len(oh), oh[0], oh[-1]))
    for label, v in acc.items():
        print('    %-52s %7.3f s  %5.1f %%' % (label, v, 100.0 * v / wall))
    known = sum(v for label, v in acc.items() if not label.startswith('   '))
    print('    %-52s %7.3f s  %5.1f %%' % ('the rest (argument checks, X >= 0, numpy glue)', wall - known, 100.0 * (wall - known) / wall))
    sw = acc.get('sweeps', 0.0) + acc.get('objective after a sweep', 0.0)
    print('    share of the fit in sweeps + objective: %.0f %%' % (100.0 * sw / wall))
    assert np.all(np.diff(oh) <= 1e-12 * abs(oh[0]))
acc.clear()
t0 = time.perf_counter()
W = est.transform(X[:5000])
print('\ntransform of 5000 documents (float64 storage): %.2f s' % (time.perf_counter() - t0))
for label, v in acc.items():
    print('    %-52s %7.3f s' % (label, v))
assert abs(W.sum(1) - 1).max() < 1e-9

# one more sweep per call on the same array (one_iter, sklearn_interface.py:316-318): a handle per call against keep_resident
for keep in (False, True):
    est = si.NMF_TM_Estimator(n, d, k, random_state=0, max_iter=2, handle_tfidf=True, handle_normalization=True, keep_resident=keep,
                              nmf_kwargs={'dtype': np.float32, 'eps_stop': -1}).fit(X)
    times = []
    for _ in range(4):
        t0 = time.perf_counter()
        est.one_iter(X)
        times.append(time.perf_counter() - t0)
    print('\none_iter on the same 100000 x 10000 array, fp32 storage, %s: %s s per call'
          % ('keep_resident=True (one handle, X stays on the device)' if keep else 'a handle per call (upload + preprocessing every time)',
             ' '.join('%.3f' % v for v in times)))
    est.release()
